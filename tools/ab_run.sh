mkdir -p gpurun_out/ab
for box in 64 32 25; do
 for v in A B A B; do
  if [ $v = A ]; then export MESO_LIB=$PWD/meso_amd/libmeso_hip_A.so; else unset MESO_LIB; fi
  timeout -k 10 200 python bench.py --box $box --steps 2000 --warmup 200 --no-cpu-baseline --profile-steps 50 > gpurun_out/ab/${box}_$v.json 2>gpurun_out/ab/err.txt || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/${box}_$v.json").read().strip().splitlines()[-1])
print("$box $v", d["value"])
PY
 done
done
