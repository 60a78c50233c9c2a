# A/B of two builds inside one GPU session: A = meso_amd/libmeso_hip_A.so (MESO_LIB), B = the in-tree library
mkdir -p gpurun_out/ab
for box in ${BOXES:-64 32}; do
 for v in A B A B; do
  if [ $v = A ]; then export MESO_LIB=$PWD/meso_amd/libmeso_hip_A.so; else unset MESO_LIB; fi
  timeout -k 10 200 python3 bench.py --box $box --steps ${STEPS:-2000} --warmup 200 --no-cpu-baseline --profile-steps 50 ${EXTRA} > gpurun_out/ab/${box}_$v.json 2>gpurun_out/ab/err.txt || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab/${box}_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("$box $v %.0f steps/s  pair %.1f us (only %.1f)  neigh %.0f reorder %.0f bin %.0f" % (d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3))
PY
 done
done
