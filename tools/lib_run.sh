# bench.py under several in-tree builds in one GPU session: LIBS="A W8" -> meso_amd/libmeso_hip_<name>.so ('-' = the default library)
mkdir -p gpurun_out/lib
for rep in 1 2; do
for box in ${BOXES:-64}; do
 for v in ${LIBS}; do
  if [ $v = - ]; then unset MESO_LIB; else export MESO_LIB=$PWD/meso_amd/libmeso_hip_$v.so; fi
  timeout -k 10 200 python3 bench.py --box $box --steps ${STEPS:-1000} --warmup 100 --no-cpu-baseline --profile-steps 50 ${EXTRA} > gpurun_out/lib/${box}_$v.json 2>gpurun_out/lib/err.txt || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/lib/${box}_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("$box %-6s %.0f steps/s  pair %.1f us (only %.1f)  neigh %.0f reorder %.0f bin %.0f" % ("$v", d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3))
PY
 done
done
done
