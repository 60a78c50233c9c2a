#!/bin/bash
# tools/lib_ab.sh "LIBS" [bench args]: bench.py under several in-tree builds ('-' = default library) in one GPU session
mkdir -p gpurun_out/q
libs=$1; shift
for rep in 1 2; do
for v in $libs; do
  if [ $v = - ]; then unset MESO_LIB; else export MESO_LIB=$PWD/meso_amd/libmeso_hip_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps ${STEPS:-600} --warmup 100 --no-cpu-baseline --profile-steps 50 "$@" > gpurun_out/q/lib_$v.json 2>gpurun_out/q/lib_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/q/lib_$v.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/q/lib_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("%-10s %.0f steps/s  fused %.1f us  pair-only %.1f us (frac %.3f)  neigh %.0f reorder %.0f bin %.0f T %.3f" % ("$v", d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], r["frac"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3, d["config"]["temperature_end"]))
PY
done
done
