#!/bin/bash
# tools/neigh_ablate.sh [bench args]: list builder time per build, whole and with parts switched off (pair_debug 11: staging only, 12: no row-out, 13: no scan)
mkdir -p gpurun_out/ab
for o in 0 11 12 13; do
  args=""; [ $o != 0 ] && args="--opt pair_debug=$o"
  timeout -k 10 300 python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --other-boxes "" --profile-steps 100 $args "$@" > gpurun_out/ab/n_$o.json 2>gpurun_out/ab/n_$o.err || { echo "$o FAILED"; tail -5 gpurun_out/ab/n_$o.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab/n_$o.json").read().strip().splitlines()[-1])
p=d["phases_ms"]
print("pair_debug $o: neigh %.1f us  reorder %.1f  bin %.1f" % (p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3))
PY
done
