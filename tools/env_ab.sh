#!/bin/bash
# tools/env_ab.sh "VAR=a VAR=b ..." [bench args]: bench.py under several settings of one environment variable ('-' = unset), interleaved, one GPU session
mkdir -p gpurun_out/q
sets=$1; shift
for rep in 1 2; do
for kv in $sets; do
  if [ "$kv" = - ]; then pre=""; else pre="$kv"; fi
  env $pre timeout -k 10 200 python3 bench.py --steps ${STEPS:-600} --warmup 100 --no-cpu-baseline --profile-steps 50 "$@" > gpurun_out/q/env.json 2>gpurun_out/q/env.err || { echo "$kv FAILED"; tail -3 gpurun_out/q/env.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/q/env.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("%-26s %.0f steps/s  fused %.1f us  pair-only %.1f us (frac %.3f)  neigh %.0f reorder %.0f bin %.0f T %.3f" % ("$kv", d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], r["frac"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3, d["config"]["temperature_end"]))
PY
done
done
