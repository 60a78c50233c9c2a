#!/bin/bash
# tools/opt_ab.sh "tag:bench args" ... : A/B of bench.py invocations in one GPU session (two repetitions)
mkdir -p gpurun_out/q
for rep in 1 2; do
for spec in "$@"; do
  tag=${spec%%:*}; args=${spec#*:}
  timeout -k 10 200 python3 bench.py --steps ${STEPS:-1000} --warmup 200 --no-cpu-baseline --profile-steps 50 $args > gpurun_out/q/$tag.json 2>gpurun_out/q/$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/q/$tag.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/q/$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("%-22s %7.0f steps/s  fused %.1f us  pair-only %.1f us  neigh %.0f reorder %.0f bin %.0f halo %.1f T %.3f" % ("$tag", d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3, (p["halo"] or 0)*1e3, d["config"]["temperature_end"]))
PY
done
done
