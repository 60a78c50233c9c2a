#!/bin/bash
# tools/opt_ab2.sh "OPTS_A" "OPTS_B" [bench args]: bench.py twice each under two sets of engine options ('-' = none), interleaved, one GPU session
mkdir -p gpurun_out/ab
a=$1; b=$2; shift 2
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then o=$a; else o=$b; fi
  args=""
  if [ "$o" != "-" ]; then for kv in $o; do args="$args --opt $kv"; done; fi
  timeout -k 10 300 python3 bench.py --steps ${STEPS:-600} --warmup 100 --no-cpu-baseline --other-boxes "" --profile-steps 50 $args "$@" > gpurun_out/ab/o_$v.json 2>gpurun_out/ab/o_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/ab/o_$v.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab/o_$v.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("$v [%s] %.0f steps/s  fused %.1f us  pair-only %.1f us (frac %.3f)  neigh %.0f reorder %.0f bin %.0f T %.4f %s" % ("$o", d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], r["frac"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3, d["config"]["temperature_end"], r["kernel_variant"]))
PY
done
done
