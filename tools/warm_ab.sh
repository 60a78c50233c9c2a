#!/bin/bash
# tools/warm_ab.sh [bench args]: the driver's short timed window (--steps 20) after 5, 10, 25, 50 and 200 warm-up steps, and 1000 after 200
mkdir -p gpurun_out/q
for cfg in "20 5" "20 10" "20 25" "20 50" "20 200" "1000 200" "20 5"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline --profile-steps 20 ${WARM_ARGS} > gpurun_out/q/warm.json 2>gpurun_out/q/warm.err || { echo "$cfg FAILED"; tail -3 gpurun_out/q/warm.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/q/warm.json").read().strip().splitlines()[-1])
print("steps $1 warmup $2: %.0f steps/s  %.1f us/step" % (d["value"], d["ms_per_step"]*1e3))
PY
done
