#!/bin/bash
# tools/ab_state.sh "OPTS_A" "OPTS_B" [bench args]: the state (x, v, f by tag) after the same bench run under two sets of engine options must be
# bit-identical for variants that keep the order of the row entries (scheduling, refresh path, ...): a check at full size.  (Variants
# that change the entry order - row_part - differ in the last bits of setup()'s forces, which come from the lane-per-atom kernel's
# per-thread floating-point sums, and the thermostat amplifies that: tests/test_gpu_configs_at_size.py compares those from the ring
# kernel's forces on.)
mkdir -p gpurun_out/ab
a=$1; b=$2; shift 2
for v in A B; do
  if [ $v = A ]; then o=$a; else o=$b; fi
  args=""
  for kv in $o; do args="$args --opt $kv"; done
  timeout -k 10 300 python3 bench.py --steps ${STEPS:-300} --warmup 50 --no-cpu-baseline --profile-steps 20 --dump-state gpurun_out/ab/state_$v.npy $args "$@" > gpurun_out/ab/s_$v.json 2>gpurun_out/ab/s_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/ab/s_$v.err; exit 1; }
done
python3 - <<PY
import numpy as np
a, b = np.load("gpurun_out/ab/state_A.npy"), np.load("gpurun_out/ab/state_B.npy")
print("states equal bit for bit:", np.array_equal(a, b), " max |diff|", float(np.abs(a - b).max()), " shape", a.shape)
raise SystemExit(0 if np.array_equal(a, b) else 1)
PY
rm -f gpurun_out/ab/state_A.npy gpurun_out/ab/state_B.npy
