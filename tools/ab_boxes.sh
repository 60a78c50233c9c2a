#!/bin/bash
# tools/ab_boxes.sh "BOXES" lib1 lib2 ...: bench.py per box under several in-tree builds ('-' = the default library), interleaved rounds
boxes=$1; shift
mkdir -p gpurun_out/q
for b in $boxes; do
for round in 1 2; do
  for lib in "$@"; do
    if [ $lib = - ]; then unset MESO_LIB; else export MESO_LIB=$PWD/meso_amd/libmeso_hip_$lib.so; fi
    timeout -k 10 300 python3 bench.py --box $b --no-cpu-baseline --steps ${STEPS:-600} --warmup 100 $EXTRA 2>gpurun_out/q/ab_$lib.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['phases_ms']
print('box $b %-8s round $round' % '$lib', 'steps/s %.0f' % d['value'], 'pair alone %.2f' % r['us_per_launch'], 'fused %.2f' % r['fused']['us_per_launch'], 'neigh %.1f reorder %.1f' % (p['neigh']*1e3, p['reorder']*1e3), 'T %.4f' % d['config']['temperature_end'])" || { echo "$lib failed"; tail -3 gpurun_out/q/ab_$lib.err; }
  done
done
done
