#!/bin/bash
# A/B of force-kernel builds in one GPU session: tools/ab_pair.sh box lib1 lib2 ... (interleaved rounds, force kernel alone and whole step)
box=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do
    MESO_LIB=$PWD/meso_amd/$lib python3 bench.py --box $box --no-cpu-baseline --steps 600 --warmup 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', 'round $round', 'steps/s %.0f' % d['value'], 'pair alone us %.2f' % r['us_per_launch'], 'fused us %.2f' % r['fused']['us_per_launch'], 'T %.4f' % d['config']['temperature_end'])" || exit 1
  done
done
