#!/bin/bash
# timing ablations of the list builder (MESO_PAIR_DEBUG: 11 = staging only, 12 = no rows out, 13 = no scan): tools/ab_build_dbg.sh box
box=${1:-32}
for dbg in 0 11 12 13; do
  MESO_PAIR_DEBUG=$dbg python3 bench.py --box $box --no-cpu-baseline --steps 300 --warmup 50 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg $dbg', 'steps/s %.0f' % d['value'], 'neigh us %.1f' % (1e3*d['phases_ms']['neigh']), 'reorder us %.1f' % (1e3*d['phases_ms']['reorder']))" || exit 1
done
