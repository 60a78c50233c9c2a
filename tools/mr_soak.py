#!/usr/bin/env python3
"""tools/mr_soak.py [L] [steps] [frac]: 8 in-process ranks (2x2x2, LOCAL transport) of a polymer deck for many steps - every tag
still there exactly once, momentum conserved, temperature sane, and the same numbers from the one-rank run of the same deck."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_configs_at_size as T
from meso_amd.datagen import make_polymer_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.3
deck = make_polymer_box(L, frac=frac)
n = len(deck[0])
for nr, grid in ((1, (1, 1, 1)), (8, (2, 2, 2))):
    got, counts, Tt, info = T._ranks(nr, grid, deck, "dpd/fast/meso", steps, want=("end",), timeout=600)
    end = got["end"]
    ok = np.array_equal(end[3], np.arange(1, n + 1))
    x, v, types, bonds, lo, hi = deck
    d = end[0][bonds[:, 0] - 1] - end[0][bonds[:, 1] - 1]
    d -= np.round(d / (hi - lo)) * (hi - lo)
    r = np.sqrt((d * d).sum(axis=1))
    print("%d rank(s): tags once %s  |sum v| %.3e  T %.4f  bond length mean %.3f max %.3f  locals %s" % (
        nr, ok, np.abs(end[1].sum(axis=0)).max(), Tt[0], r.mean(), r.max(), [c[0] for c in counts]), flush=True)
