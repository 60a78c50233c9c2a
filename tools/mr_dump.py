#!/usr/bin/env python3
"""tools/mr_dump.py OUT.npz [L] [steps] [chain fraction]: 8 LOCAL ranks (2x2x2) of the rho=4 fluid, tag-ordered x, v, f after the run -> OUT.npz
(bit-level comparison of two builds of the library through MESO_LIB)"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_configs_at_size as T
from meso_amd.datagen import make_box
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 23
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
if frac > 0:
    from meso_amd.datagen import make_polymer_box
    deck = make_polymer_box(L, frac=frac)
else:
    deck = make_box(L)
got, counts, Tt, info = T._ranks(8, (2, 2, 2), deck, "dpd/fast/meso", steps, want=("setup", "end"), timeout=120)
np.savez(sys.argv[1], x=got["end"][0], v=got["end"][1], f=got["end"][2], f0=got["setup"][2])
print("wrote", sys.argv[1], counts[0])
