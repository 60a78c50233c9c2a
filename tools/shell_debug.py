#!/usr/bin/env python3
"""Where does the shell walk first differ from the plain-row kernel?  tools/shell_debug.py [L] [style]"""
import sys, numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso
from meso_amd.datagen import make_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
style = sys.argv[2] if len(sys.argv) > 2 else "dpd/fast/meso"
x, v, lo, hi = make_box(L)
def run(opts, steps):
    m = Meso()
    for k, val in opts: m.set_option(k, val)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style(style, 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    m.setup(); m.run(steps)
    out = m.gather()[:3]; name = m.pair_kernel_name(); m.close()
    return out, name
for steps in (1, 2, 3, 5, 6, 7):
    ref, n0 = run((("shell_walk", 0),), steps)
    for opts in ((("shell_walk", 2),), (("shell_walk", 2), ("pair_share", 0)), (("shell_walk", 2), ("pair_npart", 1)), (), (("fuse_pair", 0),)):
        got, n1 = run(opts, steps)
        bad = [int((a != b).any(1).sum()) for a, b in zip(ref, got)]
        print(steps, opts, n1, "differing atoms x/v/f:", bad, "max |df|", np.abs(ref[2] - got[2]).max())
