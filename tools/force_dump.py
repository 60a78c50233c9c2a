"""forces of one configuration with the library MESO_LIB names, by tag: tools/force_dump.py box out.npy [steps]  (A/B of kernel builds)"""
import os
import sys
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box

L = int(sys.argv[1]); out = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
x, v, lo, hi = make_box(L)
m = Meso()
m.read_atoms(x, v, lo, hi)
m.neighbor(0.3)
m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, int(os.environ.get("FD_SEED", "12345")))
m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
m.timestep(0.005)
m.setup()
for kv in os.environ.get("FD_OPTS", "").split():
    k, val = kv.split("=")
    m.set_option(k, int(val))
if os.environ.get("FD_BENCH"):
    # the sequence of bench.py in front of its timed region
    m.run(100); m.set_option("profile", 1); m.timer_reset(); m.run(20); m.set_option("profile", 0)
    if os.environ["FD_BENCH"] >= "2":
        m.set_option("fuse_pair", 0); m.timer_reset(); m.set_option("profile", 1); m.run(20); m.set_option("profile", 0); m.set_option("fuse_pair", 1)
    if os.environ["FD_BENCH"] >= "3": m.membw_probe(1 << 30, 5)
if steps: m.run(steps)
m.force_clear("local")
m.compute()
g = m.gather()
np.save(out, np.hstack([g[0], g[1], g[2]]))
m.close()
