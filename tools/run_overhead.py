import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from meso_amd.api import Meso
from meso_amd.datagen import make_box
x, v, lo, hi = make_box(64)
m = Meso(0)
m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
m.setup(); m.run(200); m.sync()
for K in (5, 10, 20, 40, 80, 160, 20, 5):
    ts = []
    for rep in range(5):
        m.sync(); t0 = time.perf_counter(); m.run(K); m.sync(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print("K %4d  %.1f us total  %.1f us/step" % (K, t * 1e6, t * 1e6 / K), flush=True)
