import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box
for opts in ((), (("fused_cap", 2),)):
    x, v, lo, hi = make_box(32)
    m = Meso()
    for k, val in opts: m.set_option(k, val)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    print("setup", opts, flush=True)
    m.setup()
    print("run", flush=True)
    m.run(12)
    print("redone", m.timer("rebuilds_redone"), flush=True)
    m.close()
