import sys, numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso
from meso_amd.datagen import make_box
x, v, lo, hi = make_box(25)
for stride, opts in ((20, ()), (10, ()), (10, (("fused_rebuild", 0),)), (10, (("ghost_epilogue", 0),)), (7, ())):
    m = Meso(0)
    for k, val in opts: m.set_option(k, val)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    m.setup(); m.run(1000)
    T = []; T0 = []
    for k in range(6):
        m.run(stride)
        T0.append(m.temperature())
        m.force_clear("local"); m.compute(eflag=1, vflag=1)
        T.append(m.temperature())
    print(stride, opts, "T before compute", np.round(T0, 4), "after", np.round(T, 4), "pe", round(m.pe() / len(x), 4), "P", round(m.pressure(), 3))
    m.close()
