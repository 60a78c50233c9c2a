import sys, numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso
from meso_amd.datagen import make_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
style = sys.argv[2] if len(sys.argv) > 2 else "dpd/meso"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 23
x, v, lo, hi = make_box(L)
res = []
for opts in ((("fused_rebuild", 0),), (), (("fused_cap", 2),)):
    m = Meso(0)
    for k, val in opts: m.set_option(k, val)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style(style, 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    m.setup(); m.run(nsteps)
    g = m.gather()
    print(opts, "counts", m.counts(), m.neigh_info())
    res.append(g); m.close()
for k in (1, 2):
    for name, a, b in zip("xvf", res[0][:3], res[k][:3]):
        d = np.abs(a - b)
        print(k, name, "max", d.max(), "n differing atoms", int((d.max(1) > 0).sum()))
