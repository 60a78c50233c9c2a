import sys, numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso
from meso_amd.datagen import make_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x, v, lo, hi = make_box(L)
res = []
for opts in ((("fused_rebuild", 0),), ()):
    m = Meso(0)
    for k, val in opts: m.set_option(k, val)
    m.set_option("ghost_epilogue", 0)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 0.0, 1.0, 1.0); m.timestep(0.005)
    m.setup(); m.run(nsteps)
    nl, nb, ng = m.counts()
    c4, v4 = m.merged()
    print(opts, m.counts(), c4.shape)
    res.append((nl, ng, c4.copy(), v4.copy())); m.close()
(nl, nga, ca, va), (_, ngb, cb, vb) = res
print("locals equal:", np.array_equal(ca[:nl], cb[:nl]))
ga = ca[nl:nl + nga]; gb = cb[nl:nl + ngb]
sa = set(map(lambda r: r.tobytes(), ga)); sb = set(map(lambda r: r.tobytes(), gb))
print("ghosts chain %d (distinct %d) fused %d (distinct %d), only in chain %d, only in fused %d" % (nga, len(sa), ngb, len(sb), len(sa - sb), len(sb - sa)))
for r in list(sa - sb)[:10]: print("missing in fused", np.frombuffer(r, np.float32))
for r in list(sb - sa)[:10]: print("extra in fused", np.frombuffer(r, np.float32))
