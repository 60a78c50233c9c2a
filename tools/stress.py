"""Longer runs than the test-suite allows: conservation and stability checks (not timing)."""
import threading, sys
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box, make_polymer_box

def base(m, x, v, lo, hi, style, types=None, ntypes=1):
    m.read_atoms(x, v, lo, hi, types=types, ntypes=ntypes); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style(style, 1.0, 419084618)
    for i in range(1, ntypes + 1):
        for j in range(i, ntypes + 1):
            m.pair_coeff(i, j, 15.0 if i == j else 40.0, 4.5, 3.0, 1.0, 1.0)
    m.timestep(0.005)

for style in ("dpd/fast/meso", "dpd/meso"):
    x, v, lo, hi = make_box(32)
    m = Meso(0); base(m, x, v, lo, hi, style); m.setup()
    m.run(10000)
    xg, vg, fg, tag, typ = m.gather()
    print(style, "32^3 10000 steps: T %.4f  |p| %.2e  atoms ok %s  finite %s  rebuilds %d" % (
        m.temperature(), np.abs(vg.sum(0)).max(), np.array_equal(tag, np.arange(1, len(x) + 1)), np.isfinite(xg).all(), m.neigh_info()["nbuild"]))
    m.close()

x, v, types, bonds, lo, hi = make_polymer_box(14, frac=0.2)
m = Meso(0); m.read_atoms(x, v, lo, hi, types=types, ntypes=2); m.special_bonds(0.0, 1.0, 1.0); m.read_bonds(bonds)
m.bond_style("harmonic/meso", 1); m.bond_coeff(1, 50.0, 0.5); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618)
for (i, j), a in {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}.items(): m.pair_coeff(i, j, a, 4.5, 3.0, 1.0, 1.0)
m.timestep(0.005); m.setup(); m.run(3000)
print("polymer 14^3 3000 steps: T %.4f  ebond/bond %.4f  atoms %d" % (m.temperature(), m.ebond() / len(bonds), m.counts()[0]))
m.close()

nr, grid, L = 8, (2, 2, 2), 20
x, v, lo, hi = make_box(L)
gid = np.frombuffer(np.random.default_rng(5).bytes(8), np.uint8)
res = [None] * nr
def work(r):
    m = Meso(); m.comm_init(nr, r, grid, "local", gid); base(m, x, v, lo, hi, "dpd/fast/meso"); m.setup(); m.run(1500)
    res[r] = (m.temperature(), m.counts()[0], m.gather(by_tag=False)[3]); m.close()
th = [threading.Thread(target=work, args=(r,)) for r in range(nr)]
[t.start() for t in th]; [t.join() for t in th]
tags = np.sort(np.concatenate([r[2] for r in res]))
print("8 ranks 20^3 1500 steps: T %.4f  atoms %d of %d  unique %s" % (res[0][0], sum(r[1] for r in res), len(x), np.array_equal(tags, np.arange(1, len(x) + 1))))
