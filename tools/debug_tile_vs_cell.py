"""Compare the wave-per-bin tile builder with the lane-per-atom cell builder on the same atoms (large box, late state)."""
import numpy as np, sys
from meso_amd.api import Meso
from meso_amd.datagen import make_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
x, v, lo, hi = make_box(L)
m = Meso(0)
m.set_option("neigh_kernel", 0)
m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
m.setup()
m.run(steps)
m.reneighbor()
ca, ta = m.neigh_table()
m.set_option("neigh_kernel", 1)
m.reneighbor()
cb, tb = m.neigh_table()
print("counts equal:", np.array_equal(ca, cb), "sum", ca.sum(), cb.sum(), "max", ca.max(), cb.max())
bad = np.nonzero(ca != cb)[0]
print("atoms with different count:", len(bad), bad[:10])
nbad = 0
for i in range(len(ca)):
    if ca[i] == cb[i]:
        if not np.array_equal(np.sort(ta[i, :ca[i]]), np.sort(tb[i, :cb[i]])):
            nbad += 1
            if nbad < 5: print("set differs at", i, np.setxor1d(ta[i, :ca[i]], tb[i, :cb[i]]))
print("atoms with equal count but different set:", nbad)
for i in bad[:5]:
    print(i, "cell", ca[i], "tile", cb[i], "missing", np.setdiff1d(ta[i, :ca[i]], tb[i, :cb[i]])[:10], "extra", np.setdiff1d(tb[i, :cb[i]], ta[i, :ca[i]])[:10])
