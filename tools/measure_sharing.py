"""Fraction of in-range pairs (and of list entries) whose two atoms share a 64- / 256- / 512-atom aligned group of the
cell-ordered storage order: the evaluations a workgroup-level Newton pairing would save."""
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box
x, v, lo, hi = make_box(32)
m = Meso(0)
m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
m.setup(); m.run(50); m.reneighbor()
count, table = m.neigh_table()
c4 = m.merged()[0][:, :3]
n = len(count)
valid = np.arange(table.shape[1])[None, :] < count[:, None]
ii = np.repeat(np.arange(n), table.shape[1]).reshape(table.shape)[valid]
jj = table[valid]
d = c4[ii] - c4[jj]
r2 = (d * d).sum(1)
inr = r2 < 1.0
print("entries/atom", len(ii) / n, "in range/atom", inr.sum() / n, "n_bulk-ish", n)
for g in (6, 7, 8, 9, 10):
    same = (ii >> g) == (jj >> g)
    same &= jj < n
    print("group", 1 << g, ": list entries internal %.3f, in-range pairs internal %.3f" % (same.mean(), same[inr].mean()))
