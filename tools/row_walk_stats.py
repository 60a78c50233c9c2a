"""tools/row_walk_stats.py [BOX]: how many 8-entry chunks a wave of the force kernel walks (its longest row) against what its atoms need, on the
front sections of a thermalised table - and what dealing the 256 atoms of a workgroup to the lanes by row length would give
(profiles/r06_notes.md section 8)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 48
x, v, lo, hi = make_box(L)
m = Meso()
m.set_option("pair_npart", 1)
m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
m.setup(); m.run(300)
p = m.neigh_parts()
nf = p["nfront"].astype(np.int64); n = len(nf) // 256 * 256
nf = nf[:n]
ch = (nf + 7) // 8
w = ch.reshape(-1, 64)
print("atoms", n, "mean front", nf.mean(), "std", nf.std(), "mean chunks/atom", ch.mean())
print("chunk iterations per wave now: mean of wave max", w.max(1).mean(), " (ideal = mean chunks", ch.mean(), ")")
g = np.sort(ch.reshape(-1, 256), axis=1)[:, ::-1].reshape(-1, 4, 64)
print("sorted inside the 256-group: mean of wave max", g.max(2).mean())
# entries-level: sum over lanes of wave max*8 vs entries
print("slot efficiency now", nf.sum() / (w.max(1).sum() * 64 * 8), " sorted", nf.sum() / (g.max(2).sum() * 64 * 8))
# with 4-entry chunks
ch4 = (nf + 3) // 4
print("4-entry chunks: now", ch4.reshape(-1, 64).max(1).mean() / 2, "sorted", np.sort(ch4.reshape(-1, 256), axis=1)[:, ::-1].reshape(-1, 4, 64).max(2).mean() / 2, "(in units of 8-entry chunks)")
m.close()
