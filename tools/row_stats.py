"""Row-length statistics of the force kernel's light phase: chunks of 8 entries per lane (atom), mean and max over the 64 lanes of a wave,
and what a redistribution of the chunks beyond a cut would leave (tools/row_stats.py [box])."""
import sys
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box

L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x, v, lo, hi = make_box(L)
m = Meso()
m.read_atoms(x, v, lo, hi)
m.neighbor(0.3)
m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 12345)
m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
m.timestep(0.005)
m.setup()
m.run(200)
count, _ = m.neigh_table()
n = len(count) // 64 * 64
c = count[:n].astype(np.int64)
nch = (c + 7) // 8
w = nch.reshape(-1, 64)
print("atoms %d  mean row %.2f  sd %.2f  chunks per lane: mean %.3f  wave max: mean %.3f" % (n, c.mean(), c.std(), nch.mean(), w.max(1).mean()))
for cut in (3, 4, 5):
    rest = np.maximum(w - cut, 0).sum(1)                 # chunks beyond the cut, per wave
    passes = (rest + 63) // 64
    walked = np.minimum(w.max(1), cut) + passes
    print("cut %d: chunks left per wave mean %.1f max %d -> passes %.3f, iterations %.3f (now %.3f)" % (cut, rest.mean(), rest.max(), passes.mean(), walked.mean(), w.max(1).mean()))
m.close()
