import numpy as np, sys
from meso_amd.api import Meso, MesoError
from meso_amd.datagen import make_box
L = 64
x, v, lo, hi = make_box(L)
m = Meso(0)
for kv in sys.argv[1:]:
    k, val = kv.split('='); m.set_option(k, float(val))
m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
m.setup()
step = 0
try:
    for k in range(120):
        m.run(5); step += 5
    print("no failure in", step, "steps", "T", m.temperature())
except MesoError as e:
    print("failed after step", step, ":", e)
xg = m.gather()[0]
print("pos range", xg.min(0), xg.max(0), "nan", np.isnan(xg).sum())
cut = 1.3; mb = int(L / cut) + 2; bs = L / (mb - 2)
cur = xg % L
for d in range(3):
    a = cur[cur[:, d] <= cut].copy(); a[:, d] += L
    b = cur[cur[:, d] >= L - cut].copy(); b[:, d] -= L
    cur = np.concatenate([cur, a, b])
b = np.clip(np.floor(cur / bs + 1.0).astype(int), 0, mb - 1)
H = np.zeros((mb + 8,) * 3, int)
np.add.at(H, (b[:, 0] + 1, b[:, 1] + 1, b[:, 2] + 1), 1)
print("bin max", H.max(), "ghosts", len(cur) - len(xg))
best = (0, None)
for bx in range(0, mb, 4):
    for by in range(0, mb, 4):
        for bz in range(0, mb, 4):
            s = H[bx:bx + 6, by:by + 6, bz:bz + 6].sum()
            if s > best[0]: best = (s, (bx, by, bz))
print("halo max", best)
