#!/usr/bin/env python3
"""tools/local_ranks_bench.py NRANKS GX GY GZ [L] [STEPS]: N engine contexts of one process on the in-process LOCAL transport,
all on the one GPU of the box - the device work and host overheads of the decomposed path without any wire (RCCL needs one GPU
per rank).  Prints whole-job steps/s and the per-phase timers of rank 0."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso
from meso_amd.datagen import make_box

n = int(sys.argv[1]); grid = tuple(int(a) for a in sys.argv[2:5])
L = int(sys.argv[5]) if len(sys.argv) > 5 else 64
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 200
x, v, lo, hi = make_box(L)
gid = np.frombuffer(np.random.default_rng(5).bytes(8), np.uint8)
bar = threading.Barrier(n)
out = {}

def work(r):
    m = Meso()
    if n > 1:
        m.comm_init(n, r, grid, "local", gid)
    m.read_atoms(x, v, lo, hi)
    m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    m.setup(); m.run(50); m.sync()
    bar.wait(); t0 = time.time()
    m.run(steps); m.sync()
    bar.wait(); t1 = time.time()
    if r == 0:
        m.set_option("profile", 1); m.run(50); m.sync()
        out["timers"] = {k: m.timer(k) for k in ("pair", "neigh", "reorder", "bin", "halo", "migrate", "merge", "nve")}
        m.set_option("profile", 0)
    else:
        m.run(50); m.sync()
    out[r] = (t1 - t0, m.counts())
    m.close()

th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(n)]
[t.start() for t in th]
[t.join(300) for t in th]
dt = max(out[r][0] for r in range(n))
print("%d ranks %s on %d^3: %.0f steps/s whole job (%.1f us/step), counts rank 0 %s" % (n, grid, L, steps / dt, dt / steps * 1e6, out[0][1]))
print("rank 0 phase averages (us per call, calls):", {k: (round(v[0] / max(v[1], 1) * 1e3, 1), v[1]) for k, v in out.get("timers", {}).items()})
