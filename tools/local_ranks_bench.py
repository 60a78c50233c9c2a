"""Work inflation of the spatial decomposition, measured on ONE GPU: N engine contexts (threads, LOCAL transport) share
the card, so the wall time of the N-rank run is roughly the SUM of the ranks' GPU work plus host overhead.  The ratio
N * T(1 rank) / T(N ranks) bounds the strong-scaling speed-up N GPUs could reach before any exchange latency."""
import sys, threading, time
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box

L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
x, v, lo, hi = make_box(L)


def run(nranks, grid):
    gid = np.frombuffer(np.random.default_rng(nranks).bytes(8), np.uint8)
    bar = threading.Barrier(nranks)
    times = [0.0] * nranks

    def work(r):
        m = Meso()
        if nranks > 1:
            m.comm_init(nranks, r, grid, "local", gid)
        m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
        m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
        m.setup(); m.run(50); m.sync()
        bar.wait()
        t0 = time.perf_counter()
        m.run(steps); m.sync()
        bar.wait()
        times[r] = time.perf_counter() - t0
        m.close()

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]; [t.join() for t in th]
    return max(times)


t1 = run(1, (1, 1, 1))
print("1 rank : %.1f us/step" % (t1 / steps * 1e6))
for n, g in ((2, (2, 1, 1)), (4, (2, 2, 1)), (8, (2, 2, 2))):
    tn = run(n, g)
    print("%d ranks: %.1f us/step on one GPU -> per-rank work %.1f us/step, speed-up bound %.2fx" % (n, tn / steps * 1e6, tn / steps * 1e6 / n, n * t1 / tn))
