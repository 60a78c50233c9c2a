#!/usr/bin/env python3
"""tools/fuzz_paths.py [NCASES] [SEED] [MIN_EDGE] [MAX_EDGE] [MAX_ATOMS]: random decks (box shape, density, periodicity, types and masses, rebuild interval, style) run 23
steps through the default path and through the plain one (every fusion of the rebuild and of the step boundary switched off): positions,
velocities and forces must be equal bit for bit (both start from the ring kernel's forces, whose sums do not depend on entry order); and
a third time through the default path with an outgrown ghost capacity planted at a random step (option debug_ghost_cap: the rebuild that
reports it is redone inside run()) - the same bits again.
A check beyond the test-suite's fixed decks; prints one line per case and exits non-zero on the first difference."""
import sys
import numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
emin = int(sys.argv[3]) if len(sys.argv) > 3 else 4
emax = int(sys.argv[4]) if len(sys.argv) > 4 else 28
nmax = int(sys.argv[5]) if len(sys.argv) > 5 else 150000
PLAIN = (("fuse_count", 0), ("merge_ghosts", 0), ("lean_boundary", 0), ("fused_rebuild", 0), ("ghost_epilogue", 0), ("xcd_balance", 0),
         ("row_part", 0), ("async_counts", 0))
bad = 0
for case in range(ncases):
    dims = rng.integers(emin, emax, 3).astype(float) + rng.random(3)
    rho = float(rng.choice([3.0, 4.0, 6.0]))
    n = int(rho * dims.prod())
    if n > nmax:
        dims *= (nmax / n) ** (1 / 3); n = int(rho * dims.prod())
    per = tuple(int(p) for p in rng.integers(0, 2, 3)) if rng.random() < 0.4 else (1, 1, 1)
    x = rng.random((n, 3)) * dims
    # (a layer next to a non-periodic face stays empty: atoms must not leave the box)
    for d in range(3):
        if not per[d]:
            x[:, d] = 1.0 + x[:, d] * (dims[d] - 2.0) / dims[d]
    v = rng.random((n, 3)) - 0.5; v -= v.mean(0); v *= np.sqrt(1.0 / ((v * v).sum() / (3 * n - 3)))
    ntypes = int(rng.integers(1, 4))
    types = rng.integers(1, ntypes + 1, n).astype(np.int32)
    masses = np.concatenate([[0.0], 0.5 + 2.0 * rng.random(ntypes)])
    every = int(rng.choice([1, 2, 5, 7]))
    style = str(rng.choice(["dpd/meso", "dpd/fast/meso"]))
    res = []
    # third run (round 6): the default path with an outgrown ghost capacity planted at a random step - the rebuild is redone inside run()
    k_plant = int(rng.integers(1, 20))
    for opts in ((), PLAIN, "planted"):
        planted = opts == "planted"
        if planted:
            opts = ()
        m = Meso()
        for k, val in opts:
            m.set_option(k, val)
        m.read_atoms(x, v, np.zeros(3), dims, types=types, ntypes=ntypes, masses=masses, periodicity=per)
        m.neighbor(0.3); m.neigh_modify(delay=0, every=every, check=False)
        m.pair_style(style, 1.0, 419084618)
        for i in range(1, ntypes + 1):
            for j in range(i, ntypes + 1):
                m.pair_coeff(i, j, 15.0 if i == j else 30.0, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.004); m.setup()
        m.force_clear(); m.compute(0, 0)
        if planted:
            m.run(k_plant)
            m.set_option("debug_ghost_cap", 3)
            m.run(23 - k_plant)
        else:
            m.run(23)
        res.append(m.gather()[:3]); m.close()
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1])) and all(np.array_equal(a, b) for a, b in zip(res[0], res[2]))
    fin = bool(np.isfinite(res[0][0]).all())
    print("case %2d  box %5.1f x %5.1f x %5.1f  rho %.0f  n %6d  per %s  types %d  every %d  %-13s  %s" % (
        case, dims[0], dims[1], dims[2], rho, n, per, ntypes, every, style, "equal" if same and fin else "DIFFERENT"), flush=True)
    bad += not (same and fin)
sys.exit(1 if bad else 0)
