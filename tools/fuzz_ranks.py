#!/usr/bin/env python3
"""tools/fuzz_ranks.py [NCASES] [SEED]: random decks (box shape, density, types and masses, style) on a random processor grid of in-process
ranks (LOCAL transport, all on the one GPU) against the same deck on one rank: setup forces of every atom equal to the rounding of the
fp32 merged coordinates (each rank centres them on its own sub-box), and after 23 steps with migration every tag exists exactly once, all
values are finite and the temperature is the one-rank run's to a few per cent.  (Trajectories themselves differ between processor grids
once the thermostat is on: the TEA signature hashes the top mantissa bits of the fp32 velocity - the reference's design.)"""
import sys, threading
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa: F401
from meso_amd.api import Meso

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
GRIDS = [(2, 1, 1), (1, 2, 1), (1, 1, 2), (2, 2, 1), (2, 1, 2), (1, 2, 2), (2, 2, 2), (4, 1, 1), (1, 1, 4)]
bad = 0


def run(nranks, grid, deck, style):
    x, v, dims, types, ntypes, masses, every = deck
    gid = np.frombuffer(np.random.default_rng(nranks * 7919 + len(x)).bytes(8), np.uint8)
    out, errs = [None] * nranks, []

    def work(r):
        try:
            m = Meso()
            if nranks > 1:
                m.comm_init(nranks, r, grid, "local", gid)
            m.read_atoms(x, v, np.zeros(3), dims, types=types, ntypes=ntypes, masses=masses)
            m.neighbor(0.3); m.neigh_modify(delay=0, every=every, check=False)
            m.pair_style(style, 1.0, 419084618)
            for i in range(1, ntypes + 1):
                for j in range(i, ntypes + 1):
                    m.pair_coeff(i, j, 15.0 if i == j else 30.0, 4.5, 3.0, 1.0, 1.0)
            m.timestep(0.004); m.setup()
            f0 = m.gather(by_tag=False)
            m.run(23)
            out[r] = (f0, m.gather(by_tag=False), m.temperature())
            m.close()
        except Exception as e:      # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(240) for t in th]
    if errs or any(o is None for o in out):
        return None, errs

    def merge(idx):
        cols = [np.concatenate([o[idx][k] for o in out]) for k in range(4)]
        order = np.argsort(cols[3], kind="stable")
        return [c[order] for c in cols]
    return (merge(0), merge(1), [o[2] for o in out]), None


for case in range(ncases):
    grid = GRIDS[int(rng.integers(len(GRIDS)))]
    nranks = grid[0] * grid[1] * grid[2]
    # (every sub-box at least two ghost cutoffs wide plus a bin: 4 per rank and dimension is plenty)
    dims = np.array([g * (4.0 + 6.0 * rng.random()) for g in grid]) + rng.integers(0, 6, 3)
    rho = float(rng.choice([3.0, 4.0]))
    n = int(rho * dims.prod())
    if n > 120000:
        continue
    x = rng.random((n, 3)) * dims
    v = rng.random((n, 3)) - 0.5; v -= v.mean(0); v *= np.sqrt(1.0 / ((v * v).sum() / (3 * n - 3)))
    ntypes = int(rng.integers(1, 3))
    types = rng.integers(1, ntypes + 1, n).astype(np.int32)
    masses = np.concatenate([[0.0], 0.5 + 2.0 * rng.random(ntypes)])
    every = int(rng.choice([1, 5]))
    style = str(rng.choice(["dpd/meso", "dpd/fast/meso"]))
    deck = (x, v, dims, types, ntypes, masses, every)
    one, e1 = run(1, (1, 1, 1), deck, style)
    many, e2 = run(nranks, grid, deck, style)
    ok, why = True, ""
    if one is None or many is None:
        ok, why = False, "a run failed: %s %s" % (e1, e2)
    else:
        f1, fm = one[0][2], many[0][2]
        tol = (2e-3 if style != "dpd/meso" else 2e-4) * np.abs(f1).max()      # fp32 merged coordinates, centred per sub-box
        tags = many[1][3]
        if not np.array_equal(tags, np.arange(1, n + 1)): ok, why = False, "tags lost or doubled"
        elif np.abs(f1 - fm).max() > tol: ok, why = False, "setup forces differ by %.3g (tol %.3g)" % (np.abs(f1 - fm).max(), tol)
        elif not np.isfinite(many[1][0]).all(): ok, why = False, "not finite"
        elif abs(many[2][0] - one[2][0]) > 0.05: ok, why = False, "temperature %.4f against %.4f" % (many[2][0], one[2][0])
    print("case %2d  box %5.1f x %5.1f x %5.1f  rho %.0f  n %6d  grid %s  types %d  every %d  %-13s  %s" % (
        case, dims[0], dims[1], dims[2], rho, n, grid, ntypes, every, style, "ok" if ok else "FAILED: " + why), flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
