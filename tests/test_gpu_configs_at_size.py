"""BASELINE.json configs[3] and configs[4] at their own sizes, on the one GPU of the test box.

configs[3] = 64^3 rho=4, pair_style dpd/meso (fp64), 8 ranks of a 2x2x2 brick decomposition.  RCCL needs one GPU per rank,
so the 8 ranks are 8 engine contexts of this process on the in-process LOCAL transport: the same device kernels (border
lists, pack, unpack, scatter), the same per-peer message schedule, only the copy primitive differs (comm.hip xchg).
configs[4] = 128^3 rho=4 (8.4 M beads), pair_style dpd/fast/meso, 10 % of the beads in bonded A2B4 chains - on one rank
and over 8 LOCAL ranks.  At these sizes the oracle is out of reach (minutes per step), so the checks are the
size-independent properties of the domain; the same code paths are compared with the oracle at L = 12 below and in
test_gpu_multirank.py / test_gpu_bonds.py."""
import threading

import numpy as np
from conftest import join_ranks
import pytest

from meso_amd.datagen import make_box, make_polymer_box

pytestmark = pytest.mark.gpu

A = {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}


def _ranks(nranks, grid, deck, style, steps, sigma=3.0, want=("setup", "end"), split=False, timeout=900):
    """deck = (x, v, lo, hi) or (x, v, types, bonds, lo, hi).  Returns per-phase tag-ordered (x, v, f), counts, T, info.
    split: each rank is handed only the atoms of its own sub-box (what the LAMMPS glue does) instead of the whole deck."""
    from meso_amd.api import Meso
    poly = len(deck) == 6
    gid = np.frombuffer(np.random.default_rng(nranks * 7919 + len(deck[0])).bytes(8), np.uint8)
    out, errs = [None] * nranks, []

    def work(r):
        try:
            m = Meso()
            if nranks > 1:
                m.comm_init(nranks, r, grid, "local", gid)
            if poly:
                x, v, types, bonds, lo, hi = deck
                m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
                m.special_bonds(0.0, 1.0, 1.0)
                m.read_bonds(bonds)
                m.bond_style("harmonic/meso", 1)
                m.bond_coeff(1, 50.0, 0.5)
            else:
                x, v, lo, hi = deck
                if split:
                    loc = np.array([r % grid[0], (r // grid[0]) % grid[1], r // (grid[0] * grid[1])])
                    w = (hi - lo) / np.array(grid)
                    cell = np.minimum(((x - lo) / w).astype(int), np.array(grid) - 1)
                    mine = (cell == loc).all(axis=1)
                    m.read_atoms(x[mine], v[mine], lo, hi, tags=np.nonzero(mine)[0].astype(np.int32) + 1, ntypes=1)
                else:
                    m.read_atoms(x, v, lo, hi)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=5, check=False)
            m.pair_style(style, 1.0, 419084618)
            if poly:
                for (i, j), a in A.items():
                    m.pair_coeff(i, j, a, 4.5, sigma, 1.0, 1.0)
            else:
                m.pair_coeff(1, 1, 15.0, 4.5, sigma, 1.0, 1.0)
            m.timestep(0.005)
            m.setup()
            res = {"info": m.neigh_info()}
            if "setup" in want:
                res["setup"] = m.gather(by_tag=False)
            if steps:
                m.run(steps)
            res["end"] = m.gather(by_tag=False)
            res["counts"] = m.counts()
            res["T"] = m.temperature()
            out[r] = res
            m.close()
        except Exception as e:   # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
    [t.start() for t in th]
    join_ranks(th, errs, timeout)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"

    def merge(key):
        cols = [np.concatenate([o[key][k] for o in out]) for k in range(5)]
        order = np.argsort(cols[3], kind="stable")
        return [c[order] for c in cols]

    return {k: merge(k) for k in want if k in out[0]}, [o["counts"] for o in out], [o["T"] for o in out], [o["info"] for o in out]


def test_config3_64cube_fp64_on_8_ranks():
    """configs[3] at size: forces of the 8-rank decomposition equal the 1-rank forces (TEA makes the random force
    decomposition-independent), no net force, and 20 steps with 4 rebuilds/migrations keep every tag exactly once."""
    deck = make_box(64)
    n = len(deck[0])
    one, _, T1, info1 = _ranks(1, (1, 1, 1), deck, "dpd/meso", 0, want=("setup",))
    got, counts, T, info = _ranks(8, (2, 2, 2), deck, "dpd/meso", 20)
    assert n == 4 * 64 ** 3 and sum(c[0] for c in counts) == n
    assert min(c[0] for c in counts) > 0.95 * n / 8 and max(c[1] for c in counts) < 0.45 * n / 8     # ghost shells of 32^3 sub-boxes
    f1, f8 = one["setup"][2], got["setup"][2]
    scale = np.abs(f1).max()
    assert np.array_equal(got["setup"][3], np.arange(1, n + 1))
    # fp32 merged coordinates are centred on each rank's own sub-box (atom_vec_meso.cu:154-156): the operands of a pair
    # differ by an ulp of the coordinate, 2^-18 at |x| = 32 (one rank) against 2^-19 at 16 (eight ranks) - 8x the ulp of the
    # L = 12 decks, whose 5e-6 tolerance scales accordingly; most components agree far better
    err = np.abs(f8 - f1)
    assert err.max() < 4e-5 * scale and np.quantile(err, 0.999) < 1e-5 * scale and np.median(err) < 1e-6 * scale
    assert np.abs(f8.sum(axis=0)).max() < 1e-6 * scale * np.sqrt(n)
    # (a handful of pairs at the 1.3 list cutoff fall on the other side in the other frame's fp32 rounding)
    assert abs(sum(i["avg_count"] * c[0] for i, c in zip(info, counts)) / n - info1[0]["avg_count"]) < 1e-5
    # after 20 steps: every atom still owned exactly once, momentum conserved, thermostat sane, one global T on all ranks
    end = got["end"]
    assert np.array_equal(end[3], np.arange(1, n + 1))
    assert np.abs(end[1].sum(axis=0)).max() < 1e-6 * n
    assert all(abs(t - T[0]) < 1e-12 for t in T) and 0.9 < T[0] < 1.6
    assert np.isfinite(end[0]).all() and np.isfinite(end[2]).all()


@pytest.mark.parametrize("nranks,grid", [(1, (1, 1, 1)), (8, (2, 2, 2))])
def test_config4_128cube_polymer(nranks, grid):
    """configs[4] at size (8 388 608 beads, 139 810 A2B4 chains): atoms, bonds and chain geometry are conserved through
    rebuilds (and migration on 8 ranks), the list holds the expected ~35.9 - exclusions neighbours, net force ~ 0."""
    deck = make_polymer_box(128, frac=0.1)
    x, v, types, bonds, lo, hi = deck
    n = len(x)
    got, counts, T, info = _ranks(nranks, grid, deck, "dpd/fast/meso", 10, want=("setup", "end"))
    assert n == 4 * 128 ** 3 and sum(c[0] for c in counts) == n
    f0 = got["setup"][2]
    scale = np.abs(f0).max()
    assert np.array_equal(got["setup"][3], np.arange(1, n + 1)) and np.array_equal(got["setup"][4], types)
    assert np.abs(f0.sum(axis=0)).max() < 2e-4 * scale * np.sqrt(n)                   # fp32 pair arithmetic
    nbar = sum(i["avg_count"] * c[0] for i, c in zip(info, counts)) / n
    # uniformly random solvent positions: rho 4/3 pi 1.3^3 = 36.8 list partners; the chain beads sit within 0.5 of their
    # bonded neighbours (a few more partners each), whose 1-2 entries special_bonds 0 1 1 removes again: 37.1 measured
    assert 36.6 < nbar < 37.6
    assert max(i["max_count"] for i in info) <= info[0]["n_col"]
    end = got["end"]
    assert np.array_equal(end[3], np.arange(1, n + 1)) and np.array_equal(end[4], types)
    assert np.abs(end[1].sum(axis=0)).max() < 1e-5 * n
    assert all(abs(t - T[0]) < 1e-12 for t in T) and 0.8 < T[0] < 2.0
    # bonds stay bonded: every bonded pair within a few r0 after 10 steps (minimum image)
    d = end[0][bonds[:, 0] - 1] - end[0][bonds[:, 1] - 1]
    d -= np.round(d / (hi - lo)) * (hi - lo)
    r = np.sqrt((d * d).sum(axis=1))
    assert r.max() < 1.3 and 0.3 < r.mean() < 0.8


@pytest.mark.parametrize("style,tol", [("dpd/meso", 5e-6), ("dpd/fast/meso", 2e-3)])
def test_8_ranks_against_the_oracle(oracle, style, tol):
    """The N-rank path compared with the CPU ORACLE (not only with the 1-rank HIP run): setup forces with the thermostat on,
    and - fp64 style, sigma = 0 - positions after 10 steps with two rebuilds and migration."""
    from oracle.meso_sim import MesoRefSim
    deck = make_box(12)
    x, v, lo, hi = deck
    s = MesoRefSim(x, v, lo, hi, fast=style != "dpd/meso")
    s.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
    s.setup()
    got, counts, _, _ = _ranks(8, (2, 2, 2), deck, style, 0, want=("setup",))
    assert sum(c[0] for c in counts) == len(x)
    assert np.abs(got["setup"][2] - s.f).max() < tol * np.abs(s.f).max()
    if style == "dpd/meso":
        s0 = MesoRefSim(x, v, lo, hi)
        s0.pair_coeff(1, 1, 15.0, 4.5, 0.0, 1.0, 1.0)
        s0.setup()
        s0.run(10)
        got, _, _, _ = _ranks(8, (2, 2, 2), deck, style, 10, sigma=0.0, want=("end",))
        d = got["end"][0] - s0.x
        d -= np.round(d / (hi - lo)) * (hi - lo)
        assert np.abs(d).max() < 2e-6 and np.abs(got["end"][1] - s0.v).max() < 2e-5


def test_sparse_rank_grows_during_the_border_stage():
    """A rank that starts empty allocates the minimum capacity; its first ghost shell is larger than that, so the arrays
    (and the border-list scratch) are regrown in the middle of halo_borders_multi.  Half-filled box, 2 ranks along x."""
    L = np.array([8.0, 40.0, 40.0])
    rng = np.random.default_rng(77)
    n = int(4 * 4 * 40 * 40)
    x = rng.random((n, 3)) * np.array([3.999, 40.0, 40.0])
    v = rng.normal(size=(n, 3)) * 0.1
    v -= v.mean(axis=0)
    deck = (x, v, np.zeros(3), L)
    one, _, _, _ = _ranks(1, (1, 1, 1), deck, "dpd/meso", 0, want=("setup",))
    got, counts, _, _ = _ranks(2, (2, 1, 1), deck, "dpd/meso", 5, want=("setup", "end"), split=True, timeout=120)
    assert min(c[0] for c in counts) < 2000                       # (after 5 steps a few atoms have crossed into the empty half)
    f1, f2 = one["setup"][2], got["setup"][2]
    assert np.array_equal(got["setup"][3], np.arange(1, n + 1))
    assert np.abs(f2 - f1).max() < 5e-6 * np.abs(f1).max()
    assert np.array_equal(got["end"][3], np.arange(1, n + 1))


@pytest.mark.parametrize("style,tol_f", [("dpd/fast/meso", 6e-3), ("dpd/meso", 6e-3)])
def test_config2_64cube_against_the_pinned_cpu_restatement(oracle, style, tol_f):
    """configs[2] (and the one-rank half of configs[3]) at size against the restatement of the reference's stock CPU path that
    oracle/_ref pins bit for bit: sigma = 0 (the deterministic part of the force), 64^3 = 1 048 576 atoms - forces, neighbour
    count, potential energy, pressure and the temperature after 10 steps with two rebuilds.  The GPU path (like the
    reference's) works on fp32 coordinates relative to the box centre (atom_vec_meso.cu:154-156): their ulp at |x| = 32 is
    3.8e-6, 8x that of the L = 8 comparison (tolerance 2e-4 there), and the maximum is taken over 3 M components instead of
    6 k: largest force difference 3.2e-3 on |F| ~ 200, median 5e-5."""
    import os
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(64)
    n = len(x)
    s = oracle.LmpDpd(x, lo, hi, nthreads=min(16, len(os.sched_getaffinity(0))))
    s.pair_style(0.0, 1.0, 419084618)
    s.pair_coeff(1, 1, 15.0, 4.5)
    s.set_velocities(v)
    s.neighbor(0.3, 5, 0)
    s.timestep(0.005)
    s.setup()
    with Meso() as m:
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)
        m.pair_style(style, 1.0, 419084618)
        m.pair_coeff(1, 1, 15.0, 4.5, 0.0, 1.0, 1.0)
        m.timestep(0.005)
        m.setup()
        f = m.gather()[2]
        fs = s.state()[2]
        assert np.abs(f - fs).max() < tol_f and np.median(np.abs(f - fs)) < 1e-4
        assert m.neigh_info()["avg_count"] == pytest.approx(2.0 * s.nneigh / n, abs=64.0 / n)
        assert m.pe() / n == pytest.approx(s.pe_per_atom, rel=5e-6)
        assert m.pressure() == pytest.approx(s.pressure, rel=5e-6)
        m.run(10)
        s.run(10)
        xg, vg = m.gather()[:2]
        xs, vs, _ = s.state()
        d = xg - xs
        d -= np.round(d / (hi - lo)) * (hi - lo)
        assert np.abs(d).max() < 2e-5 and np.abs(vg - vs).max() < 4e-4
        assert m.temperature() == pytest.approx(s.temperature, rel=1e-6)


def test_config1_25cube_fp64_rebuild_every_step(oracle):
    """configs[1] as written: 25^3 rho=4 (62 500 atoms), pair_style dpd/meso (fp64 arithmetic), neighbour rebuild EVERY step,
    20 steps, sigma = 0, against the restatement of the reference's stock CPU path that oracle/_ref pins bit for bit
    (same rebuild cadence on both sides): positions, velocities, forces and temperature after the 20 steps."""
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(25)
    n = len(x)
    s = oracle.LmpDpd(x, lo, hi, nthreads=min(8, len(__import__("os").sched_getaffinity(0))))
    s.pair_style(0.0, 1.0, 419084618)
    s.pair_coeff(1, 1, 15.0, 4.5)
    s.set_velocities(v)
    s.neighbor(0.3, 1, 0)
    s.timestep(0.005)
    s.setup()
    s.run(20)
    with Meso() as m:
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=1, check=False)
        m.pair_style("dpd/meso", 1.0, 419084618)
        m.pair_coeff(1, 1, 15.0, 4.5, 0.0, 1.0, 1.0)
        m.timestep(0.005)
        m.setup()
        m.run(20)
        assert m.neigh_info()["nbuild"] == 20
        xg, vg, fg = m.gather()[:3]
        xs, vs, fs = s.state()
        d = xg - xs
        d -= np.round(d / (hi - lo)) * (hi - lo)
        # fp32 merged coordinates at |x| <= 12.5 (ulp 9.5e-7): the tolerances of the L = 8 comparison scaled by the coordinate range
        assert np.abs(d).max() < 5e-6 and np.abs(vg - vs).max() < 2e-4
        assert np.abs(fg - fs).max() < 2e-3 and np.median(np.abs(fg - fs)) < 5e-5
        assert m.temperature() == pytest.approx(s.temperature, rel=1e-6)
        assert n == 62500


@pytest.mark.skipif(not __import__("os").environ.get("MESO_TEST_SLOW"), reason="24 s for a size no BASELINE config names: MESO_TEST_SLOW=1 runs it")
def test_256cube_on_one_gpu_beyond_2_25_atoms():
    """67 108 864 atoms (+ 2.4 M ghosts) on ONE MI355X (about 90 GB of its 288 GB): more atoms than the 25-bit index of the
    force kernel's record word can name, so the launcher switches to the wide records (whole 32-bit index, owner lane and
    pairing flag in a byte ring; bit-identical forces at small sizes: test_lanes_per_atom_give_identical_forces).  Size-independent
    properties: every pair force has its opposite, the list holds the expected 35.8 entries per atom, 10 steps with two rebuilds
    conserve momentum and atom identities."""
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(256)
    n = len(x)
    assert n == 4 * 256 ** 3 and n > 2 ** 25
    with Meso() as m:
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)
        m.pair_style("dpd/fast/meso", 1.0, 419084618)
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.005)
        m.setup()
        info = m.neigh_info()
        assert abs(info["avg_count"] - 35.82) < 0.2 and info["max_count"] < 90
        f = m.gather()[2]
        scale = np.abs(f).max()
        assert 50 < scale < 1500 and np.abs(f.sum(0)).max() < 2e-4 * scale * np.sqrt(n)
        del f
        m.run(10)
        xg, vg, fg, tag, typ = m.gather()
        assert np.array_equal(tag, np.arange(1, n + 1))
        assert np.abs(vg.sum(0)).max() < 1e-5 * n
        assert 0.8 < m.temperature() < 2.0


@pytest.mark.parametrize("style", ["dpd/fast/meso", "dpd/meso"])
def test_config2_64cube_two_section_rows_are_bit_identical_to_plain_rows(style):
    """configs[2] / the one-GPU leg of configs[3] at size: with the list builder's rows in two sections (round 5, RowPartArgs in
    kernels.h; XCD-balanced launch, ghost refresh in the epilogue) the ring kernel's forces and 12 steps with two rebuilds equal
    those of plain rows (row_part 0, one range of atoms per XCD, refresh kernel) bit for bit - the same pairs, summed as integers.
    The run starts from the ring kernel's forces (setup's come from the lane-per-atom kernel, whose per-thread sums follow the entry
    order)."""
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(64)
    res = []
    # (third and fourth variant - the rebuild at this size, more than 4096 tiles of cells: the count kernel on its own instead of in
    # the force kernel's epilogue, per-atom reads in the step boundary; the chain of small launches instead of the fused rebuild)
    for opts in ((("row_part", 1),), (("row_part", 0), ("xcd_balance", 0), ("ghost_epilogue", 0)),
                 (("row_part", 1), ("fuse_count", 0), ("lean_boundary", 0)), (("row_part", 1), ("fused_rebuild", 0)),
                 # tile_persist: the list builder as persistent workgroups drawing bricks from counters (round 6; measured 6 % slower
                 # than one workgroup per brick, default off) - the same table, entry for entry
                 (("row_part", 1), ("tile_persist", 1)), (("row_part", 0), ("tile_persist", 1))):
        with Meso() as m:
            for k, val in opts:
                m.set_option(k, val)
            m.read_atoms(x, v, lo, hi)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=5, check=False)
            m.pair_style(style, 1.0, 419084618)
            m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
            m.timestep(0.005)
            m.setup()
            assert m.neigh_parts()["parted"] == bool(opts[0][1])
            m.force_clear(); m.compute(0, 0)
            f0 = m.gather()[2]
            m.run(12)
            res.append((f0,) + tuple(m.gather()[:3]))
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a, b)
