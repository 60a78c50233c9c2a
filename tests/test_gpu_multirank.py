"""Spatial decomposition on ONE GPU: several ranks (one engine context + one host thread each) exchange
migrants and ghosts through the in-process LOCAL transport, which runs the same device pack / unpack /
scatter kernels and the same per-peer message schedule as the RCCL transport (only the copy primitive
differs).  Decomposition invariance is the property the reference's TEA signatures are designed for
(SURVEY.md 4 vi): forces and trajectories must not depend on the processor grid."""
import threading

import numpy as np
import torch  # noqa: F401  (before libmeso_hip.so is loaded: torch ships its own HIP runtime, and the one loaded first serves both)
from conftest import join_ranks
import pytest

from meso_amd.datagen import make_box

pytestmark = pytest.mark.gpu


def _run_ranks(nranks, grid, L, style, sigma, steps, every=5, overlap=1, opts=()):
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(L)
    gid = np.frombuffer(np.random.default_rng(nranks * 1000 + L).bytes(8), np.uint8)
    out, errs = [None] * nranks, []

    def work(r):
        try:
            m = Meso()
            m.set_option("overlap", overlap)
            for k, val in opts:
                m.set_option(k, val)
            if nranks > 1:
                m.comm_init(nranks, r, grid, "local", gid)
            m.read_atoms(x, v, lo, hi)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=every, check=False)
            m.pair_style(style, 1.0, 419084618)
            m.pair_coeff(1, 1, 15.0, 4.5, sigma, 1.0, 1.0)
            m.timestep(0.005)
            m.setup()
            f0 = m.gather(by_tag=False)
            m.run(steps)
            T = m.temperature()
            out[r] = (f0, m.gather(by_tag=False), m.counts(), T)
            m.close()
        except Exception as e:   # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
    [t.start() for t in th]
    join_ranks(th, errs, 300)
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"

    def merge(idx):
        cols = [np.concatenate([o[idx][k] for o in out]) for k in range(4)]
        order = np.argsort(cols[3], kind="stable")
        return [c[order] for c in cols]

    return merge(0), merge(1), [o[2] for o in out], [o[3] for o in out], (x, v, lo, hi)


@pytest.mark.parametrize("nranks,grid", [(2, (2, 1, 1)), (4, (2, 2, 1)), (8, (2, 2, 2)), (3, (1, 3, 1))])
def test_decomposition_invariance(nranks, grid):
    L = 12 if nranks != 3 else 13
    ref0, ref1, _, Tref, (x, v, lo, hi) = _run_ranks(1, (1, 1, 1), L, "dpd/meso", 3.0, 1)
    got0, got1, counts, T, _ = _run_ranks(nranks, grid, L, "dpd/meso", 3.0, 1)
    n = len(x)
    assert sum(c[0] for c in counts) == n                        # every atom owned exactly once
    assert np.array_equal(got0[3], np.arange(1, n + 1)) and np.array_equal(got1[3], np.arange(1, n + 1))
    # merged coordinates are recentred on each rank's own sub-box (atom_vec_meso.cu:154-156), so fp32 rounding
    # differs between processor grids: forces agree to the fp32-coordinate tolerance, not bit for bit
    scale = np.abs(ref0[2]).max()
    assert np.abs(got0[2] - ref0[2]).max() < 5e-6 * scale        # setup forces (thermostat on: same TEA numbers)
    prd = hi - lo
    d = got1[0] - ref1[0]
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < 1e-7                                # positions after one step: step-0 forces only
    # the step-1 random forces are keyed on the top 11 mantissa bits of each fp32 velocity (math_meso.h:436-442):
    # a 1e-8 velocity difference re-keys a few particles (and their partners), everything else is bit-close
    dv = np.abs(got1[1] - ref1[1]).max(axis=1)
    assert (dv > 1e-5).mean() < 0.05 and np.median(dv) < 1e-7
    assert all(abs(t - T[0]) < 1e-12 for t in T) and abs(T[0] - Tref[0]) < 1e-3   # one global value on every rank


@pytest.mark.parametrize("style", ["dpd/meso", "dpd/fast/meso"])
def test_migration_and_rebuilds_conserve_atoms(style):
    """60 steps on a 2x2x2 grid: atoms cross sub-domain faces, edges and corners; none is lost or duplicated,
    momentum stays zero and the thermostat holds the same temperature as the single-rank run."""
    L = 12
    _, ref1, _, Tref, (x, v, lo, hi) = _run_ranks(1, (1, 1, 1), L, style, 3.0, 60)
    _, got1, counts, T, _ = _run_ranks(8, (2, 2, 2), L, style, 3.0, 60)
    n = len(x)
    assert sum(c[0] for c in counts) == n
    assert np.array_equal(got1[3], np.arange(1, n + 1))
    assert len({c[0] for c in counts}) > 1                       # populations drifted: migration really happened
    assert np.abs(got1[1].sum(0)).max() < 1e-2
    assert abs(T[0] - Tref[0]) < 0.08 and all(abs(t - T[0]) < 1e-9 for t in T)


@pytest.mark.parametrize("style,L", [("dpd/meso", 12), ("dpd/fast/meso", 16)])
def test_overlapped_refresh_is_bit_identical(style, L):
    """bulk kernel || ghost exchange on the side stream, then border kernel == one kernel after the exchange.
    fp32 style: the ring kernel with its step-boundary epilogue and workgroup Newton pairing runs in both parts (the
    split point is a multiple of its 256-atom groups; fixed-point sums do not depend on the launch structure)."""
    a = _run_ranks(4, (2, 2, 1), L, style, 3.0, 12, overlap=0)[1]
    b = _run_ranks(4, (2, 2, 1), L, style, 3.0, 12, overlap=1)[1]
    for u, w in zip(a[:3], b[:3]):
        assert np.array_equal(u, w)


@pytest.mark.parametrize("style,tol", [("dpd/meso", 2e-6), ("dpd/fast/meso", 2e-4)])
def test_sigma0_trajectory_is_grid_independent(style, tol):
    L = 12
    _, ref1, _, _, (x, v, lo, hi) = _run_ranks(1, (1, 1, 1), L, style, 0.0, 20)
    _, got1, _, _, _ = _run_ranks(8, (2, 2, 2), L, style, 0.0, 20)
    prd = hi - lo
    d = got1[0] - ref1[0]
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < tol and np.abs(got1[1] - ref1[1]).max() < 10 * tol


def test_rccl_communicator_next_to_torch_distributed():
    """bench.py's multi-GPU path in miniature on one GPU: torch.distributed (RCCL) is initialised first, then the engine
    obtains an ncclUniqueId and creates ITS OWN communicator in the same process (world size 1) and runs a few steps.
    Checks that the library's RCCL and the one torch loaded coexist; the N > 1 exchange itself runs over the LOCAL
    transport in the tests above (the box has one GPU)."""
    import os
    import torch
    import torch.distributed as dist
    from meso_amd.api import Meso, nccl_unique_id
    if not dist.is_nccl_available():
        pytest.skip("torch built without RCCL")
    os.environ["MESO_FORCE_RCCL"] = "1"
    try:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)
        uid = torch.from_numpy(nccl_unique_id().copy()).cuda()
        dist.broadcast(uid, 0)
        x, v, lo, hi = make_box(8)
        m = Meso(0)
        m.comm_init(1, 0, (1, 1, 1), "rccl", uid.cpu().numpy())
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)
        m.pair_style("dpd/fast/meso", 1.0, 419084618)
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.005)
        m.setup()
        m.run(10)
        assert 0.5 < m.temperature() < 2.0
        m.close()
    finally:
        os.environ.pop("MESO_FORCE_RCCL", None)
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("every", [1, 3, 5])
def test_borders_without_host_round_trip_equal_the_synchronous_path(every):
    """Several ranks, option async_counts (default): from the second rebuild on the ghost stage sends fixed-capacity messages with
    their counts in a header and leaves every count on the device until the next ghost refresh needs it.  Same ghosts in the
    same order as the exact two-phase exchange: 23 steps on 2 x 2 x 2 ranks are bit-identical, for rebuilds on every step too."""
    a = _run_ranks(8, (2, 2, 2), 12, "dpd/fast/meso", 3.0, 23, every=every, opts=(("async_counts", 0),))
    b = _run_ranks(8, (2, 2, 2), 12, "dpd/fast/meso", 3.0, 23, every=every)
    for k in range(3):
        assert np.array_equal(a[1][k], b[1][k])
    assert a[2] == b[2]                                   # (nlocal, nghost, nsend) of every rank
    # migration messages carry their counts too (capacity 2 x previous count + 64); with the floor at 0 every message that follows
    # a rebuild without migrants to that peer does not fit and is sent again, exactly: still the same trajectory
    c = _run_ranks(8, (2, 2, 2), 12, "dpd/fast/meso", 3.0, 23, every=every, opts=(("mig_cap_floor", 0),))
    for k in range(3):
        assert np.array_equal(a[1][k], c[1][k])
    assert a[2] == c[2]
    # the receiving side in one kernel (ghosts stay in message order, the list builder reads the ghost cells as runs of equal cell
    # codes the sender wrote: default) against the unpack + count + scan + place + order + merge chain (border_runs 0)
    d = _run_ranks(8, (2, 2, 2), 12, "dpd/fast/meso", 3.0, 23, every=every, opts=(("border_runs", 0),))
    for k in range(3):
        assert np.array_equal(a[1][k], d[1][k])
    assert a[2] == d[2]
    # the migration's sending side in two launches (leavers appended to their direction's list by atomics, ranked when packed:
    # default) against the count / scan / fill chain (mig_slim 0): the same messages in the same order; likewise the border lists,
    # headers and records written by one kernel (default) against fill + header + pack (border_fused 0), and the per-step ghost
    # refresh received straight into the merged arrays (default) against exchange + scatter kernel (refresh_direct 0) and written by
    # the force kernel's step boundary (default) against k_pack_forward_multi (refresh_epilogue 0)
    e = _run_ranks(8, (2, 2, 2), 12, "dpd/fast/meso", 3.0, 23, every=every, opts=(("mig_slim", 0), ("border_fused", 0), ("refresh_direct", 0), ("refresh_epilogue", 0)))
    for k in range(3):
        assert np.array_equal(a[1][k], e[1][k])
    assert a[2] == e[2]


def test_border_message_capacity_is_checked():
    """A border message that outgrows the capacity both ranks derived from the previous rebuild must end the run with an error,
    not with a truncated ghost list: with a negative margin (capacity = 60 % of the previous count) every rank reports it."""
    from meso_amd.api import Meso, MesoError
    x, v, lo, hi = make_box(12)
    gid = np.frombuffer(np.random.default_rng(424242).bytes(8), np.uint8)
    errs = [None, None]

    def work(r):
        m = Meso()
        m.set_option("mr_cap_margin", -0.4)
        m.comm_init(2, r, (2, 1, 1), "local", gid)
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
        m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
        try:
            m.setup()
            m.run(7)
        except MesoError as e:
            errs[r] = str(e)
        m.close()

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in th]
    join_ranks(th, None, 120)
    assert all(e and "capacity" in e for e in errs), errs


@pytest.mark.parametrize("style,tol", [("dpd/meso", 1e-10), ("dpd/fast/meso", 1e-5)])
def test_two_section_rows_on_eight_ranks(style, tol):
    """Rows in two sections (round 5) under the bulk / border split of several ranks: every rank partitions for its own pairing group
    and both of its force launches pair inside it.  sigma = 0 (no thermostat to amplify the last-bit difference of setup's
    lane-per-atom forces): 12 steps with two rebuilds and migration equal the run on plain rows, and no atom is lost."""
    a0, a1, ca, _, (x, v, lo, hi) = _run_ranks(8, (2, 2, 2), 12, style, 0.0, 12, opts=(("row_part", 1),))
    b0, b1, cb, _, _ = _run_ranks(8, (2, 2, 2), 12, style, 0.0, 12, opts=(("row_part", 0),))
    n = len(x)
    assert np.array_equal(a1[3], np.arange(1, n + 1)) and ca == cb
    scale = np.abs(b0[2]).max()
    assert np.abs(a0[2] - b0[2]).max() < 1e-5 * scale
    prd = hi - lo
    d = a1[0] - b1[0]
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < tol and np.abs(a1[1] - b1[1]).max() < 100 * tol
