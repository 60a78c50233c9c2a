"""Partitioned neighbour rows (round 5; RowPartArgs in meso_amd/csrc/kernels.h, option row_part, default on).

The list builder writes every row in two sections - front: the pairs the atom evaluates, back: in-group partners that evaluate the
pair themselves - and the ring kernel walks the front section only.  The reference keeps two sections per row as well ("core from the
row front, skin from the back", /root/reference/src/USER-MESO/neigh_build_meso.cu:91-115).  What must hold:

  * the neighbours of every atom are the ones of a plain table (sets), nothing twice, nothing lost;
  * of every pair inside one aligned pairing group exactly one atom has the other in its front section, the split is balanced, and the
    padding behind each section is the atom itself;
  * forces and trajectories are bit-identical to plain rows (row_part 0: pairing decided per entry from the two indices) - the same
    pairs are evaluated, once from one side or once from each, and the sums are integers.
"""
import numpy as np
import pytest

from conftest import DP_RUN
from meso_amd.datagen import make_box, make_polymer_box

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Meso():
    from meso_amd.api import Meso
    return Meso


def _start(Meso, x, v, lo, hi, style, opts=(), types=None, sigma=3.0, every=5):
    m = Meso()
    for k, val in opts:
        m.set_option(k, val)
    if types is None:
        m.read_atoms(x, v, lo, hi)
        pairs = {(1, 1): 15.0}
    else:
        m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
        pairs = {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=every, check=False)
    m.pair_style(style, 1.0, DP_RUN["seed"])
    for (i, j), a0 in pairs.items():
        m.pair_coeff(i, j, a0, 4.5, sigma, 1.0, 1.0)
    m.timestep(0.005)
    m.setup()
    return m


@pytest.mark.parametrize("L,npart", [(10, 0), (10, 1), (12, 4), (16, 0)])
def test_row_sections(Meso, L, npart):
    """structure of the stored rows against the rule, and the neighbour sets against a plain table"""
    x, v, lo, hi = make_box(L)
    opts = (("pair_npart", npart),) if npart else ()
    m = _start(Meso, x, v, lo, hi, "dpd/fast/meso", opts)
    nlocal = m.counts()[0]
    parts = m.neigh_parts(raw=True)
    count, table = m.neigh_table()
    m.close()
    m0 = _start(Meso, x, v, lo, hi, "dpd/fast/meso", opts + (("row_part", 0),))
    p0 = m0.neigh_parts()
    count0, table0 = m0.neigh_table()
    m0.close()
    assert parts["parted"] and not p0["parted"]
    group = parts["group"]
    assert group == (64 * 4 // npart if npart else (128 if nlocal <= 163840 else 256))
    assert np.array_equal(count, count0)
    nf, nb, fr, bk = parts["nfront"], parts["nback"], parts["front"], parts["back"]
    assert np.array_equal(nf + nb, count) and not p0["nback"].any() and np.array_equal(p0["nfront"], count0)
    nfront_in = nback = 0
    front_pairs = set()
    for i in range(nlocal):
        assert set(table[i, :count[i]]) == set(table0[i, :count0[i]]) and len(set(table[i, :count[i]])) == count[i]
        front, back = fr[i, :nf[i]], bk[i, :nb[i]]
        assert (fr[i, nf[i]:(nf[i] + 7) & ~7] == i).all() and (bk[i, nb[i]:(nb[i] + 7) & ~7] == i).all()
        assert not (front == i).any() and not (back == i).any()
        assert np.array_equal(np.concatenate([front, back]), table[i, :count[i]])
        # mine(i, j): in-group pair that atom i evaluates for both (pairing inside aligned groups that lie wholly below nlocal)
        full = (i | (group - 1)) < nlocal
        for j in front:
            if (j ^ i) < group and full:
                assert (i < j) != bool((i ^ j) & 1), (i, j)
                front_pairs.add((min(i, j), max(i, j)))
                nfront_in += 1
        for j in back:
            assert (j ^ i) < group and full and (j < i) != bool((i ^ j) & 1), (i, j)
            nback += 1
    # every in-group pair is in exactly one front section
    assert nfront_in == nback == len(front_pairs) and nback > 0
    # balance: the atoms of a group's first and last quarter leave about the same number of pairs to their partners
    q = np.arange(nlocal) % group // (group // 4)
    first, last = nb[q == 0].mean(), nb[q == 3].mean()
    assert abs(first - last) < 0.15 * (first + last), (first, last)


@pytest.mark.parametrize("style", ["dpd/fast/meso", "dpd/meso"])
@pytest.mark.parametrize("npart", [0, 1, 2, 4])
def test_forces_and_trajectory_are_bit_identical_to_plain_rows(Meso, style, npart):
    """thermostat on, two atom types, 23 steps with 4 rebuilds (fused step boundary): x, v, f equal bit for bit"""
    x, v, types, _, lo, hi = make_polymer_box(11, frac=0.3)
    res = []
    for part in (1, 0):
        opts = (("row_part", part),) + ((("pair_npart", npart),) if npart else ())
        m = _start(Meso, x, v, lo, hi, style, opts, types=types)
        # (setup's forces come from the lane-per-atom kernel, whose per-thread floating-point sums depend on the entry order:
        # the run starts from the ring kernel's forces instead, which do not)
        f_lane = m.gather()[2]
        m.force_clear(); m.compute(0, 0)
        f0 = m.gather()[2]
        assert np.abs(f_lane - f0).max() < (1e-4 if "fast" in style else 1e-10) * np.abs(f0).max()
        m.run(23)
        res.append((f0,) + tuple(m.gather()[:3]))
        assert m.neigh_parts()["parted"] == bool(part)
        m.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)


def test_unpaired_and_lane_kernels_walk_whole_rows(Meso):
    """pair_share 0 (no pairing: plain rows are built), the lane-per-atom force kernel on partitioned rows (it walks the whole
    extent and skips the padding) and the energy / virial tally agree with the default path"""
    x, v, lo, hi = make_box(10)
    ref = _start(Meso, x, v, lo, hi, "dpd/meso")
    f_ref = ref.gather()[2]
    pe_ref, p_ref = ref.pe(), ref.pressure()
    ref.close()
    m = _start(Meso, x, v, lo, hi, "dpd/meso", (("pair_share", 0),))
    # (setup's forces: the lane-per-atom kernel's per-thread fp64 sums, which depend on the entry order in the last bits)
    assert not m.neigh_parts()["parted"]
    assert np.abs(m.gather()[2] - f_ref).max() < 1e-12 * np.abs(f_ref).max()
    m.close()
    m = _start(Meso, x, v, lo, hi, "dpd/meso", (("row_part", 0),))
    assert np.abs(m.gather()[2] - f_ref).max() < 1e-12 * np.abs(f_ref).max()
    assert m.pe() == pytest.approx(pe_ref, rel=1e-13) and m.pressure() == pytest.approx(p_ref, rel=1e-13)
    m.close()


@pytest.mark.parametrize("special", [(0.0, 0.0, 0.0), (0.0, 1.0, 1.0)])
def test_exclusion_filter_keeps_the_sections(Meso, special):
    """bonded chains: the special-bond filter compacts front and back section separately; rows, forces and a 23-step trajectory
    equal those of plain rows"""
    from test_gpu_bonds import _setup
    x, v, types, bonds, lo, hi = make_polymer_box(12, frac=0.4)
    res = []
    for part in (1, 0):
        m = Meso()
        m.set_option("row_part", part)
        _setup(m, x, v, types, bonds, lo, hi, sigma=3.0, special=special, style="dpd/fast/meso")
        nlocal = m.counts()[0]
        parts = m.neigh_parts(raw=True)
        count, table = m.neigh_table()
        rows = [frozenset(table[i, :count[i]]) for i in range(nlocal)]
        if part:
            assert parts["parted"]
            nf, nb, fr, bk = parts["nfront"], parts["nback"], parts["front"], parts["back"]
            assert np.array_equal(nf + nb, count) and nb.sum() > 0
            for i in range(nlocal):
                assert (fr[i, nf[i]:(nf[i] + 7) & ~7] == i).all() and (bk[i, nb[i]:(nb[i] + 7) & ~7] == i).all()
                assert not (fr[i, :nf[i]] == i).any() and not (bk[i, :nb[i]] == i).any()
        m.force_clear(); m.compute(0, 0); m.bond_compute(0)      # (the ring kernel's forces: independent of the entry order)
        f0 = m.gather()[2]
        m.run(23)
        res.append((rows, f0) + tuple(m.gather()[:3]))
        m.close()
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("style,every,parted", [("dpd/fast/meso", 1, False), ("dpd/fast/meso", 3, False), ("dpd/fast/meso", 4, True),
                                                 ("dpd/meso", 1, False), ("dpd/meso", 2, True), ("dpd/meso", 5, True)])
def test_sections_are_written_when_they_pay(Meso, style, every, parted):
    """option row_part -1 (default): the two sections cost the list builder and save every force launch of the interval - they are
    written from a rebuild interval of 4 steps (fp32 style) / 2 steps (fp64 style) on; plain rows come out of the cheaper row-out"""
    x, v, lo, hi = make_box(8)
    m = _start(Meso, x, v, lo, hi, style, every=every)
    m.run(2 * every + 1)
    p = m.neigh_parts()
    assert p["parted"] == parted and (p["nback"].sum() > 0) == parted
    m.close()
