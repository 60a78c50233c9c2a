"""Tagged neighbour rows and the pruned ("shell") walk of the force kernel (round 4).

The list builder leaves in every row entry its distance shell at build time and its Newton-pairing class; the step boundaries book
the fastest atom of every step; the force kernel looks only at the entries whose shell can be inside the cutoff on the current step
(r_build < r_c + 2 D).  The reference keeps the same build-time knowledge as "core entries from the row front, skin entries from the
back" (/root/reference/src/USER-MESO/neigh_build_meso.cu:91-115) and walks everything (pair_dpd_fast_meso.cu:124-145).

A skipped entry contributes exactly zero and the 64-bit fixed-point force sums do not depend on the order of the others, so the
claim tested here is BIT-IDENTITY with the full walk (shell_walk 2: same rows, every shell; shell_walk 0: plain rows, every entry
gathered and tested by its atom's lane - the round-3 kernel), and the tags themselves against distances recomputed on the host.
The option is OFF by default: measured against the plain-row kernel it is 1-4 % faster as a kernel at 64^3 and pays 12 % more
list-builder time for the tags, slower at every size as a whole step (profiles/r04_notes.md).
"""
import numpy as np
import pytest

from conftest import DP_RUN
from meso_amd.datagen import make_box

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Meso():
    from meso_amd.api import Meso
    return Meso


def _run(Meso, L, style, opts, steps, every=5, vscale=1.0, types=None, dt=0.005, skin=0.3):
    x, v, lo, hi = make_box(L)
    m = Meso()
    for k, val in opts:
        m.set_option(k, val)
    if types is None:
        m.read_atoms(x, v * vscale, lo, hi)
    else:
        m.read_atoms(x, v * vscale, lo, hi, types=types, ntypes=2)
    m.neighbor(skin)
    m.neigh_modify(delay=0, every=every, check=False)
    m.pair_style(style, 1.0, DP_RUN["seed"])
    if types is None:
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
    else:
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
        m.pair_coeff(2, 2, 15.0, 4.5, 3.0, 1.0, 0.8)
        m.pair_coeff(1, 2, 40.0, 4.5, 3.0, 1.0, 0.9)
    m.timestep(dt)
    m.setup()
    m.run(steps)
    out = m.gather()[:3]
    name = m.pair_kernel_name()
    m.close()
    return out, name


@pytest.mark.parametrize("style", ["dpd/fast/meso", "dpd/meso"])
@pytest.mark.parametrize("L", [10, 16])
def test_shell_walk_is_bit_identical(Meso, style, L):
    """23 steps (4 rebuilds, thermostat on): pruned walk == every shell walked == plain rows, bit for bit - positions, velocities,
    forces.  L = 10: two lanes per atom (pairing groups of 128); L = 16: below and above ... one lane per atom is forced too."""
    ref, kname = _run(Meso, L, style, (("shell_walk", 0),), 23)
    assert kname.rstrip(">").endswith(", 0")
    for opts in ((), (("shell_walk", 2),), (("pair_npart", 1),), (("pair_npart", 4),), (("pair_share", 0),), (("fuse_pair", 0),)):
        if not any(k == "shell_walk" for k, _ in opts):
            opts = opts + (("shell_walk", 1),)
        got, kname = _run(Meso, L, style, opts, 23)
        assert kname.rstrip(">").endswith(", 1"), kname
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), opts


@pytest.mark.parametrize("style", ["dpd/fast/meso", "dpd/meso"])
def test_shell_walk_with_a_bound_beyond_the_skin_and_long_intervals(Meso, style):
    """(i) velocities scaled by 12 (2 D per step ~ 0.6 > skin): from the second step of an interval on the bound covers every
    shell - the walk is the full walk; (ii) rebuild every 20 steps: more steps than the displacement account has slots (16) - full
    walk from there on.  Both bit-identical to plain rows (the list is the same on both sides, valid or not)."""
    for kw in (dict(vscale=12.0, steps=12), dict(every=20, steps=43)):
        ref, _ = _run(Meso, 9, style, (("shell_walk", 0),), **kw)
        got, _ = _run(Meso, 9, style, (("shell_walk", 1),), **kw)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), kw


def test_shell_walk_two_types_per_pair_cutoffs(Meso):
    """Several types with unequal cutoffs: the shells are relative to the LARGEST cutoff; bit-identical to plain rows."""
    rng = np.random.default_rng(5)
    types = rng.integers(1, 3, 4 * 9 ** 3).astype(np.int32)
    for style in ("dpd/fast/meso", "dpd/meso"):
        ref, _ = _run(Meso, 9, style, (("shell_walk", 0),), 17, types=types)
        got, _ = _run(Meso, 9, style, (("shell_walk", 1),), 17, types=types)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b)


def test_row_tags_match_build_time_distances(Meso):
    """Raw table after setup: every entry's shell against r^2 recomputed from the merged coordinates the builder read (shell 0:
    r^2 < base; s >= 1: r^2 in [base + (s-1)/k, base + s/k), up to the rounding of one fp32 fma at the boundaries), the mirror and
    pair-once bits against the two indices, the tail slots of the last chunk = the atom itself with the pad bits."""
    L = 10
    x, v, lo, hi = make_box(L)
    m = Meso()
    m.set_option("shell_walk", 1)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/fast/meso", 1.0, DP_RUN["seed"]); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    m.setup()
    tg = m.neigh_tags(raw=True)
    assert tg["tagged"] and tg["group"] in (64, 128, 256)
    count, table = m.neigh_table()
    raw = tg["table"]
    c4, _ = m.merged()
    nl = m.counts()[0]
    base, k = tg["base"], tg["k"]
    assert abs(base - (1.0 + 2 * tg["eps"]) ** 2) < 1e-6 and 0 < tg["eps"] < 1e-4
    idx = (raw & 0x01FFFFFF).astype(np.int64)
    shell = (raw >> 28) & 7
    mirror = (raw >> 31) & 1
    once = (raw >> 25) & 1
    g = tg["group"]
    nbad = 0
    for i in range(nl):
        n = count[i]
        j = idx[i, :n]
        assert np.array_equal(j, table[i, :n])
        d = c4[i, :3].astype(np.float64) - c4[j, :3].astype(np.float64)
        r2 = (d * d).sum(1)
        want = np.where(r2 < base, 0, np.floor((r2 - base) * k).astype(np.int64) + 1)
        off = shell[i, :n].astype(np.int64) != want
        if off.any():
            # only at a shell boundary (fp32 rounding of r^2 * k + off)
            t = (r2[off] - base) * k
            assert np.all(np.abs(t - np.round(t)) < 2e-4), (i, r2[off])
            nbad += int(off.sum())
        same = (j // g) == (i // g)
        assert np.array_equal(mirror[i, :n].astype(bool), same & (j < i))
        assert np.array_equal(once[i, :n].astype(bool), same & (j > i))
        assert shell[i, :n].max() <= 7
        pad = raw[i, n:(n + 7) // 8 * 8]
        assert np.all(pad == ((0xFE000000 | i) & 0xFFFFFFFF))
    assert nbad < 1e-4 * count.sum()
    m.close()
