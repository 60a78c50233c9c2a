"""configs[4]: bonded chains (atom_style dpd/bond/meso, bond_style harmonic/meso, special-bond exclusions) against
the CPU oracle, on one rank and decomposed over 8 ranks of one GPU."""
import threading

import numpy as np
from conftest import join_ranks
import pytest

from meso_amd.datagen import make_box, make_polymer_box

pytestmark = pytest.mark.gpu

A = {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}


FENE = (40.0, 1.2, 0.5, 0.4)       # K R0 epsilon sigma: WCA core below 0.45, bonds start near 0.5


def _setup(m, x, v, types, bonds, lo, hi, style="dpd/meso", sigma=3.0, special=(0.0, 0.0, 0.0), bond="harmonic"):
    m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
    m.special_bonds(*special)
    m.read_bonds(bonds)
    m.bond_style(bond + "/meso", 1)
    if bond == "fene":
        m.bond_coeff(1, *FENE)
    else:
        m.bond_coeff(1, 50.0, 0.5)
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style(style, 1.0, 419084618)
    for (i, j), a in A.items():
        m.pair_coeff(i, j, a, 4.5, sigma, 1.0, 1.0)
    m.timestep(0.005)
    m.setup()


def _oracle(x, v, types, bonds, lo, hi, sigma=3.0, special=(0.0, 0.0, 0.0), fast=False, bond="harmonic"):
    from oracle.meso_sim import MesoRefSim
    s = MesoRefSim(x, v, lo, hi, types=types, ntypes=2, fast=fast)
    for (i, j), a in A.items():
        s.pair_coeff(i, j, a, 4.5, sigma, 1.0, 1.0)
    s.set_bonds(bonds, {1: FENE if bond == "fene" else (50.0, 0.5)}, special, style=bond)
    s.setup()
    return s


@pytest.mark.parametrize("special", [(0.0, 0.0, 0.0), (0.0, 1.0, 1.0), (1.0, 1.0, 1.0)])
def test_forces_rows_and_bond_energy(oracle, special):
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(8, frac=0.3)
    s = _oracle(x, v, types, bonds, lo, hi, special=special)
    with Meso() as m:
        _setup(m, x, v, types, bonds, lo, hi, special=special)
        xg, vg, fg, tag, typ = m.gather()
        count, _ = m.neigh_table()
        _, _, _, tag_dev, _ = m.gather(by_tag=False)
        assert np.array_equal(typ, types)
        assert np.array_equal(count[np.argsort(tag_dev)], s.count)          # special partners are not in the rows
        assert np.abs(fg - s.f).max() < 1e-9 * np.abs(s.f).max()
        assert m.ebond() == pytest.approx(s.e_bond, rel=1e-10)
    if special == (1.0, 1.0, 1.0):
        assert s.count.sum() > _oracle(x, v, types, bonds, lo, hi).count.sum()


@pytest.mark.parametrize("special", [(0.0, 0.0, 0.0), (1.0, 1.0, 1.0)])
def test_fene_forces_and_bond_energy(oracle, special):
    """bond_style fene/meso (bond_fene_meso.cu:82-147): forces, rows and bond energy against the oracle; some bonds are
    pushed into the WCA core and one beyond the clamp of the log argument"""
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(8, frac=0.3)
    x = x.copy()
    a, b = int(bonds[0][0]) - 1, int(bonds[0][1]) - 1
    x[b] = x[a] + np.array([1.17, 0.0, 0.0])          # 1 - r^2/R0^2 = 0.049 < 0.1: clamped
    a, b = int(bonds[5][0]) - 1, int(bonds[5][1]) - 1
    x[b] = x[a] + np.array([0.0, 0.41, 0.0])          # inside the WCA core
    x = lo + np.mod(x - lo, hi - lo)
    s = _oracle(x, v, types, bonds, lo, hi, special=special, bond="fene")
    with Meso() as m:
        _setup(m, x, v, types, bonds, lo, hi, special=special, bond="fene")
        fg = m.gather()[2]
        assert np.abs(fg - s.f).max() < 1e-9 * np.abs(s.f).max()
        assert m.ebond() == pytest.approx(s.e_bond, rel=1e-10)
        with pytest.raises(Exception):
            m.bond_coeff(1, 40.0, 1.2)                 # fene needs K R0 epsilon sigma


def test_fene_polymer_trajectory(oracle):
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(7, frac=0.3)
    s = _oracle(x, v, types, bonds, lo, hi, bond="fene")
    with Meso() as m:
        _setup(m, x, v, types, bonds, lo, hi, bond="fene")
        m.run(12)
        s.run(12)
        xg, vg = m.gather()[:2]
        eb = m.ebond()
    prd = hi - lo
    d = xg - s.x
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < 1e-9 and np.abs(vg - s.v).max() < 1e-7
    assert eb == pytest.approx(s.e_bond, rel=1e-7)


@pytest.mark.parametrize("style,tol", [("dpd/meso", 1e-9), ("dpd/fast/meso", 5e-5)])
def test_polymer_trajectory(oracle, style, tol):
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(7, frac=0.3)
    sigma = 3.0 if style == "dpd/meso" else 0.0
    s = _oracle(x, v, types, bonds, lo, hi, sigma=sigma, fast=style != "dpd/meso")
    with Meso() as m:
        _setup(m, x, v, types, bonds, lo, hi, style=style, sigma=sigma)
        m.run(12)
        s.run(12)
        xg, vg = m.gather()[:2]
    prd = hi - lo
    d = xg - s.x
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < tol and np.abs(vg - s.v).max() < 100 * tol


def test_chains_survive_decomposition():
    """8 ranks: chains straddle sub-domain faces and beads migrate with their bond / special lists."""
    from meso_amd.api import Meso
    L = 12
    x, v, types, bonds, lo, hi = make_polymer_box(L, frac=0.2)
    gid = np.frombuffer(np.random.default_rng(77).bytes(8), np.uint8)

    def run(nranks, grid):
        out, errs = [None] * nranks, []

        def work(r):
            try:
                m = Meso()
                if nranks > 1:
                    m.comm_init(nranks, r, grid, "local", gid)
                _setup(m, x, v, types, bonds, lo, hi)
                e0 = m.ebond()
                f0 = m.gather(by_tag=False)
                m.run(40)
                m.force_clear("local"); m.compute(); m.bond_compute(1)
                out[r] = (e0, f0, m.gather(by_tag=False), m.temperature(), m.counts(), m.ebond())
                m.close()
            except Exception as e:   # noqa: BLE001
                errs.append((r, repr(e)))
        th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
        [t.start() for t in th]
        join_ranks(th, errs, 300)
        assert not errs, errs
        return out

    one = run(1, (1, 1, 1))
    many = run(8, (2, 2, 2))
    n = len(x)
    assert sum(o[4][0] for o in many) == n
    # chains start at r = r0 exactly, so the initial bond energy is fp32 rounding noise; after 40 steps it is O(N)
    assert all(abs(o[0] - one[0][0]) < 1e-6 for o in many)
    assert all(abs(o[5] - many[0][5]) < 1e-9 * abs(many[0][5]) for o in many)          # one global value on all ranks
    assert abs(many[0][5] - one[0][5]) < 0.15 * abs(one[0][5]) and one[0][5] > 1.0
    f1 = one[0][1][2][np.argsort(one[0][1][3])]
    tags = np.concatenate([o[1][3] for o in many])
    f8 = np.concatenate([o[1][2] for o in many])[np.argsort(tags)]
    assert np.array_equal(np.sort(tags), np.arange(1, n + 1))
    assert np.abs(f8 - f1).max() < 5e-6 * np.abs(f1).max()
    tags40 = np.concatenate([o[2][3] for o in many])
    assert np.array_equal(np.sort(tags40), np.arange(1, n + 1))                          # nobody lost after 40 steps
    assert abs(many[0][3] - one[0][3]) < 0.1


DECK = """dimension 3
units lj
boundary p p p
atom_style dpd/bond/meso
neighbor 0.3 bin
neigh_modify delay 0 every 5 check no
special_bonds lj 0.0 1.0 1.0
read_data {data}
run_style mvv/meso
bond_style harmonic/meso
bond_coeff 1 50.0 0.5
pair_style dpd/meso 1.0 419084618
pair_coeff 1 1 15.0 4.5 3.0 1.0 1.0
pair_coeff 2 2 15.0 4.5 3.0 1.0 1.0
pair_coeff 1 2 40.0 4.5 3.0 1.0 1.0
compute mobile all temp/meso
fix 1 all nve/meso
thermo_style custom step c_mobile
thermo 10
timestep 0.005
run 10
"""


def test_script_driver_runs_a_polymer_deck(oracle, tmp_path):
    """The mini driver reads atom_style dpd/bond/meso files (molecule column, Bonds section) and the bond commands;
    the trajectory equals the API-driven one (same engine calls) and the oracle's."""
    from meso_amd.api import Meso
    from meso_amd.datagen import write_data
    x, v, types, bonds, lo, hi = make_polymer_box(6, frac=0.3)
    write_data(str(tmp_path / "poly.data"), x, lo, hi, v=v, types=types, ntypes=2, bonds=bonds)
    (tmp_path / "poly.run").write_text(DECK.format(data=tmp_path / "poly.data"))
    special = (0.0, 1.0, 1.0)
    s = _oracle(x, v, types, bonds, lo, hi, special=special)
    s.run(10)
    with Meso() as m:
        log = m.script(str(tmp_path / "poly.run"))
        xg, vg, fg, tag, typ = m.gather()
        d = xg - s.x
        d -= np.round(d / (hi - lo)) * (hi - lo)
        assert np.abs(d).max() < 1e-9
        assert m.ebond() == pytest.approx(s.e_bond, rel=1e-8)
        rows = [ln.split() for ln in log.splitlines() if ln.split() and ln.split()[0] in ("0", "10")]
        assert float(rows[-1][1]) == pytest.approx(s.temperature, rel=1e-7)      # thermo prints 8 digits


def test_bonded_rebuild_without_host_round_trip_gives_the_same_trajectory():
    """The rebuild that leaves its counts on the device (async_counts, default) also serves bonded systems: the tag map and the
    cell-ordered tags are built from the device-side ghost count.  23 steps (4 rebuilds) of a polymer deck with exclusions are
    bit-identical with and without the host round trip, also when the ghost kernels' grids are far too small and have to loop."""
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(14, frac=0.4)
    res = []
    # (round 5, bonded decks too: the rebuild's count in the epilogue of the force launch in front of it - fuse_count -, the ordering
    # kernel + a streaming gather that moves the topology lists as well - split_gather, by default from 50 000 atoms on -, the ghost
    # tiles in the gather's launch - merge_ghosts)
    for opts in ((("async_counts", 0), ("ghost_epilogue", 0)), (), (("async_grid_scale", 0.05),), (("ghost_epilogue", 0),),
                 (("fuse_count", 0), ("split_gather", 0)), (("split_gather", 1),), (("split_gather", 1), ("merge_ghosts", 0)),
                 (("split_gather", 1), ("fuse_count", 0), ("lean_boundary", 0)), (("fused_rebuild", 0),)):
        with Meso() as m:
            for k, val in opts:
                m.set_option(k, val)
            _setup(m, x, v, types, bonds, lo, hi, sigma=3.0, special=(0.0, 1.0, 1.0), style="dpd/fast/meso")
            m.run(23)
            res.append(m.gather())
            assert m.neigh_info()["nbuild"] >= 4 and m.counts()[1] > 0
    for other in res[1:]:
        for a, b in zip(res[0][:3], other[:3]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("bond", ["harmonic", "fene"])
def test_bonds_in_the_force_kernel_epilogue_give_the_same_trajectory(bond):
    """fuse_bonds (default): on steps whose step boundary runs in the force kernel's epilogue the bonds of an atom are
    evaluated there too, with the device function the bond kernel calls - 23 steps are bit-identical to the run that launches
    the bond kernel first (fuse_bonds 0) and to the run with every kernel on its own (fuse_pair 0)."""
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(12, frac=0.4)
    res = []
    for opts in ((("fuse_bonds", 0),), (), (("fuse_pair", 0),)):
        with Meso() as m:
            for k, val in opts:
                m.set_option(k, val)
            _setup(m, x, v, types, bonds, lo, hi, sigma=3.0, special=(0.0, 1.0, 1.0), style="dpd/fast/meso", bond=bond)
            m.run(23)
            res.append(m.gather())
    for other in res[1:]:
        for a, b in zip(res[0][:3], other[:3]):
            assert np.array_equal(a, b)


def test_star_molecule_with_long_special_lists(oracle):
    """A hub bead bonded to 20 arms: with special_bonds 0 0 0 the hub has 20 special partners and every arm has 20 too (the hub
    and the 19 other arms) - longer than the 16 the exclusion filter keeps in registers, so the rest of the list is read from
    memory (k_filter_exclusion, brick.hip).  Rows and forces against the oracle."""
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(8)
    x = x.copy()
    rng = np.random.default_rng(11)
    hub = 100
    arms = np.arange(200, 220)
    d = rng.normal(size=(20, 3))
    x[arms] = x[hub] + 0.45 * d / np.linalg.norm(d, axis=1)[:, None]
    x = lo + np.mod(x - lo, hi - lo)
    types = np.ones(len(x), np.int32)
    types[arms] = 2
    bonds = np.array([(hub + 1, a + 1, 1) for a in arms], np.int32)
    s = _oracle(x, v, types, bonds, lo, hi, special=(0.0, 0.0, 0.0))
    with Meso() as m:
        _setup(m, x, v, types, bonds, lo, hi, special=(0.0, 0.0, 0.0))
        xg, vg, fg, tag, typ = m.gather()
        count, _ = m.neigh_table()
        tag_dev = m.gather(by_tag=False)[3]
        cnt = count[np.argsort(tag_dev)]
        assert np.array_equal(cnt, s.count)
        assert np.abs(fg - s.f).max() < 1e-9 * np.abs(s.f).max()
        m.run(7)          # a rebuild with the long lists on the way
        assert np.isfinite(m.gather()[0]).all()
    # the arms sit within 0.9 of each other: without the filter each of them would list the other 20 beads of the star
    s_none = _oracle(x, v, types, bonds, lo, hi, special=(1.0, 1.0, 1.0))
    assert (s_none.count[arms] - s.count[arms]).min() >= 19
