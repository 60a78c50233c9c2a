"""atom_style dpd/angle/meso + angle_style harmonic/meso (angle_harmonic_meso.cu:46-172, tag mapping
neighbor_meso.cu:161-182) against the CPU oracle, on one rank and decomposed over 8 ranks of one GPU."""
import threading

import numpy as np
from conftest import join_ranks
import pytest

from meso_amd.datagen import chain_angles, make_polymer_box

pytestmark = pytest.mark.gpu

A = {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}
ANGLE = (8.0, 150.0)      # K, theta0 in degrees


def _setup(m, x, v, types, bonds, angles, lo, hi, sigma=3.0, special=(0.0, 0.0, 1.0), r0=0.5):
    m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
    m.special_bonds(*special)
    m.read_bonds(bonds)
    m.read_angles(angles)
    m.bond_style("harmonic/meso", 1)
    m.bond_coeff(1, 50.0, r0)
    m.angle_style("harmonic/meso", 1)
    m.angle_coeff(1, *ANGLE)
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/meso", 1.0, 419084618)
    for (i, j), a in A.items():
        m.pair_coeff(i, j, a, 4.5, sigma, 1.0, 1.0)
    m.timestep(0.005)
    m.setup()


def _oracle(x, v, types, bonds, angles, lo, hi, sigma=3.0, special=(0.0, 0.0, 1.0)):
    from oracle.meso_sim import MesoRefSim
    s = MesoRefSim(x, v, lo, hi, types=types, ntypes=2)
    for (i, j), a in A.items():
        s.pair_coeff(i, j, a, 4.5, sigma, 1.0, 1.0)
    s.set_bonds(bonds, {1: (50.0, 0.5)}, special)
    s.set_angles(angles, {1: ANGLE})
    s.setup()
    return s


def test_forces_and_angle_energy(oracle):
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(8, frac=0.3)
    angles = chain_angles(bonds)
    assert len(angles) == 4 * (len(bonds) // 5)            # A2B4 chains: 5 bonds, 4 angles
    s = _oracle(x, v, types, bonds, angles, lo, hi)
    with Meso() as m:
        _setup(m, x, v, types, bonds, angles, lo, hi)
        fg = m.gather()[2]
        assert np.abs(fg - s.f).max() < 1e-9 * np.abs(s.f).max()
        assert m.eangle() == pytest.approx(s.e_angle, rel=1e-10)
        assert m.ebond() == pytest.approx(s.e_bond, rel=1e-10, abs=1e-9)
        # Angle::compute as a separate call (host-driven step): adds the same forces once more
        f0 = m.gather()[2]
        m.angle_compute(0)
        f1 = m.gather()[2]
    only = _oracle(x, v * 0.0, types, bonds, angles, lo, hi, sigma=0.0)
    only.f[:] = 0.0
    only._angle_forces()
    assert np.abs((f1 - f0) - only.f).max() < 1e-9 * np.abs(only.f).max()


def test_trajectory_with_angles(oracle):
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(7, frac=0.3)
    angles = chain_angles(bonds)
    s = _oracle(x, v, types, bonds, angles, lo, hi)
    with Meso() as m:
        _setup(m, x, v, types, bonds, angles, lo, hi)
        m.run(12)
        s.run(12)
        xg, vg = m.gather()[:2]
        ea = m.eangle()
    prd = hi - lo
    d = xg - s.x
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < 1e-9 and np.abs(vg - s.v).max() < 1e-7
    assert ea == pytest.approx(s.e_angle, rel=1e-7)


def test_angles_need_bonds_first():
    from meso_amd.api import Meso, MesoError
    x, v, types, bonds, lo, hi = make_polymer_box(6, frac=0.3)
    with Meso() as m:
        m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
        with pytest.raises(MesoError):
            m.read_angles(chain_angles(bonds))
        m.read_bonds(bonds)
        with pytest.raises(MesoError):
            m.read_angles(np.array([[1, 1, 2, 1]]))       # repeated atom
        with pytest.raises(MesoError):
            m.angle_coeff(1, 5.0, 120.0)                  # before angle_style


def test_angles_survive_decomposition():
    """8 ranks: angles straddle sub-domain faces and beads migrate with their angle lists."""
    from meso_amd.api import Meso
    L = 12
    # both outer beads of an angle must be inside the apex' rank or its 1.3 ghost shell (LAMMPS: "Angle atoms missing"
    # otherwise): bonds of 0.35 keep 1-3 distances below 0.9
    x, v, types, bonds, lo, hi = make_polymer_box(L, frac=0.2, r0=0.35)
    angles = chain_angles(bonds)
    gid = np.frombuffer(np.random.default_rng(78).bytes(8), np.uint8)

    def run(nranks, grid):
        out, errs = [None] * nranks, []

        def work(r):
            try:
                m = Meso()
                if nranks > 1:
                    m.comm_init(nranks, r, grid, "local", gid)
                _setup(m, x, v, types, bonds, angles, lo, hi, r0=0.35)
                e0 = m.eangle()
                f0 = m.gather(by_tag=False)
                m.run(40)
                out[r] = (e0, f0, m.gather(by_tag=False), m.temperature(), m.counts(), m.eangle())
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
    # step-0 angle energy, one global value; coordinates are fp32 relative to each rank's own sub-box centre
    assert all(abs(o[0] - one[0][0]) < 2e-5 * abs(one[0][0]) for o in many), (one[0][0], [o[0] for o in many])
    f1 = one[0][1][2][np.argsort(one[0][1][3])]
    tags = np.concatenate([o[1][3] for o in many])
    f8 = np.concatenate([o[1][2] for o in many])[np.argsort(tags)]
    assert np.array_equal(np.sort(tags), np.arange(1, n + 1))
    assert np.abs(f8 - f1).max() < 5e-6 * np.abs(f1).max()
    tags40 = np.concatenate([o[2][3] for o in many])
    assert np.array_equal(np.sort(tags40), np.arange(1, n + 1))                          # nobody lost after 40 steps
    assert all(abs(o[5] - many[0][5]) < 1e-9 * abs(many[0][5]) for o in many)
    assert abs(many[0][5] - one[0][5]) < 0.15 * abs(one[0][5])
    assert abs(many[0][3] - one[0][3]) < 0.1


def test_a_partner_outside_the_ghost_shell_is_reported():
    """2 ranks, one chain stretched across more than the ghost cutoff: an error (LAMMPS' "Bond atoms missing"), not a
    fault - the missing index is replaced by the atom itself before any kernel uses it"""
    from meso_amd.api import Meso, MesoError
    L = 8
    x, v, types, bonds, lo, hi = make_polymer_box(L, frac=0.1)
    x = x.copy()
    a, b = int(bonds[0][0]) - 1, int(bonds[0][1]) - 1
    x[a] = [1.0, 4.0, 4.0]
    x[b] = [6.0, 4.0, 4.0]          # 3 from the periodic image, far beyond 1.3
    angles = chain_angles(bonds)
    gid = np.frombuffer(np.random.default_rng(79).bytes(8), np.uint8)
    errs = [None, None]

    def work(r):
        m = Meso()
        m.comm_init(2, r, (2, 1, 1), "local", gid)
        try:
            _setup(m, x, v, types, bonds, angles, lo, hi)
        except MesoError as e:
            errs[r] = str(e)
        m.close()
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(2)]
    [t.start() for t in th]
    join_ranks(th, errs, 120)
    assert any(e and "missing" in e for e in errs), errs


DECK = """dimension 3
units lj
boundary p p p
atom_style dpd/angle/meso
neighbor 0.3 bin
neigh_modify delay 0 every 5 check no
special_bonds lj 0.0 0.0 1.0
read_data {data}
run_style mvv/meso
bond_style fene/meso
bond_coeff 1 40.0 1.2 0.5 0.4
angle_style harmonic/meso
angle_coeff 1 8.0 150.0
pair_style dpd/meso 1.0 419084618
pair_coeff 1 1 15.0 4.5 3.0 1.0 1.0
pair_coeff 2 2 15.0 4.5 3.0 1.0 1.0
pair_coeff 1 2 40.0 4.5 3.0 1.0 1.0
compute mobile all temp/meso
compute pe all pe/meso
fix 1 all nve/meso
thermo_style custom step c_mobile pe
thermo 10
timestep 0.005
run 10
"""


def test_script_driver_runs_an_angle_deck(oracle, tmp_path):
    """atom_style dpd/angle/meso data file (Bonds + Angles), bond_style fene/meso, angle_style harmonic/meso through the
    mini driver: the trajectory equals the oracle's, thermo pe includes bond and angle energy."""
    from meso_amd.api import Meso
    from meso_amd.datagen import write_data
    from oracle.meso_sim import MesoRefSim
    x, v, types, bonds, lo, hi = make_polymer_box(6, frac=0.3)
    angles = chain_angles(bonds)
    write_data(str(tmp_path / "ang.data"), x, lo, hi, v=v, types=types, ntypes=2, bonds=bonds, angles=angles)
    (tmp_path / "ang.run").write_text(DECK.format(data=tmp_path / "ang.data"))
    s = MesoRefSim(x, v, lo, hi, types=types, ntypes=2)
    for (i, j), a in A.items():
        s.pair_coeff(i, j, a, 4.5, 3.0, 1.0, 1.0)
    s.set_bonds(bonds, {1: (40.0, 1.2, 0.5, 0.4)}, (0.0, 0.0, 1.0), style="fene")
    s.set_angles(angles, {1: ANGLE})
    s.setup()
    s.run(10)
    with Meso() as m:
        log = m.script(str(tmp_path / "ang.run"))
        xg = m.gather()[0]
        d = xg - s.x
        d -= np.round(d / (hi - lo)) * (hi - lo)
        assert np.abs(d).max() < 1e-9
        assert m.eangle() == pytest.approx(s.e_angle, rel=1e-8)
        assert m.ebond() == pytest.approx(s.e_bond, rel=1e-8)
        rows = [ln.split() for ln in log.splitlines() if ln.split() and ln.split()[0] in ("0", "10")]
        assert float(rows[-1][1]) == pytest.approx(s.temperature, rel=1e-7)
