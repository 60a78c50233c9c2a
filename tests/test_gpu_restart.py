"""restart files of the device-resident state and the profiler window.  A run continued from a restart
file written on a neighbour-rebuild step is bit-identical to the uninterrupted run: the file carries the forces of the
interrupted step, and the fixed-point force sums do not depend on the storage order.  (Between rebuilds the unwrapped
positions round differently in the fp32 merged coordinates once setup has wrapped them - the same trajectory to 1e-7.)"""
import threading

import numpy as np
from conftest import join_ranks
import pytest

from meso_amd.datagen import chain_angles, make_box, make_polymer_box

pytestmark = pytest.mark.gpu


def _fluid(m, x, v, lo, hi, style):
    m.read_atoms(x, v, lo, hi)
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style(style, 1.0, 419084618)
    m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
    m.timestep(0.005)
    m.setup()


@pytest.mark.parametrize("style", ["dpd/meso", "dpd/fast/meso"])
@pytest.mark.parametrize("at", [20, 35])
def test_continuation_is_bit_identical(tmp_path, style, at):
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(8)
    f = tmp_path / "a.rst"
    with Meso() as m:
        _fluid(m, x, v, lo, hi, style)
        m.run(at)
        m.write_restart(f)
        m.run(20)
        ref = m.gather()
        t_ref = m.temperature()
    with Meso() as m:
        m.read_restart(f)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)
        m.setup()
        assert m.ntimestep == at
        m.run(20)
        got = m.gather()
        assert m.temperature() == t_ref
    for a, b in zip(ref[:3], got[:3]):
        assert np.array_equal(a, b)
    assert np.array_equal(ref[3], got[3]) and np.array_equal(ref[4], got[4])


@pytest.mark.parametrize("at", [20, 25])
def test_restarted_run_matches_the_oracles_uninterrupted_run(tmp_path, at):
    """SURVEY.md 8f row 4 against the ORACLE, not against the HIP path itself: pair_style dpd/meso with the thermostat on
    (sigma = 3), written at step `at` (a rebuild step: between two rebuilds the restarted run wraps and re-merges the coordinates, whose
    fp32 rounding then differs - see the module docstring) by MesoPairDPD::write_restart's counterpart
    (/root/reference/src/USER-MESO/pair_dpd_meso.cu:363-447: the pair record carries seed and coefficients, so the TEA stream
    continues), read back by a fresh context and continued to step 32: positions, velocities and forces must equal the CPU mirror's
    (oracle/meso_sim.py) UNINTERRUPTED 32 steps - the tolerance of test_trajectory_vs_meso_oracle (1e-9 / 5e-8)."""
    from meso_amd.api import Meso
    from oracle.meso_sim import MesoRefSim
    L = 7
    x, v, lo, hi = make_box(L)
    s = MesoRefSim(x, v, lo, hi, every=5)
    s.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
    s.setup()
    s.run(32)
    f = tmp_path / "o.rst"
    with Meso() as m:
        _fluid(m, x, v, lo, hi, "dpd/meso")
        m.run(at)
        m.write_restart(f)
    with Meso() as m:
        m.read_restart(f)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)
        m.setup()
        assert m.ntimestep == at
        m.run(32 - at)
        xg, vg, fg = m.gather()[:3]
        assert m.ntimestep == 32
        assert m.temperature() == pytest.approx(s.temperature, rel=1e-9)
    prd = s.hi - s.lo
    d = xg - s.x
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < 1e-9 and np.abs(vg - s.v).max() < 5e-8
    assert np.abs(fg - s.f).max() <= 1e-8 * np.abs(s.f).max()


def test_polymers_with_fene_and_angles_continue(tmp_path):
    from meso_amd.api import Meso
    x, v, types, bonds, lo, hi = make_polymer_box(7, frac=0.3)
    angles = chain_angles(bonds)
    f = tmp_path / "p.rst"

    def styles(m):
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)

    with Meso() as m:
        m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
        m.special_bonds(0.0, 0.0, 1.0)
        m.read_bonds(bonds)
        m.read_angles(angles)
        m.bond_style("fene/meso", 1)
        m.bond_coeff(1, 40.0, 1.2, 0.5, 0.4)
        m.angle_style("harmonic/meso", 1)
        m.angle_coeff(1, 8.0, 150.0)
        styles(m)
        m.pair_style("dpd/meso", 1.0, 419084618)
        for (i, j), a in {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}.items():
            m.pair_coeff(i, j, a, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.005)
        m.setup()
        m.run(15)
        m.write_restart(f)
        m.run(15)
        ref = m.gather()
        e_ref = (m.ebond(), m.eangle())
    with Meso() as m:
        m.read_restart(f)
        styles(m)
        m.setup()
        m.run(15)
        got = m.gather()
        assert (m.ebond(), m.eangle()) == e_ref
    for a, b in zip(ref[:3], got[:3]):
        assert np.array_equal(a, b)


def test_restart_over_two_ranks(tmp_path):
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(10)
    f = tmp_path / "two.rst"
    gid = [np.frombuffer(np.random.default_rng(90 + k).bytes(8), np.uint8) for k in range(2)]
    out = {}

    def work(r, phase):
        m = Meso()
        m.comm_init(2, r, (2, 1, 1), "local", gid[phase])
        if phase == 0:
            _fluid(m, x, v, lo, hi, "dpd/meso")
            m.run(20)
            m.write_restart(f)
            m.run(20)
        else:
            m.read_restart(f)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=5, check=False)
            m.setup()
            m.run(20)
        out[phase, r] = m.gather(by_tag=False)
        m.close()

    for phase in (0, 1):
        th = [threading.Thread(target=work, args=(r, phase), daemon=True) for r in range(2)]
        [t.start() for t in th]
        join_ranks(th, None, 200)
    assert (tmp_path / "two.rst.0").exists() and (tmp_path / "two.rst.1").exists()

    def by_tag(phase):
        tag = np.concatenate([out[phase, r][3] for r in range(2)])
        xs = np.concatenate([out[phase, r][0] for r in range(2)])
        vs = np.concatenate([out[phase, r][1] for r in range(2)])
        o = np.argsort(tag)
        return tag[o], xs[o], vs[o]
    a, b = by_tag(0), by_tag(1)
    assert np.array_equal(a[0], np.arange(1, len(x) + 1)) and np.array_equal(a[0], b[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_two_rank_restart_between_rebuilds_keeps_migrant_forces(tmp_path):
    """A file written BETWEEN rebuilds (step 24 of a rebuild-every-5 run): setup's rebuild migrates the atoms that crossed a
    sub-domain face since step 20, and their saved forces must travel with them - the first half-kick after the restart uses
    them.  (A migrant arriving with f = 0 would be off by dt/2 |f| ~ 1e-2 in velocity after one step.)"""
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(10)
    f = tmp_path / "mid.rst"
    gid = [np.frombuffer(np.random.default_rng(190 + k).bytes(8), np.uint8) for k in range(2)]
    out, owner = {}, {}

    def work(r, phase):
        m = Meso()
        m.comm_init(2, r, (2, 1, 1), "local", gid[phase])
        if phase == 0:
            _fluid(m, x, v, lo, hi, "dpd/meso")
            m.run(24)
            owner[phase, r] = m.gather(by_tag=False)[3]
            m.write_restart(f)
            m.run(1)
        else:
            m.read_restart(f)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=5, check=False)
            m.setup()
            owner[phase, r] = m.gather(by_tag=False)[3]
            m.run(1)
        out[phase, r] = m.gather(by_tag=False)
        m.close()

    for phase in (0, 1):
        th = [threading.Thread(target=work, args=(r, phase), daemon=True) for r in range(2)]
        [t.start() for t in th]
        join_ranks(th, None, 200)

    def by_tag(phase):
        tag = np.concatenate([out[phase, r][3] for r in range(2)])
        o = np.argsort(tag)
        return [np.concatenate([out[phase, r][k] for r in range(2)])[o] for k in range(3)]
    a, b = by_tag(0), by_tag(1)
    # the restart's rebuild really moved atoms to the other rank
    assert set(owner[0, 0].tolist()) != set(owner[1, 0].tolist())
    d = a[0] - b[0]
    d -= np.round(d / (hi - lo)) * (hi - lo)
    assert np.abs(d).max() < 1e-9 and np.abs(a[1] - b[1]).max() < 1e-6


def test_bad_files_are_refused(tmp_path):
    from meso_amd.api import Meso, MesoError
    (tmp_path / "junk").write_bytes(b"not a restart file at all")
    with Meso() as m:
        with pytest.raises(MesoError):
            m.read_restart(tmp_path / "junk")
        with pytest.raises(MesoError):
            m.read_restart(tmp_path / "missing")
        with pytest.raises(MesoError):
            m.write_restart(tmp_path / "early")          # nothing to write yet


def test_script_driver_restart_commands(tmp_path):
    from meso_amd.api import Meso
    from meso_amd.datagen import write_data
    x, v, lo, hi = make_box(6)
    write_data(str(tmp_path / "b.data"), x, lo, hi, v=v)
    head = """dimension 3
units lj
boundary p p p
atom_style dpd/atomic/meso
neighbor 0.3 bin
neigh_modify delay 0 every 5 check no
"""
    tail = """run_style mvv/meso
compute mobile all temp/meso
fix 1 all nve/meso
thermo_style custom step c_mobile
thermo 10
timestep 0.005
"""
    pair = "pair_style dpd/meso 1.0 419084618\npair_coeff 1 1 15.0 4.5 3.0 1.0 1.0\n"
    (tmp_path / "a.run").write_text(head + "read_data %s\n" % (tmp_path / "b.data") + pair + tail +
                                    "run 10\nwrite_restart %s\nrun 10\n" % (tmp_path / "s.rst"))
    (tmp_path / "b.run").write_text(head + "read_restart %s\n" % (tmp_path / "s.rst") + tail + "run 10\n")
    with Meso() as m:
        m.script(str(tmp_path / "a.run"))
        ref = m.gather()
    with Meso() as m:
        log = m.script(str(tmp_path / "b.run"))
        got = m.gather()
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    assert any(ln.split() and ln.split()[0] == "20" for ln in log.splitlines())


def test_profile_window_modes():
    """without an attached profiler the window calls are no-ops; the run is unchanged"""
    from meso_amd.api import Meso, MesoError
    x, v, lo, hi = make_box(6)
    res = []
    for mode in (None, "core", "interval"):
        with Meso() as m:
            if mode:
                m.profile_window(mode, 5, 12)
            _fluid(m, x, v, lo, hi, "dpd/meso")
            m.run(20)
            res.append(m.gather()[0])
            if mode is None:
                with pytest.raises(MesoError):
                    m.profile_window("interval", 10, 5)
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[0], res[2])
