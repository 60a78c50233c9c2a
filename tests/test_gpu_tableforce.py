"""pair_style dpd/tableforce/meso (pair_dpd_tableforce_meso.cu) - the fp32 kernel with a tabulated
conservative force (L points uniform in r/rc, linear filter) and uniform TEA pair noise."""
import numpy as np
import pytest

from meso_amd.datagen import make_box, make_polymer_box

pytestmark = pytest.mark.gpu

L = 33
R = np.linspace(0.0, 1.0, L)
TABLES = {(1, 1): 15.0 * (1 - R) + 3.0 * np.sin(np.pi * R), (2, 2): 15.0 * (1 - R), (1, 2): 40.0 * (1 - R) ** 2}


def _tab(m, x, v, lo, hi, types, ntypes, sigma):
    m.read_atoms(x, v, lo, hi, types=types, ntypes=ntypes)
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/tableforce/meso", 1.0, 419084618)
    for (i, j), t in TABLES.items():
        if j <= ntypes:
            m.pair_coeff_table(i, j, 4.5, sigma, t)
    m.timestep(0.005)
    m.setup()


def _oracle(x, v, lo, hi, types, ntypes, sigma):
    from oracle.meso_sim import MesoRefSim
    s = MesoRefSim(x, v, lo, hi, types=types, ntypes=ntypes, fast=True)
    for (i, j), t in TABLES.items():
        if j <= ntypes:
            s.pair_coeff_table(i, j, 4.5, sigma, t)
    s.setup()
    return s


@pytest.mark.parametrize("ntypes", [1, 2])
def test_forces_and_short_trajectory(oracle, ntypes):
    from meso_amd.api import Meso
    if ntypes == 1:
        x, v, lo, hi = make_box(8)
        types = None
    else:
        x, v, types, _, lo, hi = make_polymer_box(8, frac=0.3)
    s = _oracle(x, v, lo, hi, types, ntypes, 3.0)
    with Meso() as m:
        _tab(m, x, v, lo, hi, types, ntypes, 3.0)
        f0 = m.gather()[2]
    assert np.abs(f0 - s.f).max() < 5e-5 * np.abs(s.f).max()
    s = _oracle(x, v, lo, hi, types, ntypes, 0.0)              # trajectories without noise (fp32 signatures, see test_gpu_mini)
    with Meso() as m:
        _tab(m, x, v, lo, hi, types, ntypes, 0.0)
        m.run(10)
        s.run(10)
        xg, vg = m.gather()[:2]
    prd = hi - lo
    d = xg - s.x
    d -= np.round(d / prd) * prd
    assert np.abs(d).max() < 5e-5 and np.abs(vg - s.v).max() < 5e-3


def test_a_linear_table_is_the_standard_conservative_force():
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(8)
    out = {}
    for style in ("dpd/tableforce/meso", "dpd/fast/meso"):
        with Meso() as m:
            m.read_atoms(x, v, lo, hi)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=5, check=False)
            m.pair_style(style, 1.0, 419084618)
            if style == "dpd/fast/meso":
                m.pair_coeff(1, 1, 15.0, 4.5, 0.0, 1.0, 1.0)
            else:
                m.pair_coeff_table(1, 1, 4.5, 0.0, 15.0 * (1 - R))
            m.timestep(0.005)
            m.setup()
            out[style] = m.gather()[2]
    # the filter weight has 8 fractional bits: a linear function is reproduced to 1/256 of a table step
    assert np.abs(out["dpd/tableforce/meso"] - out["dpd/fast/meso"]).max() < 2e-3 * np.abs(out["dpd/fast/meso"]).max()


def test_uniform_noise_thermostat():
    """uniform pair noise of unit variance: T = sigma^2 / (2 gamma) = 1"""
    from meso_amd.api import Meso
    x, v, lo, hi = make_box(16)
    with Meso() as m:
        _tab(m, x, v, lo, hi, None, 1, 3.0)
        m.run(600)
        t = []
        for _ in range(10):
            m.run(20)
            t.append(m.temperature())
    assert abs(np.mean(t) - 1.0) < 0.02, t


def test_script_driver_reads_a_table_file(oracle, tmp_path):
    from meso_amd.api import Meso
    from meso_amd.datagen import write_data
    x, v, lo, hi = make_box(6)
    write_data(str(tmp_path / "b.data"), x, lo, hi, v=v)
    (tmp_path / "fc.txt").write_text("\n".join("%.9g" % t for t in TABLES[1, 1]))
    deck = """dimension 3
units lj
boundary p p p
atom_style dpd/atomic/meso
neighbor 0.3 bin
neigh_modify delay 0 every 5 check no
read_data {data}
run_style mvv/meso
pair_style dpd/tableforce/meso 1.0 419084618 {L}
pair_coeff 1 1 4.5 0.0 {fc}
compute mobile all temp/meso
fix 1 all nve/meso
thermo_style custom step c_mobile
thermo 10
timestep 0.005
run 10
""".format(data=tmp_path / "b.data", fc=tmp_path / "fc.txt", L=L)
    (tmp_path / "t.run").write_text(deck)
    s = _oracle(x, v, lo, hi, None, 1, 0.0)
    s.run(10)
    with Meso() as m:
        m.script(str(tmp_path / "t.run"))
        xg = m.gather()[0]
    d = xg - s.x
    d -= np.round(d / (hi - lo)) * (hi - lo)
    assert np.abs(d).max() < 5e-5


def test_argument_checks():
    from meso_amd.api import Meso, MesoError
    x, v, types, _, lo, hi = make_polymer_box(6, frac=0.3)
    with Meso() as m:
        m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
        m.pair_style("dpd/tableforce/meso", 1.0, 1)
        m.pair_coeff_table(1, 1, 4.5, 3.0, TABLES[1, 1])
        with pytest.raises(MesoError):
            m.pair_coeff_table(1, 2, 4.5, 3.0, TABLES[1, 2][:20])       # all tables share table_length
        with pytest.raises(MesoError):
            m.pair_coeff(2, 2, 15.0, 4.5, 3.0, 1.0, 1.0)
