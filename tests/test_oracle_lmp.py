"""Pins oracle/lmp_dpd_cpu.c (restatement of the reference's stock CPU pair_style dpd step) against
outputs of the reference binary itself (SURVEY.md 8c / BASELINE.md 2) on example/simple/25.data."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "lmp_simple25_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def deck25():
    d = np.load(os.path.join(GOLDEN, "simple_25_positions.npz"))
    return d["x"], d["lo"], d["hi"]


def _make(oracle, deck25, T, nthreads=1):
    x, lo, hi = deck25
    s = oracle.LmpDpd(x, lo, hi, nthreads=nthreads)
    s.pair_style(T, 1.0, 419084618)
    s.pair_coeff(1, 1, 15.0, 4.5)
    s.velocity_create(1.0, 788662042)
    s.neighbor(0.3, 5, 0)
    s.timestep(0.005)
    s.setup()
    return s


def test_fixture_matches_reference_file(deck25):
    """The committed positions are the reference's own 25.data (checked when it is mounted)."""
    path = "/root/reference/example/simple/25.data"
    if not os.path.exists(path):
        pytest.skip("reference tree not mounted")
    from meso_amd.datagen import read_data
    x, v, types, ntypes, lo, hi = read_data(path)
    assert np.array_equal(x, deck25[0]) and ntypes == 1 and v is None
    assert np.array_equal(lo, deck25[1]) and np.array_equal(hi, deck25[2])


def test_sigma0_step0_thermo(oracle, deck25, gold):
    s = _make(oracle, deck25, 0.0)
    g = gold["sigma0"]
    assert abs(s.temperature - 1.0) < 1e-12
    # golden values carry 15 / 14 significant digits
    assert s.pe_per_atom == pytest.approx(g["step0_pe_per_atom"], rel=5e-14)
    assert s.pressure == pytest.approx(g["step0_press"], rel=5e-13)


def test_sigma0_temperature_trajectory(oracle, deck25, gold):
    s = _make(oracle, deck25, 0.0)
    for step in (10, 20, 30, 40, 50):
        s.run(10)
        assert s.temperature == pytest.approx(gold["sigma0"]["temp"][str(step)], rel=2e-13), step


def test_sigma3_step0_pressure(oracle, deck25, gold):
    # with the thermostat on, step-0 pressure includes the first RanMars draws in list order
    s = _make(oracle, deck25, 1.0)
    assert s.pressure == pytest.approx(gold["sigma3"]["step0_press"], abs=5e-6)


@pytest.mark.slow
def test_sigma3_temperature_overshoot(oracle, deck25, gold):
    s = _make(oracle, deck25, 1.0)
    s.run(100)
    assert s.temperature == pytest.approx(gold["sigma3"]["temp_step100"], abs=2e-3)


def test_openmp_mode_is_statistically_equivalent(oracle):
    """nthreads>1 (cpu_baseline mode) changes only summation order / RNG streams."""
    from meso_amd.datagen import make_box
    x, v, lo, hi = make_box(8)
    res = []
    for nt in (1, 4):
        s = oracle.LmpDpd(x, lo, hi, nthreads=nt)
        s.pair_style(0.0, 1.0, 12345)
        s.pair_coeff(1, 1, 15.0, 4.5)
        s.set_velocities(v)
        s.neighbor(0.3, 5, 0)
        s.setup()
        s.run(10)
        res.append(s.state())
    for a, b in zip(res[0], res[1]):
        assert np.allclose(a, b, rtol=0, atol=1e-9)


def test_ranmars_rejects_bad_seed(oracle, deck25):
    x, lo, hi = deck25
    s = oracle.LmpDpd(x[:100], lo, hi)
    with pytest.raises(ValueError):
        s.pair_style(1.0, 1.0, 0)
