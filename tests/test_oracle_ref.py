"""Pins oracle/lmp_dpd_cpu.c against the reference ITSELF.

oracle/_ref/ref_lmp is built by oracle/build_ref.sh from the reference's own unmodified sources (RanMars, RanPark,
PairDPD::compute, Neighbor::half_bin_newton, Comm::borders/forward/reverse, Domain::pbc, FixNVE ...; the list and what
is missing - Atom::sort - is in oracle/ref_harness.cpp).  With the thermostat ON the restatement has to reproduce the
reference's positions, velocities and forces BIT FOR BIT: that fixes the RanMars stream, the order in which
half_bin_newton visits pairs, the ghost order of Comm::borders and every arithmetic statement of the step.

Two layers: (1) against committed fixtures (tests/golden/ref_*.npz, written by tests/golden/make_ref_golden.py from
the reference run in this container) - runs anywhere; (2) live against oracle/_ref when it is present/buildable.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_ref_golden import CASES, inputs  # noqa: E402


def _restatement(oracle, c, x, v, lo, hi, types):
    ntypes = 1 if types is None else int(types.max())
    s = oracle.LmpDpd(x, lo, hi, types=types, ntypes=ntypes)
    for t, m in enumerate(c.get("mass", []), start=1):
        s.set_mass(t, m)
    s.pair_style(c["T"], c["cut"], c["seed"])
    for (i, j, a0, gamma, cut) in c["coeff"]:
        s.pair_coeff(i, j, a0, gamma, cut)
    s.set_velocities(v)
    s.neighbor(0.3, c["every"], 0)
    s.timestep(0.005)
    s.atom_modify_sort(0)
    s.setup()
    return s


def _compare(oracle, c, rec_iter):
    x, v, lo, hi, types = inputs(c)
    s = _restatement(oracle, c, x, v, lo, hi, types)
    done = 0
    for r in rec_iter:
        s.run(r["step"] - done)
        done = r["step"]
        xo, vo, fo = s.state()
        e, vir = s.ev()
        assert s.nghost == r["nghost"] and s.nneigh == r["nneigh"], r["step"]
        assert np.array_equal(xo, r["x"]), "positions differ at step %d" % r["step"]
        assert np.array_equal(vo, r["v"]), "velocities differ at step %d" % r["step"]
        assert np.array_equal(fo, r["f"]), "forces differ at step %d" % r["step"]
        assert e == pytest.approx(r["eng_vdwl"], rel=1e-14)
        # F.r virial: the reference sums it over atoms in storage order, the restatement per thread block
        assert np.allclose(vir, r["virial"], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("name", sorted(CASES))
def test_restatement_reproduces_reference_fixture_bit_for_bit(oracle, name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    recs = [dict(step=int(g["steps"][k]), nghost=int(g["nghost"][k]), nneigh=int(g["nneigh"][k]),
                 eng_vdwl=float(g["eng_vdwl"][k]), virial=g["virial"][k], x=g["x"][k], v=g["v"][k], f=g["f"][k])
            for k in range(len(g["steps"]))]
    x, v, lo, hi, types = inputs(CASES[name])
    assert np.array_equal(x, g["x0"]) and np.array_equal(v, g["v0"])      # the generator is deterministic
    _compare(oracle, CASES[name], recs)


def test_rng_streams_match_reference_fixture(oracle):
    g = np.load(os.path.join(GOLDEN, "ref_rng.npz"))
    for kind, seed in (("mars", 419084618), ("mars", 90210), ("park", 788662042), ("park", 1)):
        u, ga = oracle.rng_stream(kind, seed, 2000)
        assert np.array_equal(u, g["%s_%d_uniform" % (kind, seed)]), (kind, seed)
        assert np.array_equal(ga, g["%s_%d_gaussian" % (kind, seed)]), (kind, seed)


# ---------------------------------------------------------------- live against oracle/_ref
@pytest.fixture(scope="module")
def ref():
    from oracle import ref as r
    if not r.available():
        pytest.skip("oracle/_ref not built and /root/reference not mounted")
    return r


@pytest.mark.parametrize("kind,seed", [("mars", 1), ("mars", 419084618), ("mars", 900000000), ("park", 788662042),
                                       ("park", 12345)])
def test_rng_streams_match_reference_live(oracle, ref, kind, seed):
    u, g = ref.rng(kind, seed, 5000)
    uo, go = oracle.rng_stream(kind, seed, 5000)
    assert np.array_equal(u, uo) and np.array_equal(g, go)


def test_fixtures_are_what_the_reference_produces_now(oracle, ref):
    """The committed fixture equals a fresh run of the reference (guards against a stale or edited fixture)."""
    c = CASES["ref_lmp_L6"]
    x, v, lo, hi, types = inputs(c)
    recs = ref.run(x, v, lo, hi, nsteps=c["nsteps"], sample=c["sample"], T=c["T"], cut=c["cut"], seed=c["seed"],
                   coeff=c["coeff"], every=c["every"])
    g = np.load(os.path.join(GOLDEN, "ref_lmp_L6.npz"))
    for k, r in enumerate(recs):
        assert np.array_equal(r["x"], g["x"][k]) and np.array_equal(r["f"], g["f"][k])


def test_restatement_vs_reference_live_25_deck(oracle, ref):
    """example/simple/25.data (62 500 atoms), velocities from the restated `velocity create ... loop all`, thermostat
    on, 10 steps with two rebuilds: positions, velocities and forces identical to the reference's own code."""
    d = np.load(os.path.join(GOLDEN, "simple_25_positions.npz"))
    x, lo, hi = d["x"], d["lo"], d["hi"]
    s0 = oracle.LmpDpd(x, lo, hi)
    s0.velocity_create(1.0, 788662042)
    _, v, _ = s0.state()
    c = dict(T=1.0, cut=1.0, seed=419084618, coeff=[(1, 1, 15.0, 4.5, 0.0)], every=5)
    recs = ref.run(x, v, lo, hi, nsteps=10, sample=[0, 10], T=1.0, cut=1.0, seed=419084618, coeff=c["coeff"], every=5)
    s = _restatement(oracle, c, x, v, lo, hi, None)
    done = 0
    for r in recs:
        s.run(r["step"] - done)
        done = r["step"]
        xo, vo, fo = s.state()
        assert s.nghost == r["nghost"] and s.nneigh == r["nneigh"]
        assert np.array_equal(xo, r["x"]) and np.array_equal(vo, r["v"]) and np.array_equal(fo, r["f"])
        assert s.ev()[0] == pytest.approx(r["eng_vdwl"], rel=1e-13)
