"""Random decks against the CPU restatement of the USER-MESO step (oracle/meso_sim.py: the mirror of mvv/meso with the reference's
fp64 / fp32 arithmetic; see its header for the reference lines it follows).

The fixed decks of the other parity tests are cubes of one mass.  Here: boxes of three different edges (none a multiple of the bin
width), densities 3 / 4 / 6, one to three atom types of DIFFERENT masses with cross coefficients, rebuild intervals 1 / 2 / 5 - the
geometry and the per-type paths (mass table of the step boundary, coefficient table of the force kernel, image booking of the count,
ghost tiles in the gather's launch for the largest deck) that a cube of one type does not reach.  tools/fuzz_paths.py runs many more
such decks through the default path against the engine's own plain path, bit for bit; this test pins a few to the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Meso():
    from meso_amd.api import Meso
    return Meso


def _deck(seed, nmax):
    rng = np.random.default_rng(seed)
    dims = rng.integers(5, 14, 3).astype(float) + rng.random(3)
    rho = float(rng.choice([3.0, 4.0, 6.0]))
    n = int(rho * dims.prod())
    if n > nmax:
        dims *= (nmax / n) ** (1 / 3)
        n = int(rho * dims.prod())
    x = rng.random((n, 3)) * dims
    v = rng.random((n, 3)) - 0.5
    v -= v.mean(0)
    v *= np.sqrt(1.0 / ((v * v).sum() / (3 * n - 3)))
    ntypes = int(rng.integers(1, 4))
    types = rng.integers(1, ntypes + 1, n).astype(np.int32)
    masses = np.concatenate([[0.0], 0.5 + 2.0 * rng.random(ntypes)])
    every = int(rng.choice([1, 2, 5]))
    return x, v, dims, types, ntypes, masses, every


@pytest.mark.parametrize("seed,nmax,style", [(1, 6000, "dpd/meso"), (2, 6000, "dpd/meso"), (3, 6000, "dpd/fast/meso"), (4, 6000, "dpd/meso"),
                                              (5, 60000, "dpd/meso")])
def test_random_deck_against_the_oracle(Meso, oracle, seed, nmax, style):
    from oracle.meso_sim import MesoRefSim
    x, v, dims, types, ntypes, masses, every = _deck(seed, nmax)
    fast = style != "dpd/meso"
    steps = 1 if fast else 11          # (fp32 style with the thermostat on: one step, see test_trajectory_vs_meso_oracle)
    s = MesoRefSim(x, v, np.zeros(3), dims, types=types, ntypes=ntypes, mass=masses, every=every, dt=0.004, fast=fast, stride=64 * 6)
    m = Meso()
    m.read_atoms(x, v, np.zeros(3), dims, types=types, ntypes=ntypes, masses=masses)
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=every, check=False)
    m.pair_style(style, 1.0, 419084618)
    for i in range(1, ntypes + 1):
        for j in range(i, ntypes + 1):
            a0 = 15.0 if i == j else 30.0
            s.pair_coeff(i, j, a0, 4.5, 3.0, 1.0, 1.0)
            m.pair_coeff(i, j, a0, 4.5, 3.0, 1.0, 1.0)
    m.timestep(0.004)
    s.setup()
    m.setup()
    # neighbour counts and setup forces
    count, _ = m.neigh_table()
    tag = m.gather(by_tag=False)[3]
    assert np.array_equal(count[np.argsort(tag)], s.count)
    f0 = m.gather()[2]
    assert np.abs(f0 - s.f).max() < (2e-3 if fast else 5e-9) * np.abs(s.f).max()
    m.run(steps)
    s.run(steps)
    xg, vg, fg = m.gather()[:3]
    d = xg - s.x
    d -= np.round(d / dims) * dims
    tol = 2e-5 if fast else 1e-9
    assert np.abs(d).max() < tol and np.abs(vg - s.v).max() < tol * 50
    assert m.temperature() == pytest.approx(s.temperature, rel=1e-4 if fast else 1e-9)
    m.close()
