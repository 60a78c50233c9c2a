"""The bonded terms of the oracle (oracle/meso_sim.py: harmonic and FENE bonds, harmonic angles - what tests/test_gpu_bonds.py and
tests/test_gpu_angles.py compare the HIP kernels with) pinned against the REFERENCE ITSELF: src/MOLECULE/bond_harmonic.cpp:44,
bond_fene.cpp:50 and angle_harmonic.cpp:48 compiled unmodified into oracle/_ref/ref_bonded (oracle/build_ref.sh, driver
oracle/ref_bonded.cpp).  Both sides see the same coordinates (exactly representable in the device's fp32 merged frame), so the
only differences are the order of the reference's per-bond accumulation and its sqrt against the oracle's rsqrt form: forces to
1e-12 of the largest force, energies to 1e-13 relative.  Live where /root/reference is mounted, and against the committed fixture
tests/golden/ref_bonded.npz (written by tests/golden/make_ref_bonded.py) everywhere."""
import os

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "ref_bonded.npz")


def _oracle_bonded(x, L, bonds=None, bcoef=None, style="harmonic", angles=None, acoef=None):
    """Bond / angle forces and energies of oracle/meso_sim.py on the deck (pair coefficients zero)."""
    from oracle.meso_sim import MesoRefSim
    lo, hi = np.zeros(3), np.full(3, float(L))
    s = MesoRefSim(x, np.zeros_like(x), lo, hi)
    s.pair_coeff(1, 1, 0.0, 0.0, 0.0, 1.0, 1.0)
    if bonds is not None:
        tagged = np.column_stack([bonds[:, 0] + 1, bonds[:, 1] + 1, bonds[:, 2]])
        s.set_bonds(tagged, {t + 1: tuple(c) for t, c in enumerate(bcoef)}, special=(1.0, 1.0, 1.0), style=style)
    if angles is not None:
        tagged = np.column_stack([angles[:, 0] + 1, angles[:, 1] + 1, angles[:, 2] + 1, angles[:, 3]])
        s.set_angles(tagged, {t + 1: tuple(c) for t, c in enumerate(acoef)})
    s.setup()
    return s.f.copy(), getattr(s, "e_bond", 0.0), getattr(s, "e_angle", 0.0)


def _check(g, fb_h, eb_h, fb_f, eb_f, fa, ea):
    for name, f_ref, e_ref, kw in (("harm", fb_h, eb_h, dict(bonds=g["bonds"], bcoef=g["harm"], style="harmonic")),
                                   ("fene", fb_f, eb_f, dict(bonds=g["bonds"], bcoef=g["fene"], style="fene")),
                                   ("angle", fa, ea, dict(angles=g["angles"], acoef=g["ang"]))):
        f, eb, ean = _oracle_bonded(g["x"], float(g["L"]), **kw)
        e = ean if name == "angle" else eb
        scale = np.abs(f_ref).max()
        assert scale > 1.0
        assert np.abs(f - f_ref).max() <= 1e-12 * scale, (name, np.abs(f - f_ref).max() / scale)
        assert abs(e - e_ref) <= 1e-13 * abs(e_ref), (name, e, e_ref)
        assert np.abs(f.sum(0)).max() < 1e-9 * scale


def test_bonded_oracle_equals_the_reference_fixture(oracle):
    g = np.load(GOLD)
    _check(g, g["f_harm"], float(g["e_harm"]), g["f_fene"], float(g["e_fene"]), g["f_angle"], float(g["e_angle"]))


def test_bonded_oracle_equals_the_reference_live(oracle):
    from oracle import ref
    if not (os.path.isdir(os.path.join(ref.REFERENCE, "src")) and ref.build() and os.path.exists(ref.BIN_BONDED)):
        pytest.skip("reference sources not mounted")
    g = np.load(GOLD)
    fbh, ebh, fa, ea = ref.bonded(g["x"], g["bonds"], [tuple(c) for c in g["harm"]], "harmonic", g["angles"], [tuple(c) for c in g["ang"]])
    fbf, ebf, _, _ = ref.bonded(g["x"], g["bonds"], [tuple(c) for c in g["fene"]], "fene")
    # the fixture is what the reference produces today
    assert np.array_equal(fbh, g["f_harm"]) and np.array_equal(fbf, g["f_fene"]) and np.array_equal(fa, g["f_angle"])
    _check(g, fbh, ebh, fbf, ebf, fa, ea)
