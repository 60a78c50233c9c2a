"""Generates tests/golden/ref_bonded.npz from the REFERENCE ITSELF: oracle/_ref/ref_bonded = the reference's own
src/MOLECULE/bond_harmonic.cpp, bond_fene.cpp, angle_harmonic.cpp (+ src/bond.cpp, src/angle.cpp) compiled unmodified
(oracle/build_ref.sh).  Must run where /root/reference is mounted.  The fixture holds inputs (coordinates, topology,
coefficients) and the reference's outputs (forces, energies): data, not source."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref  # noqa: E402


def deck(seed=7, nchain=40, length=6, L=12.0):
    """Linear chains well inside [0, L]^3 (no bond or angle crosses the periodic boundary), bond lengths 0.55-0.95; coordinates
    are exactly representable in fp32 RELATIVE TO THE BOX CENTRE (what the device's merged float4 holds)."""
    rng = np.random.default_rng(seed)
    x, bonds, angles = [], [], []
    for c in range(nchain):
        p = rng.uniform(3.5, L - 3.5, 3)
        first = len(x)
        for k in range(length):
            x.append(p.copy())
            step = rng.normal(size=3)
            p = p + step / np.linalg.norm(step) * rng.uniform(0.55, 0.95)
        for k in range(length - 1):
            bonds.append((first + k, first + k + 1, 1 + (c + k) % 2))
        for k in range(length - 2):
            angles.append((first + k, first + k + 1, first + k + 2, 1 + c % 2))
    x = np.array(x)
    x = (x - 0.5 * L).astype(np.float32).astype(np.float64) + 0.5 * L
    return x, np.array(bonds, np.int32), np.array(angles, np.int32), L


def main():
    assert ref.build(), "reference sources not mounted"
    x, bonds, angles, L = deck()
    harm = [(50.0, 0.5), (80.0, 0.7)]
    fene = [(30.0, 1.5, 1.0, 0.8), (25.0, 1.6, 1.2, 0.7)]
    ang = [(20.0, 120.0), (35.0, 100.0)]
    fbh, ebh, fa, ea = ref.bonded(x, bonds, harm, "harmonic", angles, ang)
    fbf, ebf, _, _ = ref.bonded(x, bonds, fene, "fene")
    np.savez_compressed(os.path.join(HERE, "ref_bonded.npz"), x=x, L=L, bonds=bonds, angles=angles, harm=np.array(harm), fene=np.array(fene),
                        ang=np.array(ang), f_harm=fbh, e_harm=ebh, f_fene=fbf, e_fene=ebf, f_angle=fa, e_angle=ea)
    print("ref_bonded.npz written:", len(x), "atoms,", len(bonds), "bonds,", len(angles), "angles; E", ebh, ebf, ea)


if __name__ == "__main__":
    main()
