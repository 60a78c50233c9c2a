"""Driver for oracle/_ref/ref_lmp - the reference's own unmodified CPU sources (TEST INFRASTRUCTURE ONLY).

`build()` runs oracle/build_ref.sh when /root/reference is present (this container); on the GPU box only the
prebuilt binary, which travels with the snapshot, is used.  Only tests/ and `__graft_entry__.build()` import this.
"""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(_HERE, "_ref", "ref_lmp")
BIN_BONDED = os.path.join(_HERE, "_ref", "ref_bonded")
REFERENCE = os.environ.get("MESO_REFERENCE", "/root/reference")


def build() -> bool:
    """Compile oracle/_ref/ref_lmp from the sources under /root/reference; False when neither sources nor binary exist."""
    if os.path.isdir(os.path.join(REFERENCE, "src")):
        subprocess.run(["sh", os.path.join(_HERE, "build_ref.sh")], check=True, stdout=subprocess.DEVNULL)
    return os.path.exists(BIN)


def available() -> bool:
    return os.path.exists(BIN) or build()


def rng(kind: str, seed: int, n: int):
    """n uniform() then n gaussian() draws of the reference's RanMars / RanPark (src/random_mars.cpp, random_park.cpp)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "o.bin")
        subprocess.run([BIN, "rng", kind, str(seed), str(n), out], check=True)
        a = np.fromfile(out, dtype=np.float64)
    return a[:n], a[n:]


def run(x, v, lo, hi, *, nsteps, sample, T=1.0, cut=1.0, seed=419084618, coeff=((1, 1, 15.0, 4.5, 0.0),), skin=0.3,
        every=5, dt=0.005, types=None, mass=None, timeout=600):
    """Stock `pair_style dpd` + `fix nve` of the reference itself (atom sorting off).  Returns a list of records
    {step, nlocal, nghost, nneigh, eng_vdwl, virial[6], x, v, f (by tag)} for the steps in `sample` (0 = after setup)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    n = len(x)
    ntypes = 1 if types is None else int(np.max(types))
    types = np.ones(n, dtype=np.int32) if types is None else np.ascontiguousarray(types, dtype=np.int32)
    mass = np.ones(ntypes) if mass is None else np.ascontiguousarray(mass, dtype=np.float64)
    cf = np.ascontiguousarray(coeff, dtype=np.float64).reshape(-1, 5)
    sample = np.ascontiguousarray(sorted(sample), dtype=np.int32)
    hdr = struct.pack("<8i10d", n, ntypes, nsteps, every, seed, len(cf), len(sample), 0,
                      *[float(t) for t in lo], *[float(t) for t in hi], T, cut, skin, dt)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "i.bin"), os.path.join(d, "o.bin")
        with open(fin, "wb") as f:
            f.write(hdr)
            for a in (x, v, types, mass, cf, sample):
                f.write(a.tobytes())
        subprocess.run([BIN, "run", fin, fout], check=True, timeout=timeout)
        raw = np.fromfile(fout, dtype=np.float64)
    recs = []
    per = 12 + 9 * n
    assert raw.size == per * len(sample), (raw.size, per, len(sample))
    for k in range(len(sample)):
        r = raw[k * per:(k + 1) * per]
        recs.append(dict(step=int(r[0]), nlocal=int(r[1]), nghost=int(r[2]), nneigh=int(r[3]), eng_vdwl=r[4],
                         virial=r[5:11].copy(), nbuild=int(r[11]), x=r[12:12 + 3 * n].reshape(n, 3).copy(),
                         v=r[12 + 3 * n:12 + 6 * n].reshape(n, 3).copy(), f=r[12 + 6 * n:].reshape(n, 3).copy()))
    return recs


def bonded(x, bonds=None, bond_coeffs=None, style="harmonic", angles=None, angle_coeffs=None):
    """The reference's own BondHarmonic / BondFENE / AngleHarmonic::compute (src/MOLECULE, compiled unmodified into
    oracle/_ref/ref_bonded) on coordinates x (no periodic images: keep the topology away from the boundary).
    bonds (nb,3: atom, atom, type; 0-based atoms, 1-based types); bond_coeffs [(k, r0)] or [(K, R0, epsilon, sigma)] per type;
    angles (na,4: atom, apex, atom, type); angle_coeffs [(K, theta0 in degrees)].
    Returns (f_bond, e_bond, f_angle, e_angle)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    bonds = np.zeros((0, 3), np.int32) if bonds is None else np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 3)
    angles = np.zeros((0, 4), np.int32) if angles is None else np.ascontiguousarray(angles, dtype=np.int32).reshape(-1, 4)
    bc = np.zeros((len(bond_coeffs or []), 4))
    for t, c in enumerate(bond_coeffs or []):
        bc[t, :len(c)] = c
    ac = np.ascontiguousarray(angle_coeffs if angle_coeffs else np.zeros((0, 2)), dtype=np.float64).reshape(-1, 2)
    hdr = struct.pack("<6i", n, len(bonds), len(angles), len(bc), len(ac), 0 if style == "harmonic" else 1)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "i.bin"), os.path.join(d, "o.bin")
        with open(fin, "wb") as f:
            f.write(hdr)
            for a in (x, bonds, bc, angles, ac):
                f.write(a.tobytes())
        subprocess.run([BIN_BONDED, fin, fout], check=True, timeout=120)
        raw = np.fromfile(fout, dtype=np.float64)
    assert raw.size == 2 * (3 * n + 1), raw.size
    fb, eb = raw[:3 * n].reshape(n, 3).copy(), float(raw[3 * n])
    fa, ea = raw[3 * n + 1:6 * n + 1].reshape(n, 3).copy(), float(raw[6 * n + 1])
    return fb, eb, fa, ea
