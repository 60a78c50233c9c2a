"""Synthetic DPD input decks.

The reference ships only ``example/simple/25.data`` (``48.data``/``64.data`` are missing
blobs, ``/root/reference/.MISSING_LARGE_BLOBS``).  ``25.data`` is laid out as 4 atoms per unit
cell, cells x-fastest, each atom at cell + U[0,1)^3; this generator reproduces that layout
for any edge length so the 48^3 / 64^3 / 128^3 cases of ``README.md:27-35`` can be run.
"""
from __future__ import annotations

import numpy as np

RHO = 4


def make_positions(L: int, seed: int | None = None, rho: int = RHO) -> np.ndarray:
    """(N,3) float64 positions in [0,L)^3, N = rho*L^3, atom k (0-based) in cell k//rho."""
    rng = np.random.default_rng(L if seed is None else seed)
    n = rho * L ** 3
    c = np.arange(n) // rho
    cell = np.stack([c % L, (c // L) % L, c // (L * L)], axis=1).astype(np.float64)
    x = cell + rng.random((n, 3))
    # guard the (measure-zero) case cell+u == L after rounding
    return np.minimum(x, np.nextafter(float(L), 0.0))


def make_velocities(n: int, seed: int, temperature: float = 1.0, mass: float = 1.0) -> np.ndarray:
    """U(-0.5,0.5)^3, momentum zeroed, rescaled to T (dof = 3N-3) - SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    v = rng.random((n, 3)) - 0.5
    v -= v.mean(axis=0)
    t = mass * (v * v).sum() / (3.0 * n - 3.0)
    return v * np.sqrt(temperature / t)


def make_box(L: int, seed: int | None = None):
    """positions, velocities, box_lo, box_hi for the rho=4 cube of edge L."""
    x = make_positions(L, seed)
    v = make_velocities(len(x), (L if seed is None else seed) + 1000)
    return x, v, np.zeros(3), np.full(3, float(L))


def chain_angles(bonds: np.ndarray, atype: int = 1) -> np.ndarray:
    """Angles (na,4: tag1, apex, tag3, type) of linear chains: every pair of bonds that share a bead."""
    adj = {}
    for i, j, _ in np.asarray(bonds).reshape(-1, 3):
        adj.setdefault(int(i), []).append(int(j))
        adj.setdefault(int(j), []).append(int(i))
    out = []
    for apex in sorted(adj):
        nb = sorted(adj[apex])
        for a in range(len(nb)):
            for b in range(a + 1, len(nb)):
                out.append((nb[a], apex, nb[b], atype))
    return np.array(out, dtype=np.int32).reshape(-1, 4)


def write_data(path: str, x: np.ndarray, lo, hi, v: np.ndarray | None = None,
               types: np.ndarray | None = None, ntypes: int = 1, bonds: np.ndarray | None = None,
               angles: np.ndarray | None = None) -> None:
    """LAMMPS ``read_data`` file (atom_style atomic / dpd/atomic/meso), optional Velocities.  With ``bonds``
    (nb,3: tag_i, tag_j, type) the file is in atom_style bond / dpd/bond/meso form (molecule-id column, Bonds); with
    ``angles`` (na,4: tag1, apex, tag3, type) it also carries the Angles section of atom_style angle / dpd/angle/meso."""
    n = len(x)
    if types is None:
        types = np.ones(n, dtype=np.int64)
    with open(path, "w") as f:
        f.write("LAMMPS\n\n%d atoms\n" % n)
        if bonds is not None:
            f.write("%d bonds\n" % len(bonds))
        if angles is not None:
            f.write("%d angles\n" % len(angles))
        f.write("\n%d atom types\n" % ntypes)
        if bonds is not None:
            f.write("%d bond types\n" % int(bonds[:, 2].max()))
        if angles is not None:
            f.write("%d angle types\n" % int(angles[:, 3].max()))
        f.write("\n")
        for d, a in enumerate("xyz"):
            f.write("%.17g %.17g %slo %shi\n" % (lo[d], hi[d], a, a))
        f.write("\nMasses\n\n")
        for t in range(1, ntypes + 1):
            f.write("%d 1.000000\n" % t)
        f.write("\nAtoms\n\n")
        for i in range(n):
            if bonds is not None:
                f.write("%d 0 %d %.17g %.17g %.17g\n" % (i + 1, types[i], x[i, 0], x[i, 1], x[i, 2]))
            else:
                f.write("%d %d %.17g %.17g %.17g\n" % (i + 1, types[i], x[i, 0], x[i, 1], x[i, 2]))
        if v is not None:
            f.write("\nVelocities\n\n")
            for i in range(n):
                f.write("%d %.17g %.17g %.17g\n" % (i + 1, v[i, 0], v[i, 1], v[i, 2]))
        if bonds is not None:
            f.write("\nBonds\n\n")
            for b, (i, j, t) in enumerate(bonds):
                f.write("%d %d %d %d\n" % (b + 1, t, i, j))
        if angles is not None:
            f.write("\nAngles\n\n")
            for a, (i, j, k, t) in enumerate(angles):
                f.write("%d %d %d %d %d\n" % (a + 1, t, i, j, k))


def read_data(path: str):
    """Minimal reader for the files above and the reference's 25.data."""
    with open(path) as f:
        lines = [ln.strip() for ln in f]
    n = ntypes = 0
    lo, hi = np.zeros(3), np.zeros(3)
    for ln in lines[:40]:
        w = ln.split()
        if len(w) == 2 and w[1] == "atoms":
            n = int(w[0])
        elif len(w) == 3 and w[1] == "atom" and w[2] == "types":
            ntypes = int(w[0])
        elif len(w) == 4 and w[2] in ("xlo", "ylo", "zlo"):
            d = "xyz".index(w[2][0])
            lo[d], hi[d] = float(w[0]), float(w[1])
    ia = lines.index("Atoms")
    rows = [ln.split() for ln in lines[ia + 1:] if ln][:n]
    arr = np.array(rows, dtype=np.float64)
    order = np.argsort(arr[:, 0].astype(np.int64), kind="stable")
    arr = arr[order]
    x = np.ascontiguousarray(arr[:, 2:5])
    types = arr[:, 1].astype(np.int32)
    v = None
    if "Velocities" in lines:
        iv = lines.index("Velocities")
        rows = [ln.split() for ln in lines[iv + 1:] if ln][:n]
        va = np.array(rows, dtype=np.float64)
        va = va[np.argsort(va[:, 0].astype(np.int64), kind="stable")]
        v = np.ascontiguousarray(va[:, 1:4])
    return x, v, types, ntypes, lo, hi


def make_polymer_box(L: int, frac: float = 0.1, seed: int | None = None, chain=(1, 1, 2, 2, 2, 2), r0: float = 0.5):
    """configs[4] deck (build-defined, SURVEY.md 8d: the reference ships no polymer input): a rho=4 cube in which
    ``frac`` of the beads form linear A2B4 amphiphiles (types 1,1,2,2,2,2, consecutive ids, random-walk chains of
    step r0), the rest is type-1 solvent.  Returns x, v, types, bonds(nb,3: tag_i, tag_j, type), lo, hi."""
    rng = np.random.default_rng((L if seed is None else seed) + 5000)
    n = RHO * L ** 3
    m = len(chain)
    nchain = int(frac * n / m)
    x = rng.random((n, 3)) * L
    types = np.ones(n, dtype=np.int32)
    bonds = []
    for c in range(nchain):
        base = c * m
        for k in range(1, m):
            step = rng.normal(size=3)
            step *= r0 / np.linalg.norm(step)
            x[base + k] = x[base + k - 1] + step
            bonds.append((base + k, base + k + 1, 1))
        types[base:base + m] = chain
    x %= L
    x = np.minimum(x, np.nextafter(float(L), 0.0))
    v = make_velocities(n, (L if seed is None else seed) + 6000)
    return x, v, types, np.array(bonds, dtype=np.int32).reshape(-1, 3), np.zeros(3), np.full(3, float(L))
