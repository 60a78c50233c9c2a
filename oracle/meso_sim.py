"""CPU mirror of one-rank ``run_style mvv/meso`` built from the oracle's C pieces
(TEST INFRASTRUCTURE ONLY - see oracle/meso_ref.c header; parity unpinned vs the CUDA path).

Follows ModifiedVerlet::setup/run (/root/reference/src/USER-MESO/mvv_meso.cu:139-219, 243-425)
for a periodic box on a single rank: ghosts are the periodic images inside
[lo-cutghost, hi+cutghost] (Comm::borders slab rule ``x >= lo && x <= hi``,
/root/reference/src/comm.cpp:994-998, shift added once per dimension as
pack_border_vel does, atom_vec_dpd_atomic_meso.cu:61-135).  Atom order is never
permuted here (the reference's reorder only changes summation order).
"""
from __future__ import annotations

import numpy as np

from . import bindings as ob


def periodic_ghosts(x, lo, hi, cut):
    """Return (src, shift) of ghost images: 3-stage scheme (x, then y incl. x-ghosts, then z)."""
    prd = hi - lo
    src = np.arange(len(x))
    sh = np.zeros((len(x), 3))
    pos = x.copy()
    nloc = len(x)
    for d in range(3):
        m_lo = pos[:, d] <= lo[d] + cut      # sent "down": appears above the high face
        m_hi = pos[:, d] >= hi[d] - cut
        add_src = np.concatenate([src[m_lo], src[m_hi]])
        s_lo = sh[m_lo].copy(); s_lo[:, d] = prd[d]
        s_hi = sh[m_hi].copy(); s_hi[:, d] = -prd[d]
        add_sh = np.concatenate([s_lo, s_hi])
        p_lo = pos[m_lo].copy(); p_lo[:, d] += prd[d]
        p_hi = pos[m_hi].copy(); p_hi[:, d] -= prd[d]
        src = np.concatenate([src, add_src])
        sh = np.concatenate([sh, add_sh])
        pos = np.concatenate([pos, p_lo, p_hi])
    return src[nloc:], sh[nloc:]


class MesoRefSim:
    def __init__(self, x, v, lo, hi, types=None, ntypes=1, mass=None, skin=0.3, every=5,
                 dt=0.005, seed=419084618, fast=False, stride=160, mini=False):
        self.n = len(x)
        self.x = np.array(x, dtype=np.float64)
        self.v = np.array(v, dtype=np.float64)
        self.f = np.zeros((self.n, 3))
        self.lo = np.array(lo, dtype=np.float64)
        self.hi = np.array(hi, dtype=np.float64)
        self.types = np.ones(self.n, np.int32) if types is None else np.array(types, np.int32)
        self.tags = np.arange(1, self.n + 1, dtype=np.int32)
        self.mask = np.ones(self.n, np.int32)
        self.ntypes = ntypes
        self.mass_type = np.ones(ntypes + 1) if mass is None else np.array(mass, dtype=np.float64)
        self.skin, self.every, self.dt, self.seed, self.fast = skin, every, dt, seed, fast or mini
        self.mini = mini            # pair_style dpd/mini/meso: fast arithmetic, logistic-map noise (pair_dpd_minimal_meso.cu)
        self.stride = stride
        self.coeffs = {}
        self.ntimestep = 0
        self.ago = 0
        self.M = ob.meso_lib()

    def pair_coeff(self, i, j, a0, gamma, sigma, expw=1.0, cut=1.0):
        self.coeffs[(min(i, j), max(i, j))] = (a0, gamma, sigma, expw, cut)

    def pair_coeff_poly(self, i, j, gamma, sigma, coeffs, cut=1.0):
        """pair_style dpd/polyforce/meso (construct with fast=True): conservative force polyval(1 - r/rc), coefficients from
        the highest order down (MesoPairDPDPolyForce::coeff pair_dpd_polyforce_meso.cu:290-335)"""
        self.coeffs[(min(i, j), max(i, j))] = (0.0, gamma, sigma, 1.0, cut)
        if getattr(self, "poly", None) is None:
            self.poly = np.zeros((self.ntypes, self.ntypes, 33), np.float32)
        for a, b in ((i - 1, j - 1), (j - 1, i - 1)):
            self.poly[a, b, 0] = len(coeffs) - 1
            self.poly[a, b, 1:1 + len(coeffs)] = coeffs

    def pair_coeff_table(self, i, j, gamma, sigma, table, cut=1.0):
        """pair_style dpd/tableforce/meso (construct with fast=True): tabulated conservative force over r/rc in [0,1], uniform
        TEA noise (MesoPairDPDTableForce::coeff pair_dpd_tableforce_meso.cu:306-356, kernel :171-184)"""
        self.coeffs[(min(i, j), max(i, j))] = (0.0, gamma, sigma, 1.0, cut)
        if getattr(self, "ftab", None) is None:
            self.ftab = np.zeros((self.ntypes, self.ntypes, len(table)), np.float32)
        for a, b in ((i - 1, j - 1), (j - 1, i - 1)):
            self.ftab[a, b] = table

    def set_bonds(self, bonds, coeffs, special=(0.0, 0.0, 0.0), style="harmonic"):
        """bonds (nb,3: tag_i, tag_j, type); coeffs {type: (k, r0)} (harmonic) or {type: (K, R0, epsilon, sigma)}
        (fene); special_bonds weights (0 = level excluded from the pair rows, gpu_filter_exclusion
        neigh_build_meso.cu:497-569).  Bond force: gpu_bond_harmonic bond_harmonic_meso.cu:84-101 == BondHarmonic::compute
        src/MOLECULE/bond_harmonic.cpp:44-96; gpu_bond_fene bond_fene_meso.cu:82-147 (BondFENE::compute
        src/MOLECULE/bond_fene.cpp:48-124 with the log argument clamped at 0.1)."""
        self.bond_style = style
        self.bonds = np.asarray(bonds, np.int64).reshape(-1, 3)
        self.bond_coeffs = coeffs
        adj = {}
        for a, b, _ in self.bonds:
            adj.setdefault(int(a), []).append(int(b))
            adj.setdefault(int(b), []).append(int(a))
        levels = 0
        while levels < 3 and special[levels] == 0.0:
            levels += 1
        self.special = {}
        for t in adj:
            seen, frontier, out = {t}, [t], []
            for _ in range(levels):
                nxt = []
                for u in frontier:
                    for w in adj[u]:
                        if w not in seen:
                            seen.add(w); nxt.append(w); out.append(w)
                frontier = nxt
            self.special[t] = out

    def set_angles(self, angles, coeffs):
        """angles (na,4: tag1, apex tag2, tag3, type); coeffs {type: (K, theta0 in degrees)}.  Force and energy:
        AngleHarmonic::compute src/MOLECULE/angle_harmonic.cpp:50-142 (E = K (theta - theta0)^2), which is what each of the
        three atoms evaluates for itself in gpu_angle_harmonic angle_harmonic_meso.cu:77-157."""
        self.angles = np.asarray(angles, np.int64).reshape(-1, 4)
        self.angle_coeffs = coeffs

    def _angle_forces(self):
        self.e_angle = 0.0
        if getattr(self, "angles", None) is None or len(self.angles) == 0:
            return
        c = self.c4[:self.n, :3].astype(np.float64)
        prd = self.hi - self.lo
        i1, i2, i3 = (self.angles[:, k] - 1 for k in range(3))
        k = np.array([self.angle_coeffs[int(t)][0] for t in self.angles[:, 3]])
        th0 = np.array([self.angle_coeffs[int(t)][1] for t in self.angles[:, 3]]) / 180.0 * np.pi

        def mi(d):
            return d + np.where(d > -0.5 * prd, np.where(d < 0.5 * prd, 0.0, -prd), prd)
        d1 = mi(c[i1] - c[i2])
        d2 = mi(c[i3] - c[i2])
        rsq1 = (d1 * d1).sum(1)
        rsq2 = (d2 * d2).sum(1)
        r1 = np.sqrt(rsq1)
        r2 = np.sqrt(rsq2)
        cs = np.clip((d1 * d2).sum(1) / (r1 * r2), -1.0, 1.0)
        sn = 1.0 / np.sqrt(np.maximum(1.0 - cs * cs, 0.001))          # SMALL (angle_harmonic.cpp:31)
        dth = np.arccos(cs) - th0
        tk = k * dth
        a = -2.0 * tk * sn
        a11 = a * cs / rsq1
        a12 = -a / (r1 * r2)
        a22 = a * cs / rsq2
        f1 = a11[:, None] * d1 + a12[:, None] * d2
        f3 = a22[:, None] * d2 + a12[:, None] * d1
        np.add.at(self.f, i1, f1)
        np.add.at(self.f, i3, f3)
        np.add.at(self.f, i2, -(f1 + f3))
        self.e_angle = float((tk * dth).sum())

    def _apply_exclusions(self):
        if not getattr(self, "special", None):
            return
        xa, va, ta, ga = self._all()
        for t, spec in self.special.items():
            i = t - 1
            if not spec:
                continue
            row = self.table[i, :self.count[i]]
            keep = ~np.isin(ga[row], spec)
            k = int(keep.sum())
            self.table[i, :k] = row[keep]
            self.count[i] = k

    def _bond_forces(self):
        self.e_bond = 0.0
        if getattr(self, "bonds", None) is None or len(self.bonds) == 0:
            return
        c = self.c4[:self.n, :3].astype(np.float64)
        prd = self.hi - self.lo
        i = np.concatenate([self.bonds[:, 0], self.bonds[:, 1]]) - 1
        j = np.concatenate([self.bonds[:, 1], self.bonds[:, 0]]) - 1
        bt = np.concatenate([self.bonds[:, 2], self.bonds[:, 2]])
        k = np.array([self.bond_coeffs[int(t)][0] for t in bt])
        r0 = np.array([self.bond_coeffs[int(t)][1] for t in bt])
        if getattr(self, "bond_style", "harmonic") == "fene":
            eps = np.array([self.bond_coeffs[int(t)][2] for t in bt])
            sig = np.array([self.bond_coeffs[int(t)][3] for t in bt])
            d = c[i] - c[j]                                                            # bond_fene_meso.cu:95-98: i - j
            d = d + np.where(d > -0.5 * prd, np.where(d < 0.5 * prd, 0.0, -prd), prd)
            rsq = (d * d).sum(1)
            r0sq = r0 * r0
            rlogarg = np.maximum(0.1, 1.0 - rsq / r0sq)                                # :103-109
            fb = -k / rlogarg
            e = -0.5 * k * r0sq * np.log(rlogarg)
            wca = rsq < 1.25992104989487316477 * sig * sig                             # :116
            sr2 = sig * sig / rsq
            sr6 = sr2 * sr2 * sr2
            fb = fb + np.where(wca, 48.0 * eps * sr6 * (sr6 - 0.5) / rsq, 0.0)
            e = e + np.where(wca, 4.0 * eps * sr6 * (sr6 - 1.0) + eps, 0.0)
            np.add.at(self.f, i, d * fb[:, None])
            self.e_bond = float(0.5 * e.sum())                                         # each bond stored twice, e * 0.5 (:145)
            return
        d = c[j] - c[i]
        d = d + np.where(d > -0.5 * prd, np.where(d < 0.5 * prd, 0.0, -prd), prd)     # minimum_image math_meso.h:148-152
        rsq = (d * d).sum(1)
        rinv = 1.0 / np.sqrt(rsq)
        r = rinv * rsq
        fb = 2.0 * k * (r - r0) * rinv
        np.add.at(self.f, i, d * fb[:, None])
        self.e_bond = float(0.5 * (k * (r - r0) ** 2).sum())

    # -- pieces -------------------------------------------------------------
    def _pbc(self):
        prd = self.hi - self.lo
        for d in range(3):
            c = self.x[:, d]
            c[c < self.lo[d]] += prd[d]
            m = c >= self.hi[d]
            c[m] -= prd[d]
            c[m] = np.maximum(c[m], self.lo[d])

    def _borders(self):
        self.gsrc, self.gshift = periodic_ghosts(self.x, self.lo, self.hi, self.cutghost)

    def _all(self):
        xa = np.concatenate([self.x, self.x[self.gsrc] + self.gshift])
        va = np.concatenate([self.v, self.v[self.gsrc]])
        ta = np.concatenate([self.types, self.types[self.gsrc]])
        ga = np.concatenate([self.tags, self.tags[self.gsrc]])
        return xa, va, ta, ga

    def _merge(self, seed):
        xa, va, ta, ga = self._all()
        center = 0.5 * (self.hi + self.lo)
        return ob.merge_xvt(xa, va, ta, ga, center, seed)

    def _build(self):
        c4, _ = self._merge(0)
        self.count, self.table, self.maxlen = ob.neigh_full(self.n, c4, self.cutmax + self.skin, self.stride)
        if self.maxlen > self.stride:
            raise RuntimeError("pair table overflow")
        self._apply_exclusions()
        self.ago = 0

    def _force(self):
        seed = self.M.meso_seed_now(self.seed, self.ntimestep)
        self.c4, self.v4 = self._merge(seed)
        self.f = ob.pair_dpd(self.n, self.c4, self.v4, self.count, self.table, self.coeff, self.ntypes,
                             self.dt, fast=self.fast, rng=1 if self.mini else (2 if getattr(self, "ftab", None) is not None else 0),
                             poly=getattr(self, "poly", None), ftab=getattr(self, "ftab", None))
        self._bond_forces()
        self._angle_forces()

    def setup(self):
        self.cutmax = max(c[4] for c in self.coeffs.values())
        self.cutghost = self.cutmax + self.skin
        self.coeff = ob.make_coeff(self.ntypes, self.coeffs)
        self.mass = self.mass_type[self.types]
        self._pbc()
        self._borders()
        self._build()
        self._force()

    def _cols(self, a):
        return [np.ascontiguousarray(a[:, d]) for d in range(3)]

    def step(self):
        M = self.M
        self.ntimestep += 1
        dtf, dtv = 0.5 * self.dt, self.dt
        xs, vs, fs = self._cols(self.x), self._cols(self.v), self._cols(self.f)
        P = ob._ptr
        M.meso_nve_initial(self.n, *[P(a) for a in xs], *[P(a) for a in vs], *[P(a) for a in fs],
                           P(self.mask), P(self.mass), dtf, dtv, 1)
        self.x = np.stack(xs, axis=1); self.v = np.stack(vs, axis=1)
        self.ago += 1
        if self.ago % self.every == 0:
            self._pbc()
            self._borders()
            self._build()
        self._force()
        vs, fs = self._cols(self.v), self._cols(self.f)
        M.meso_nve_final(self.n, *[P(a) for a in vs], *[P(a) for a in fs], P(self.mask), P(self.mass), dtf, 1)
        self.v = np.stack(vs, axis=1)

    def run(self, n):
        for _ in range(n):
            self.step()

    @property
    def temperature(self):
        vs = self._cols(self.v)
        P = ob._ptr
        return self.M.meso_sum_mv2(self.n, *[P(a) for a in vs], P(self.mass), P(self.mask), 1) / (3.0 * self.n - 3.0)
