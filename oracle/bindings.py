"""ctypes bindings for the CPU oracle libraries (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``meso_amd``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_vp, _i, _d, _f, _u = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_uint32


def build(force: bool = False) -> None:
    """Compile liboracle_lmp.so / liboracle_meso.so with gcc (oracle/Makefile)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle_lmp.so", "liboracle_meso.so"))
    if not need:
        for lib, src in (("liboracle_lmp.so", "lmp_dpd_cpu.c"), ("liboracle_meso.so", "meso_ref.c")):
            if os.path.getmtime(os.path.join(_HERE, src)) > os.path.getmtime(os.path.join(_HERE, lib)):
                need = True
    if need:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


_lmp = None
_meso = None


def lmp_lib():
    global _lmp
    if _lmp is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle_lmp.so"))
        L.lmp_create.restype = _vp
        L.lmp_create.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _i]
        L.lmp_destroy.argtypes = [_vp]
        L.lmp_set_velocities.argtypes = [_vp, _vp]
        L.lmp_set_mass.argtypes = [_vp, _i, _d]
        L.lmp_set_timestep.argtypes = [_vp, _d]
        L.lmp_set_neighbor.argtypes = [_vp, _d, _i, _i]
        L.lmp_set_sortfreq.argtypes = [_vp, _i]
        L.lmp_get_ev.argtypes = [_vp, _vp]
        L.lmp_ranmars_stream.argtypes = [_i, _i, _vp, _vp]
        L.lmp_ranpark_stream.argtypes = [_i, _i, _vp, _vp]
        L.lmp_pair_style_dpd.argtypes = [_vp, _d, _d, _i]
        L.lmp_pair_coeff.argtypes = [_vp, _i, _i, _d, _d, _d]
        L.lmp_velocity_create.argtypes = [_vp, _d, _i]
        L.lmp_setup.argtypes = [_vp]
        L.lmp_run.argtypes = [_vp, _i, _i]
        for fn in ("lmp_temperature", "lmp_pe_per_atom", "lmp_pressure"):
            getattr(L, fn).restype = _d
            getattr(L, fn).argtypes = [_vp]
        for fn in ("lmp_nlocal", "lmp_nghost", "lmp_nbuild"):
            getattr(L, fn).restype = _i
            getattr(L, fn).argtypes = [_vp]
        L.lmp_nneigh.restype = C.c_long
        L.lmp_nneigh.argtypes = [_vp]
        L.lmp_get_state.argtypes = [_vp, _vp, _vp, _vp]
        _lmp = L
    return _lmp


class LmpDpd:
    """Stock LAMMPS ``pair_style dpd`` + ``fix nve`` on one rank (oracle/lmp_dpd_cpu.c)."""

    def __init__(self, x, lo, hi, types=None, ntypes=1, nthreads=1):
        self.L = lmp_lib()
        x = np.ascontiguousarray(x, dtype=np.float64)
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        t = None if types is None else np.ascontiguousarray(types, dtype=np.int32)
        self.n = len(x)
        self.h = self.L.lmp_create(self.n, ntypes, _ptr(lo), _ptr(hi), _ptr(x), _ptr(t), nthreads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lmp_destroy(self.h)
            self.h = None

    def pair_style(self, T, cut, seed):
        if self.L.lmp_pair_style_dpd(self.h, T, cut, seed):
            raise ValueError("Illegal pair_style command")

    def pair_coeff(self, i, j, a0, gamma, cut=0.0):
        self.L.lmp_pair_coeff(self.h, i, j, a0, gamma, cut)

    def velocity_create(self, T, seed):
        self.L.lmp_velocity_create(self.h, T, seed)

    def set_velocities(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        self.L.lmp_set_velocities(self.h, _ptr(v))

    def neighbor(self, skin, every, delay=0):
        self.L.lmp_set_neighbor(self.h, skin, every, delay)

    def timestep(self, dt):
        self.L.lmp_set_timestep(self.h, dt)

    def set_mass(self, t, m):
        self.L.lmp_set_mass(self.h, t, m)

    def atom_modify_sort(self, freq):
        self.L.lmp_set_sortfreq(self.h, freq)

    def ev(self):
        """(eng_vdwl, virial[6]) of the last energy/virial step."""
        out = np.empty(7)
        self.L.lmp_get_ev(self.h, _ptr(out))
        return out[0], out[1:].copy()

    def setup(self):
        self.L.lmp_setup(self.h)

    def run(self, n, ev_last=True):
        self.L.lmp_run(self.h, n, 1 if ev_last else 0)

    temperature = property(lambda s: s.L.lmp_temperature(s.h))
    pe_per_atom = property(lambda s: s.L.lmp_pe_per_atom(s.h))
    pressure = property(lambda s: s.L.lmp_pressure(s.h))
    nghost = property(lambda s: s.L.lmp_nghost(s.h))
    nneigh = property(lambda s: s.L.lmp_nneigh(s.h))

    def state(self):
        x = np.empty((self.n, 3)); v = np.empty((self.n, 3)); f = np.empty((self.n, 3))
        self.L.lmp_get_state(self.h, _ptr(x), _ptr(v), _ptr(f))
        return x, v, f


def rng_stream(kind, seed, n):
    """n uniform() then n gaussian() draws of the restated RanMars ("mars") / RanPark ("park")."""
    L = lmp_lib()
    u, g = np.empty(n), np.empty(n)
    fn = L.lmp_ranmars_stream if kind == "mars" else L.lmp_ranpark_stream
    if fn(int(seed), int(n), _ptr(u), _ptr(g)):
        raise ValueError("invalid seed")
    return u, g


def meso_lib():
    global _meso
    if _meso is None:
        build()
        M = C.CDLL(os.path.join(_HERE, "liboracle_meso.so"))
        M.meso_tea_core.argtypes = [_i, _vp, _vp]
        M.meso_premix_tea.restype = _u
        M.meso_premix_tea.argtypes = [_i, _u, _u]
        M.meso_seed_now.restype = _u
        M.meso_seed_now.argtypes = [_i, C.c_int64]
        M.meso_signature.restype = _u
        M.meso_signature.argtypes = [_u, _i, _f, _f, _f]
        M.meso_mantissa.restype = _u
        M.meso_mantissa.argtypes = [_f, _f, _f]
        M.meso_morton_encode.restype = _u
        M.meso_morton_encode.argtypes = [_u, _u, _u]
        for fn in ("meso_rsqrt", "meso_sqrtd", "meso_rcp", "meso_cospi"):
            getattr(M, fn).restype = _d
            getattr(M, fn).argtypes = [_d]
        M.meso_powd.restype = _d
        M.meso_powd.argtypes = [_d, _d]
        M.meso_log2u.restype = _d
        M.meso_log2u.argtypes = [_u]
        M.meso_gaussian_tea.restype = _d
        M.meso_gaussian_tea.argtypes = [_u, _u]
        M.meso_gaussian_tea_fast.restype = _f
        M.meso_gaussian_tea_fast.argtypes = [_u, _u]
        M.meso_merge_xvt.argtypes = [_i] + [_vp] * 8 + [_d, _d, _d, _u, _vp, _vp]
        M.meso_neigh_full.restype = _i
        M.meso_neigh_full.argtypes = [_i, _i, _vp, _f, _vp, _vp, _i]
        M.meso_pair_dpd.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _d, _vp, _vp, _vp, _vp, _vp, _i]
        M.meso_pair_dpd_fast.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp]
        M.meso_pair_dpd_fast_rng.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _i]
        M.meso_pair_dpd_fast_rng.restype = None
        M.meso_logistic_noise.restype = _f
        M.meso_logistic_noise.argtypes = [_u, _u]
        M.meso_uniform_tea_fast.restype = _f
        M.meso_uniform_tea_fast.argtypes = [_u, _u]
        M.meso_nve_initial.argtypes = [_i] + [_vp] * 11 + [_d, _d, _i]
        M.meso_nve_final.argtypes = [_i] + [_vp] * 8 + [_d, _i]
        M.meso_sum_mv2.restype = _d
        M.meso_sum_mv2.argtypes = [_i, _vp, _vp, _vp, _vp, _vp, _i]
        M.meso_assign_bin_id.argtypes = [_i, _i] + [_vp] * 8
        M.meso_reorder_key.argtypes = [_i] + [_vp] * 8
        _meso = M
    return _meso


# --------------------------------------------------------------------------- helpers
def tea_core(rounds, v0, v1):
    a, b = _u(int(v0) & 0xFFFFFFFF), _u(int(v1) & 0xFFFFFFFF)
    meso_lib().meso_tea_core(rounds, C.byref(a), C.byref(b))
    return a.value, b.value


def merge_xvt(x, v, types, tags, center, seed):
    """(n,3) fp64 x,v -> coord4, veloc4 float32 (n,4) like gpu_merge_xvt."""
    M = meso_lib()
    n = len(x)
    cols = [np.ascontiguousarray(x[:, d]) for d in range(3)] + [np.ascontiguousarray(v[:, d]) for d in range(3)]
    types = np.ascontiguousarray(types, dtype=np.int32)
    tags = np.ascontiguousarray(tags, dtype=np.int32)
    c4 = np.empty((n, 4), dtype=np.float32)
    v4 = np.empty((n, 4), dtype=np.float32)
    M.meso_merge_xvt(n, *[_ptr(c) for c in cols], _ptr(types), _ptr(tags), float(center[0]),
                     float(center[1]), float(center[2]), int(seed) & 0xFFFFFFFF, _ptr(c4), _ptr(v4))
    return c4, v4


def neigh_full(nlocal, coord4, rc_tail, stride=160):
    """Full neighbour table (rows sorted by j) from merged fp32 coordinates."""
    M = meso_lib()
    nall = len(coord4)
    count = np.zeros(nlocal, dtype=np.int32)
    table = np.zeros((nlocal, stride), dtype=np.int32)
    rc2 = np.float32(np.float64(rc_tail) ** 2)
    maxlen = M.meso_neigh_full(nlocal, nall, _ptr(coord4), rc2, _ptr(count), _ptr(table), stride)
    return count, table, maxlen


def make_coeff(ntypes, entries, dtype=np.float64):
    """entries: {(i,j): (a0,gamma,sigma,expw,cut)} 1-based, symmetric fill; layout of
    pair_dpd_meso.h:15-24 -> [cut,cutsq,cutinv,expw,a0,gamma,sigma]."""
    cf = np.zeros((ntypes, ntypes, 7), dtype=np.float64)
    for (i, j), (a0, gamma, sigma, expw, cut) in entries.items():
        for a, b in ((i - 1, j - 1), (j - 1, i - 1)):
            cf[a, b] = [cut, cut * cut, 1.0 / cut, expw, a0, gamma, sigma]
    return np.ascontiguousarray(cf.reshape(-1), dtype=dtype)


def pair_dpd(nlocal, coord4, veloc4, count, table, coeff, ntypes, dt, fast=False, ev=False, rng=0, poly=None, ftab=None):
    """Forces on atoms [0,nlocal) (newton off, full list). Returns f (nlocal,3) [, e_pair, virial]."""
    M = meso_lib()
    fx = np.zeros(nlocal); fy = np.zeros(nlocal); fz = np.zeros(nlocal)
    stride = table.shape[1]
    count = np.ascontiguousarray(count, dtype=np.int32)
    table = np.ascontiguousarray(table, dtype=np.int32)
    if fast:
        cf = np.ascontiguousarray(coeff, dtype=np.float32)
        M.meso_pair_dpd_fast_rng(0, nlocal, _ptr(coord4), _ptr(veloc4), _ptr(count), _ptr(table), stride,
                                 _ptr(cf), ntypes, np.float32(1.0 / np.sqrt(dt)), _ptr(fx), _ptr(fy), _ptr(fz), int(rng),
                                 None if poly is None else _ptr(np.ascontiguousarray(poly, dtype=np.float32)),
                                 None if ftab is None else _ptr(np.ascontiguousarray(ftab, dtype=np.float32)),
                                 0 if ftab is None else int(np.asarray(ftab).shape[-1]))
        return np.stack([fx, fy, fz], axis=1)
    cf = np.ascontiguousarray(coeff, dtype=np.float64)
    e = np.zeros(nlocal) if ev else None
    vir = np.zeros((6, nlocal)) if ev else None
    M.meso_pair_dpd(0, nlocal, _ptr(coord4), _ptr(veloc4), _ptr(count), _ptr(table), stride, _ptr(cf),
                    ntypes, 1.0 / np.sqrt(dt), _ptr(fx), _ptr(fy), _ptr(fz), _ptr(e), _ptr(vir), nlocal)
    f = np.stack([fx, fy, fz], axis=1)
    return (f, e, vir) if ev else f
