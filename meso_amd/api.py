"""Host-side mirror of the reference's style interface for the DPD hot path.

The method names are the LAMMPS commands / virtuals of the USER-MESO styles so that tests read
like the reference's decks (``example/simple/dp.run``):

    m = Meso()
    m.read_atoms(x, v, box_lo, box_hi)           # read_data
    m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/meso", 1.0, 419084618)     # pair_dpd_meso.cu:272-288
    m.pair_coeff(1, 1, 15, 4.5, 3.0, 1.0, 1.0)   # pair_dpd_meso.cu:290-327
    m.fix_nve(); m.timestep(0.005)
    m.run(1000)

Everything is a thin ctypes call into libmeso_hip.so; no arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

PAIR_STYLES = {"dpd/meso": 0, "dpd/fast/meso": 1, "dpd/mini/meso": 2, "dpd/polyforce/meso": 3, "dpd/tableforce/meso": 4}
RANGES = {"local": 0, "bulk": 1, "border": 2}
TRANSPORTS = {"self": 0, "rccl": 1, "host": 2, "local": 3}


def procgrid(nranks, prd):
    """Brick processor grid of minimal surface (host-only helper; no GPU needed)."""
    lib = _lib.load()
    p = np.ascontiguousarray(prd, np.float64)
    out = np.zeros(3, np.int32)
    if lib.meso_decomp_procgrid(int(nranks), _p(p), _p(out)):
        raise MesoError(lib.meso_last_error().decode())
    return tuple(int(v) for v in out)


def decomp_plan(lo, hi, grid, rank, cutghost, periodic=(1, 1, 1)):
    """The engine's own decomposition tables for `rank` of a brick grid (host-only helper; no GPU needed): dict with sublo,
    subhi, slab_lo, slab_hi (3), peer, active (27), shift, center (27, 3); direction d = (sx+1) + 3(sy+1) + 9(sz+1)."""
    lib = _lib.load()
    lo = np.ascontiguousarray(lo, np.float64); hi = np.ascontiguousarray(hi, np.float64)
    per = np.ascontiguousarray(periodic, np.int32); pg = np.ascontiguousarray(grid, np.int32)
    out = {k: np.zeros(3) for k in ("sublo", "subhi", "slab_lo", "slab_hi")}
    out["peer"] = np.zeros(27, np.int32); out["active"] = np.zeros(27, np.int32)
    out["shift"] = np.zeros((27, 3)); out["center"] = np.zeros((27, 3))
    if lib.meso_decomp_plan(_p(lo), _p(hi), _p(per), _p(pg), int(rank), float(cutghost), _p(out["sublo"]), _p(out["subhi"]),
                            _p(out["slab_lo"]), _p(out["slab_hi"]), _p(out["peer"]), _p(out["active"]), _p(out["shift"]),
                            _p(out["center"])):
        raise MesoError(lib.meso_last_error().decode())
    return out


def nccl_unique_id():
    """128-byte RCCL unique id (call on rank 0, broadcast to the other ranks)."""
    lib = _lib.load()
    buf = np.zeros(128, np.uint8)
    if lib.meso_comm_get_unique_id(_p(buf), 128):
        raise MesoError(lib.meso_last_error().decode())
    return buf


class MesoError(RuntimeError):
    pass


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Meso:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = C.c_void_p()
        self._h = None
        self._ck(self.lib.meso_init(device, C.byref(h)))
        self._h = h
        self._setup_done = False
        self._skin, self._every, self._delay, self._check = 0.3, 1, 10, True
        self._cb = None

    # -- plumbing ---------------------------------------------------------------------------
    def _ck(self, rc):
        if rc:
            raise MesoError(self.lib.meso_last_error().decode())

    def close(self):
        if self._h is not None:
            self.lib.meso_finalize(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_option(self, key, value):
        self._ck(self.lib.meso_set_option(self._h, key.encode(), float(value)))

    # -- decomposition (one rank per GPU; call before read_atoms) ------------------------------
    def comm_init(self, nranks, rank, grid, transport="rccl", uid=None):
        g = np.ascontiguousarray(grid, np.int32)
        u = None if uid is None else np.ascontiguousarray(uid, np.uint8)
        self._ck(self.lib.meso_comm_init(self._h, nranks, rank, _p(g), TRANSPORTS[transport], _p(u),
                                         0 if u is None else len(u)))

    def set_host_exchange(self, fn):
        """transport "host": fn is a _lib.HOST_EXCHANGE_FN (e.g. hostxchg.make_exchange); a reference is kept here"""
        self._host_fn = fn
        self._ck(self.lib.meso_comm_set_host_exchange(self._h, fn, None))

    # -- read_data / create atoms -----------------------------------------------------------
    def read_atoms(self, x, v, box_lo, box_hi, types=None, tags=None, masses=None, ntypes=None,
                   periodicity=(1, 1, 1)):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        n = len(x)
        types = np.ones(n, np.int32) if types is None else np.ascontiguousarray(types, np.int32)
        tags = np.arange(1, n + 1, dtype=np.int32) if tags is None else np.ascontiguousarray(tags, np.int32)
        ntypes = int(types.max()) if ntypes is None else ntypes
        masses = np.ones(ntypes + 1) if masses is None else np.ascontiguousarray(masses, np.float64)
        lo = np.ascontiguousarray(box_lo, np.float64)
        hi = np.ascontiguousarray(box_hi, np.float64)
        per = np.ascontiguousarray(periodicity, np.int32)
        self._ck(self.lib.meso_set_box(self._h, _p(lo), _p(hi), _p(per)))
        self._ck(self.lib.meso_set_mass(self._h, ntypes, _p(masses)))
        self._ck(self.lib.meso_atoms_upload(self._h, n, _p(x), _p(v), _p(tags), _p(types), None, None))
        self.natoms = n   # atoms in the deck; with several ranks each keeps only its sub-box (see counts())
        self._setup_done = False

    # -- neighbor / neigh_modify ------------------------------------------------------------
    def neighbor(self, skin, style="bin"):
        if style != "bin":
            raise MesoError("Illegal neighbor command")
        self._skin = skin
        self._push_neigh()

    def neigh_modify(self, delay=None, every=None, check=None):
        if delay is not None:
            self._delay = delay
        if every is not None:
            self._every = every
        if check is not None:
            self._check = bool(check)
        self._push_neigh()

    def _push_neigh(self):
        self._ck(self.lib.meso_neighbor(self._h, self._skin, self._every, self._delay, int(self._check)))

    # -- styles -----------------------------------------------------------------------------
    def pair_style(self, style, cut_global, seed):
        if style not in PAIR_STYLES:
            raise MesoError("Unknown pair style " + style)
        self._ck(self.lib.meso_pair_dpd_settings(self._h, PAIR_STYLES[style], cut_global, seed))

    def pair_coeff(self, i, j, a0, gamma, sigma, expw=1.0, cut=0.0):
        self._ck(self.lib.meso_pair_dpd_coeff(self._h, i, j, a0, gamma, sigma, expw, cut))

    def pair_coeff_poly(self, i, j, gamma, sigma, coeffs):
        """dpd/polyforce/meso: pair_coeff i j gamma sigma order c_order ... c_0 (coeffs from the highest order down)"""
        c = np.ascontiguousarray(coeffs, np.float64)
        self._ck(self.lib.meso_pair_dpd_polyforce_coeff(self._h, i, j, gamma, sigma, len(c) - 1, _p(c)))

    def pair_coeff_table(self, i, j, gamma, sigma, table):
        """dpd/tableforce/meso: pair_coeff i j gamma sigma <table>: conservative force at table_length points uniform in r/rc"""
        t = np.ascontiguousarray(table, np.float64)
        self._ck(self.lib.meso_pair_dpd_tableforce_coeff(self._h, i, j, gamma, sigma, len(t), _p(t)))

    # -- bonded topology: atom_style dpd/bond/meso, bond_style harmonic/meso ---------------------
    def special_bonds(self, w12=0.0, w13=0.0, w14=0.0):
        self._ck(self.lib.meso_special_bonds(self._h, w12, w13, w14))

    def read_bonds(self, bonds):
        """bonds: (nb,3) int array of (tag_i, tag_j, type) - the Bonds section of the data file."""
        b = np.ascontiguousarray(bonds, np.int32).reshape(-1, 3)
        ti, tj, bt = (np.ascontiguousarray(b[:, k]) for k in range(3))
        self._ck(self.lib.meso_bonds_upload(self._h, len(b), _p(ti), _p(tj), _p(bt)))
        self._setup_done = False

    def bond_style(self, style, nbondtypes):
        if style == "harmonic/meso":
            self._ck(self.lib.meso_bond_style_harmonic(self._h, nbondtypes))
        elif style == "fene/meso":
            self._ck(self.lib.meso_bond_style_fene(self._h, nbondtypes))
        else:
            raise MesoError("Unknown bond style " + style)
        self._bond_style = style

    def bond_coeff(self, btype, k, r0, epsilon=None, sigma=None):
        """bond_coeff type K r0 (harmonic/meso) or type K R0 epsilon sigma (fene/meso)"""
        if getattr(self, "_bond_style", "harmonic/meso") == "fene/meso":
            if epsilon is None or sigma is None:
                raise MesoError("Incorrect args for bond coefficients")
            self._ck(self.lib.meso_bond_coeff_fene(self._h, btype, k, r0, epsilon, sigma))
        else:
            if epsilon is not None or sigma is not None:
                raise MesoError("Incorrect args for bond coefficients")
            self._ck(self.lib.meso_bond_coeff(self._h, btype, k, r0))

    def bond_compute(self, eflag=0):
        self._ck(self.lib.meso_bond_compute(self._h, eflag))

    def ebond(self):
        t = C.c_double()
        self._ck(self.lib.meso_compute_ebond(self._h, C.byref(t)))
        return t.value

    def read_angles(self, angles):
        """angles: (na,4) int array of (tag1, tag2 = apex, tag3, type) - the Angles section of the data file (after read_bonds)."""
        a = np.ascontiguousarray(angles, np.int32).reshape(-1, 4)
        c = [np.ascontiguousarray(a[:, k]) for k in range(4)]
        self._ck(self.lib.meso_angles_upload(self._h, len(a), _p(c[0]), _p(c[1]), _p(c[2]), _p(c[3])))
        self._setup_done = False

    def angle_style(self, style, nangletypes):
        if style != "harmonic/meso":
            raise MesoError("Unknown angle style " + style)
        self._ck(self.lib.meso_angle_style_harmonic(self._h, nangletypes))

    def angle_coeff(self, atype, k, theta0_degrees):
        self._ck(self.lib.meso_angle_coeff(self._h, atype, k, theta0_degrees))

    def angle_compute(self, eflag=0):
        self._ck(self.lib.meso_angle_compute(self._h, eflag))

    def eangle(self):
        t = C.c_double()
        self._ck(self.lib.meso_compute_eangle(self._h, C.byref(t)))
        return t.value

    def fix_nve(self):
        pass  # fix nve/meso is the only integrator fix on this path; always active

    def timestep(self, dt):
        self._ck(self.lib.meso_timestep(self._h, dt))

    # -- run_style mvv/meso -----------------------------------------------------------------
    def setup(self):
        self._push_neigh()
        self._ck(self.lib.meso_setup(self._h))
        self._setup_done = True

    def run(self, nsteps):
        if not self._setup_done:
            self.setup()
        self._ck(self.lib.meso_run(self._h, nsteps))

    def sync(self):
        self._ck(self.lib.meso_device_sync(self._h))

    # individual virtuals
    def initial_integrate(self):
        self._ck(self.lib.meso_nve_initial(self._h))

    def final_integrate(self):
        self._ck(self.lib.meso_nve_final(self._h))

    def decide(self):
        r = C.c_int()
        self._ck(self.lib.meso_neighbor_decide(self._h, C.byref(r)))
        return bool(r.value)

    def reneighbor(self):
        self._ck(self.lib.meso_reneighbor(self._h))

    def forward_comm(self):
        self._ck(self.lib.meso_halo_forward(self._h))

    def force_clear(self, which="local"):
        self._ck(self.lib.meso_force_clear(self._h, RANGES[which]))

    def compute(self, eflag=0, vflag=0, which="local"):
        self._ck(self.lib.meso_pair_compute(self._h, RANGES[which], eflag, vflag))

    def step_advance(self, ntimestep):
        self._ck(self.lib.meso_step_advance(self._h, ntimestep))

    @property
    def ntimestep(self):
        return self.lib.meso_ntimestep(self._h)

    # -- computes ---------------------------------------------------------------------------
    def temperature(self):
        t = C.c_double()
        self._ck(self.lib.meso_compute_temp(self._h, C.byref(t)))
        return t.value

    def pe(self):
        t = C.c_double()
        self._ck(self.lib.meso_compute_pe(self._h, C.byref(t)))
        return t.value

    def tally(self):
        """Energy and virial at the current configuration, forces untouched (a thermo output between two run() calls)."""
        self._ck(self.lib.meso_tally_ev(self._h))

    def pressure(self):
        t = C.c_double()
        self._ck(self.lib.meso_compute_pressure(self._h, C.byref(t)))
        return t.value

    # -- data access ------------------------------------------------------------------------
    def counts(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.lib.meso_atoms_count(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def gather(self, by_tag=True):
        """x, v, f, tag, type of the local atoms (device order, or sorted by tag)."""
        n = self.counts()[0]
        x = np.empty((n, 3)); v = np.empty((n, 3)); f = np.empty((n, 3))
        tag = np.empty(n, np.int32); typ = np.empty(n, np.int32)
        self._ck(self.lib.meso_atoms_download(self._h, _p(x), _p(v), _p(f), _p(tag), _p(typ), None))
        if by_tag:
            o = np.argsort(tag, kind="stable")
            return x[o], v[o], f[o], tag[o], typ[o]
        return x, v, f, tag, typ

    def neigh_info(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_double(), C.c_int64()
        self._ck(self.lib.meso_neigh_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"n_col": a.value, "max_count": b.value, "avg_count": c.value, "nbuild": d.value}

    def neigh_table(self, stride=None):
        n = self.counts()[0]
        stride = self.neigh_info()["n_col"] if stride is None else stride
        count = np.zeros(n, np.int32)
        table = np.zeros((n, stride), np.int32)
        self._ck(self.lib.meso_neigh_download(self._h, _p(count), _p(table), stride))
        return count, table

    def neigh_parts(self, raw=False, stride=None):
        """Row sections of the table in use (meso_neigh_parts); raw=True: also the two sections as stored, padding included."""
        t, g = C.c_int(), C.c_int()
        n = self.counts()[0]
        nf, nb = np.zeros(n, np.int32), np.zeros(n, np.int32)
        stride = self.neigh_info()["n_col"] if stride is None else stride
        front = np.full((n, stride), -1, np.int32) if raw else None
        back = np.full((n, stride), -1, np.int32) if raw else None
        self._ck(self.lib.meso_neigh_parts(self._h, C.byref(t), C.byref(g), _p(nf), _p(nb), _p(front) if raw else None,
                                           _p(back) if raw else None, stride))
        out = {"parted": bool(t.value), "group": g.value, "nfront": nf, "nback": nb}
        if raw:
            out["front"] = front
            out["back"] = back
        return out

    def merged(self):
        nl, ng, _ = self.counts()
        c4 = np.empty((nl + ng, 4), np.float32)
        v4 = np.empty((nl + ng, 4), np.float32)
        self._ck(self.lib.meso_merged_download(self._h, _p(c4), _p(v4), nl + ng))
        return c4, v4

    def timer_reset(self):
        self._ck(self.lib.meso_timer_reset(self._h))

    def comm_count(self):
        n = C.c_int()
        self._ck(self.lib.meso_comm_count(self._h, C.byref(n)))
        return n.value

    def xchg_stats(self):
        """{kind of exchange: dict(calls, ms_device, ms_wire, ms_back, bytes)} of the host / in-process transports (profile on)."""
        buf = C.create_string_buffer(4096)
        self._ck(self.lib.meso_xchg_stats(self._h, buf, 4096))
        out = {}
        for ln in buf.value.decode().split("\n"):
            if ln:
                what, calls, a, b, c, by = ln.split("|")
                out[what] = dict(calls=int(calls), ms_device=float(a), ms_wire=float(b), ms_back=float(c), bytes=float(by))
        return out

    def pair_kernel_name(self):
        """Instantiation of the force kernel the last launch ran, as rocprofv3 prints it."""
        buf = C.create_string_buffer(160)
        self._ck(self.lib.meso_pair_kernel_name(self._h, buf, 160))
        return buf.value.decode()

    def membw_probe(self, nbytes=1 << 30, reps=5):
        """Measured float4 copy rate of this GPU in GB/s (read + write)."""
        g = C.c_double()
        self._ck(self.lib.meso_membw_probe(self._h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    def pair_floor(self, mode, reps=20):
        """Measured floor of the fp32 force kernel's mandatory work on the table in use (meso_pair_floor): mode 1 arithmetic, 2 loads,
        3 both.  Returns (us per launch, row entries walked, pairs evaluated)."""
        us = C.c_double()
        cnt = (C.c_longlong * 2)()
        self._ck(self.lib.meso_pair_floor(self._h, int(mode), int(reps), C.byref(us), cnt))
        return us.value, int(cnt[0]), int(cnt[1])

    def timer(self, name):
        ms, calls = C.c_double(), C.c_int64()
        self._ck(self.lib.meso_timer_get(self._h, name.encode(), C.byref(ms), C.byref(calls)))
        return ms.value, calls.value

    # -- known-answer kernels ---------------------------------------------------------------
    def tea(self, rounds, u, v):
        u = np.ascontiguousarray(u, np.uint32); v = np.ascontiguousarray(v, np.uint32)
        o0 = np.empty_like(u); o1 = np.empty_like(u)
        self._ck(self.lib.meso_test_tea(self._h, len(u), rounds, _p(u), _p(v), _p(o0), _p(o1)))
        return o0, o1

    def gaussian(self, u, v):
        u = np.ascontiguousarray(u, np.uint32); v = np.ascontiguousarray(v, np.uint32)
        dp = np.empty(len(u), np.float64); sp = np.empty(len(u), np.float32)
        self._ck(self.lib.meso_test_gaussian(self._h, len(u), _p(u), _p(v), _p(dp), _p(sp)))
        return dp, sp

    def write_restart(self, path):
        self._ck(self.lib.meso_write_restart(self._h, str(path).encode()))

    def read_restart(self, path):
        self._ck(self.lib.meso_read_restart(self._h, str(path).encode()))
        self._setup_done = False

    def profile_window(self, mode, start=0, end=0):
        modes = {"off": 0, "all": 1, "core": 2, "loop": 3, "interval": 4}
        self._ck(self.lib.meso_profile_window(self._h, modes[mode], start, end))

    def logistic(self, u, v):
        u = np.ascontiguousarray(u, np.uint32); v = np.ascontiguousarray(v, np.uint32)
        sp = np.empty(len(u), np.float32)
        self._ck(self.lib.meso_test_logistic(self._h, len(u), _p(u), _p(v), _p(sp)))
        return sp

    def script(self, path, var=None, value=None, log_bytes=1 << 16):
        buf = C.create_string_buffer(log_bytes)
        rc = self.lib.meso_script_run(self._h, path.encode(), None if var is None else var.encode(),
                                      None if value is None else str(value).encode(), buf, log_bytes)
        log = buf.value.decode()
        if rc:
            raise MesoError(self.lib.meso_last_error().decode() + "\n" + log)
        self._setup_done = True
        return log
