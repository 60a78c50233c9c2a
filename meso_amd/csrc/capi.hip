// extern "C" surface of libmeso_hip.so (include/meso_hip.h).  Plain pointers and sizes only.
#include "../../include/meso_hip.h"
#include "engine.h"
#include "meso_device.h"
#include <cstring>
#include <new>
#include <string>

using meso::Engine;

struct meso_ctx {
    Engine *eng;
};

static thread_local std::string g_err;

static int set_err(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define CTX(ctx)                                                           \
    if (!(ctx) || !(ctx)->eng) return set_err(MESO_ERR_ARG, "null context"); \
    Engine &E = *(ctx)->eng;                                                \
    if (int rc_ = E.resolve_counts()) { g_err = E.err; return rc_; }
#define RET(call)                                   \
    do {                                            \
        int _rc = (call);                           \
        if (_rc) g_err = E.err;                     \
        return _rc;                                 \
    } while (0)

extern "C" {

const char *meso_last_error(void) { return g_err.c_str(); }
int meso_version(void) { return 100; }

int meso_init(int device, meso_ctx **out)
{
    if (!out) return set_err(MESO_ERR_ARG, "null ctx pointer");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return set_err(MESO_ERR_HIP, "no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0) device = (-device) % ndev;   // LAMMPS -device -N: round robin (src/lammps.cpp:432-452)
    if (device >= ndev) return set_err(MESO_ERR_ARG, "device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return set_err(MESO_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    meso_ctx *c = new (std::nothrow) meso_ctx;
    if (!c) return set_err(MESO_ERR_HIP, "out of host memory");
    c->eng = new (std::nothrow) Engine(device);
    if (!c->eng) { delete c; return set_err(MESO_ERR_HIP, "out of host memory"); }
    *out = c;
    return MESO_OK;
}

int meso_finalize(meso_ctx *ctx)
{
    if (!ctx) return MESO_OK;
    delete ctx->eng;
    delete ctx;
    return MESO_OK;
}

int meso_device_sync(meso_ctx *ctx) { CTX(ctx); RET(E.sync()); }
int meso_set_option(meso_ctx *ctx, const char *key, double value)
{
    CTX(ctx);
    if (!key) return set_err(MESO_ERR_ARG, "null option key");
    RET(E.set_option(key, value));
}

int meso_set_box(meso_ctx *ctx, const double lo[3], const double hi[3], const int per[3])
{
    CTX(ctx);
    if (!lo || !hi) return set_err(MESO_ERR_ARG, "null box");
    RET(E.set_box(lo, hi, per));
}

int meso_comm_init(meso_ctx *ctx, int nranks, int rank, const int procgrid[3], int transport, const void *uid,
                   size_t uid_bytes)
{
    CTX(ctx);
    if (!procgrid) return set_err(MESO_ERR_ARG, "null procgrid");
    RET(E.comm_init(nranks, rank, procgrid, transport, uid, uid_bytes));
}

int meso_comm_get_unique_id(void *uid, size_t uid_bytes)
{
    return meso::comm_unique_id(uid, uid_bytes) ? set_err(MESO_ERR_COMM, "ncclGetUniqueId failed") : MESO_OK;
}

int meso_comm_count(meso_ctx *ctx, int *nranks_seen)
{
    CTX(ctx);
    if (!nranks_seen) return set_err(MESO_ERR_ARG, "null output");
    RET(E.comm_count(nranks_seen));
}

int meso_pair_kernel_name(meso_ctx *ctx, char *buf, int nbuf)
{
    if (!ctx || !buf || nbuf <= 0) return 1;
    CTX(ctx);
    snprintf(buf, (size_t)nbuf, "%s", E.pair_variant);
    return 0;
}

int meso_membw_probe(meso_ctx *ctx, size_t nbytes, int reps, double *copy_gbs)
{
    CTX(ctx);
    if (!copy_gbs || nbytes < 16 || reps < 1) return set_err(MESO_ERR_ARG, "bad probe arguments");
    RET(E.membw_probe(nbytes, reps, copy_gbs));
}

int meso_decomp_procgrid(int nranks, const double prd[3], int procgrid[3])
{
    if (nranks < 1 || !prd || !procgrid) return set_err(MESO_ERR_ARG, "invalid procgrid arguments");
    meso::decomp_procgrid(nranks, prd, procgrid);
    return MESO_OK;
}

int meso_decomp_plan(const double boxlo[3], const double boxhi[3], const int periodic[3], const int procgrid[3], int rank,
                     double cutghost, double sublo[3], double subhi[3], double slab_lo[3], double slab_hi[3], int peer27[27],
                     int active27[27], double shift27[81], double center27[81])
{
    if (!boxlo || !boxhi || !periodic || !procgrid || !sublo || !subhi || !slab_lo || !slab_hi || !peer27 || !active27 || !shift27 || !center27)
        return set_err(MESO_ERR_ARG, "invalid decomposition-plan arguments");
    const int n = procgrid[0] * procgrid[1] * procgrid[2];
    if (procgrid[0] < 1 || procgrid[1] < 1 || procgrid[2] < 1 || rank < 0 || rank >= n) return set_err(MESO_ERR_ARG, "rank outside the processor grid");
    // rank -> grid location, x fastest (what meso_comm_init assumes)
    const int myloc[3] = {rank % procgrid[0], (rank / procgrid[0]) % procgrid[1], rank / (procgrid[0] * procgrid[1])};
    if (meso::decomp_plan(boxlo, boxhi, periodic, procgrid, myloc, cutghost, sublo, subhi, slab_lo, slab_hi, peer27, active27, shift27, center27))
        return set_err(MESO_ERR_ARG, "Sub-domain smaller than the ghost cutoff is not supported");
    return MESO_OK;
}

int meso_comm_set_host_exchange(meso_ctx *ctx, meso_host_exchange_fn fn, void *user)
{
    CTX(ctx);
    E.host_exchange = fn;
    E.host_exchange_user = user;
    return MESO_OK;
}

int meso_set_mass(meso_ctx *ctx, int ntypes, const double *mass)
{
    CTX(ctx);
    if (!mass) return set_err(MESO_ERR_ARG, "null mass array");
    RET(E.set_mass(ntypes, mass));
}

int meso_atoms_upload(meso_ctx *ctx, int n, const double *x, const double *v, const int *tag, const int *type,
                      const int *mask, const int *image)
{
    CTX(ctx);
    RET(E.atoms_upload(n, x, v, tag, type, mask, image));
}

int meso_atoms_count(meso_ctx *ctx, int *nlocal, int *nghost, int *n_bulk)
{
    CTX(ctx);
    if (nlocal) *nlocal = E.nlocal;
    if (nghost) *nghost = E.nghost;
    if (n_bulk) *n_bulk = E.n_bulk;
    return MESO_OK;
}

int meso_atoms_download(meso_ctx *ctx, double *x, double *v, double *f, int *tag, int *type, int *image)
{
    CTX(ctx);
    RET(E.atoms_download(x, v, f, tag, type, image));
}

int meso_neighbor(meso_ctx *ctx, double skin, int every, int delay, int check) { CTX(ctx); RET(E.neighbor(skin, every, delay, check)); }
int meso_pair_dpd_settings(meso_ctx *ctx, int style, double cut, int seed) { CTX(ctx); RET(E.pair_settings(style, cut, seed)); }
int meso_pair_dpd_tableforce_coeff(meso_ctx *ctx, int i, int j, double gamma, double sigma, int len, const double *t) { CTX(ctx); RET(E.pair_coeff_table(i, j, gamma, sigma, len, t)); }
int meso_pair_dpd_polyforce_coeff(meso_ctx *ctx, int i, int j, double gamma, double sigma, int order, const double *c) { CTX(ctx); RET(E.pair_coeff_poly(i, j, gamma, sigma, order, c)); }
int meso_pair_dpd_coeff(meso_ctx *ctx, int i, int j, double a0, double gamma, double sigma, double expw, double cut)
{
    CTX(ctx);
    RET(E.pair_coeff(i, j, a0, gamma, sigma, expw, cut));
}

int meso_special_bonds(meso_ctx *ctx, double w12, double w13, double w14) { CTX(ctx); RET(E.special_bonds(w12, w13, w14)); }
int meso_bonds_upload(meso_ctx *ctx, int nb, const int *ti, const int *tj, const int *bt) { CTX(ctx); RET(E.bonds_upload(nb, ti, tj, bt)); }
int meso_bond_style_harmonic(meso_ctx *ctx, int nbt) { CTX(ctx); RET(E.bond_style(nbt)); }
int meso_bond_coeff(meso_ctx *ctx, int type, double k, double r0) { CTX(ctx); RET(E.bond_coeff(type, k, r0)); }
int meso_bond_style_fene(meso_ctx *ctx, int nbt) { CTX(ctx); RET(E.bond_style(nbt, 1)); }
int meso_bond_coeff_fene(meso_ctx *ctx, int type, double k, double r0, double eps, double sigma) { CTX(ctx); RET(E.bond_coeff(type, k, r0, eps, sigma)); }
int meso_bond_compute(meso_ctx *ctx, int eflag) { CTX(ctx); RET(E.bond_compute(eflag)); }
int meso_angles_upload(meso_ctx *ctx, int na, const int *t1, const int *t2, const int *t3, const int *ty) { CTX(ctx); RET(E.angles_upload(na, t1, t2, t3, ty)); }
int meso_angle_style_harmonic(meso_ctx *ctx, int nat) { CTX(ctx); RET(E.angle_style(nat)); }
int meso_angle_coeff(meso_ctx *ctx, int type, double k, double theta0) { CTX(ctx); RET(E.angle_coeff(type, k, theta0)); }
int meso_angle_compute(meso_ctx *ctx, int eflag) { CTX(ctx); RET(E.angle_compute(eflag)); }
int meso_compute_eangle(meso_ctx *ctx, double *e) { CTX(ctx); if (!e) return set_err(MESO_ERR_ARG, "null output"); RET(E.compute_eangle(e)); }
int meso_compute_ebond(meso_ctx *ctx, double *e) { CTX(ctx); if (!e) return set_err(MESO_ERR_ARG, "null output"); RET(E.compute_ebond(e)); }

int meso_timestep(meso_ctx *ctx, double dt)
{
    CTX(ctx);
    if (!(dt > 0.0)) return set_err(MESO_ERR_ARG, "Illegal timestep command");
    E.dt = dt;
    return MESO_OK;
}

int meso_setup(meso_ctx *ctx) { CTX(ctx); RET(E.setup()); }
int meso_run(meso_ctx *ctx, int nsteps)
{
    CTX(ctx);
    if (nsteps < 0) return set_err(MESO_ERR_ARG, "Illegal run command");
    RET(E.run(nsteps));
}
int meso_nve_initial(meso_ctx *ctx) { CTX(ctx); RET(E.nve_initial()); }
int meso_nve_final(meso_ctx *ctx) { CTX(ctx); RET(E.nve_final()); }
int meso_neighbor_decide(meso_ctx *ctx, int *rebuild)
{
    CTX(ctx);
    if (!rebuild) return set_err(MESO_ERR_ARG, "null rebuild flag");
    RET(E.decide(rebuild));
}
int meso_reneighbor(meso_ctx *ctx) { CTX(ctx); RET(E.reneighbor()); }
int meso_halo_forward(meso_ctx *ctx) { CTX(ctx); RET(E.halo_forward()); }
int meso_force_clear(meso_ctx *ctx, int range) { CTX(ctx); RET(E.force_clear(range)); }
int meso_pair_compute(meso_ctx *ctx, int range, int eflag, int vflag)
{
    CTX(ctx);
    if (range < 0 || range > 2) return set_err(MESO_ERR_ARG, "invalid work range");
    RET(E.pair_compute(range, eflag, vflag));
}
int meso_step_advance(meso_ctx *ctx, int64_t ntimestep) { CTX(ctx); E.ntimestep = ntimestep; return MESO_OK; }

int meso_compute_temp(meso_ctx *ctx, double *t) { CTX(ctx); if (!t) return set_err(MESO_ERR_ARG, "null output"); RET(E.compute_temp(t)); }
int meso_compute_pe(meso_ctx *ctx, double *pe) { CTX(ctx); if (!pe) return set_err(MESO_ERR_ARG, "null output"); RET(E.compute_pe(pe)); }
int meso_compute_pressure(meso_ctx *ctx, double *p) { CTX(ctx); if (!p) return set_err(MESO_ERR_ARG, "null output"); RET(E.compute_pressure(p)); }

int meso_neigh_info(meso_ctx *ctx, int *n_col, int *max_count, double *avg, int64_t *nbuild)
{
    CTX(ctx);
    RET(E.neigh_info(n_col, max_count, avg, nbuild));
}

int meso_tally_ev(meso_ctx *ctx) { CTX(ctx); RET(E.tally_ev()); }

int meso_xchg_stats(meso_ctx *ctx, char *buf, int nbuf)
{
    if (!ctx || !ctx->eng || !buf || nbuf <= 0) return set_err(MESO_ERR_ARG, "null argument");
    snprintf(buf, (size_t)nbuf, "%s", ctx->eng->xchg_report().c_str());
    return 0;
}
int meso_pair_floor(meso_ctx *ctx, int mode, int reps, double *us, long long *counts)
{
    CTX(ctx);
    long c[2] = {0, 0};
    const int rc = E.pair_floor(mode, reps, us, c);
    if (counts) { counts[0] = c[0]; counts[1] = c[1]; }
    RET(rc);
}
int meso_neigh_parts(meso_ctx *ctx, int *parted, int *group, int *nfront, int *nback, int *front, int *back, int stride)
{
    CTX(ctx);
    if (!parted || !group) return set_err(MESO_ERR_ARG, "null argument");
    if ((front || back) && stride <= 0) return set_err(MESO_ERR_ARG, "invalid neighbour buffers");
    RET(E.neigh_parts(parted, group, nfront, nback, front, back, stride));
}
int meso_neigh_download(meso_ctx *ctx, int *count, int *table, int stride)
{
    CTX(ctx);
    if (!count || !table || stride <= 0) return set_err(MESO_ERR_ARG, "invalid neighbour buffers");
    RET(E.neigh_download(count, table, stride));
}
int meso_merged_download(meso_ctx *ctx, float *c4, float *v4, int nall) { CTX(ctx); RET(E.merged_download(c4, v4, nall)); }
int meso_timer_reset(meso_ctx *ctx) { CTX(ctx); RET(E.timer_reset()); }
int meso_timer_get(meso_ctx *ctx, const char *name, double *ms, int64_t *calls)
{
    CTX(ctx);
    if (!name) return set_err(MESO_ERR_ARG, "null timer name");
    RET(E.timer_get(name, ms, calls));
}
int64_t meso_ntimestep(meso_ctx *ctx) { return (ctx && ctx->eng) ? ctx->eng->ntimestep : -1; }

int meso_test_tea(meso_ctx *ctx, int n, int rounds, const uint32_t *u, const uint32_t *v, uint32_t *o0, uint32_t *o1)
{
    CTX(ctx);
    if (n < 0 || !u || !v || !o0 || !o1) return set_err(MESO_ERR_ARG, "invalid test buffers");
    RET(E.test_tea(n, rounds, u, v, o0, o1));
}
int meso_test_gaussian(meso_ctx *ctx, int n, const uint32_t *u, const uint32_t *v, double *odp, float *osp)
{
    CTX(ctx);
    if (n < 0 || !u || !v || !odp || !osp) return set_err(MESO_ERR_ARG, "invalid test buffers");
    RET(E.test_gaussian(n, u, v, odp, osp));
}
int meso_write_restart(meso_ctx *ctx, const char *path) { CTX(ctx); if (!path) return set_err(MESO_ERR_ARG, "null path"); RET(E.write_restart(path)); }
int meso_read_restart(meso_ctx *ctx, const char *path) { CTX(ctx); if (!path) return set_err(MESO_ERR_ARG, "null path"); RET(E.read_restart(path)); }
int meso_profile_window(meso_ctx *ctx, int mode, int64_t start, int64_t end) { CTX(ctx); RET(E.profile_window(mode, start, end)); }
int meso_test_logistic(meso_ctx *ctx, int n, const uint32_t *u, const uint32_t *v, float *out)
{
    CTX(ctx);
    if (n < 0 || !u || !v || !out) return set_err(MESO_ERR_ARG, "invalid test buffers");
    RET(E.test_logistic(n, u, v, out));
}
uint32_t meso_seed_now(int seed, int64_t ntimestep) { return meso::premix_tea<64>((uint32_t)seed, (uint32_t)ntimestep); }

int meso_script_run(meso_ctx *ctx, const char *path, const char *var_name, const char *var_value, char *log,
                    size_t log_bytes)
{
    CTX(ctx);
    if (!path) return set_err(MESO_ERR_ARG, "null script path");
    std::string out;
    int rc = meso::script_run(E, path, var_name, var_value, out);
    if (log && log_bytes) {
        size_t n = out.size() < log_bytes - 1 ? out.size() : log_bytes - 1;
        memcpy(log, out.data(), n);
        log[n] = 0;
    }
    if (rc) g_err = E.err;
    return rc;
}

} // extern "C"
