// Engine: device-resident DPD timestep for MI355X.  The per-step schedule follows
// ModifiedVerlet::run (/root/reference/src/USER-MESO/mvv_meso.cu:243-425) with the host round trips removed:
// particles never leave HBM between steps, ghosts are produced by device pack kernels (self transport)
// or travel as packed float4 pairs over RCCL, and the reorder / cell list use rocPRIM sorts.
#include "engine.h"
#include "meso_device.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <thread>

namespace meso {

#define HIPCHK(call)                                                      \
    do {                                                                  \
        int _rc = check((call), #call);                                   \
        if (_rc) return _rc;                                              \
    } while (0)
#define TRY(call)                                                         \
    do {                                                                  \
        int _rc = (call);                                                 \
        if (_rc) return _rc;                                              \
    } while (0)


int Engine::fail(int code, const std::string &msg)
{
    err = msg;
    return code;
}

int Engine::check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    return 2;
}

Engine::Engine(int dev) : device(dev)
{
    for (int d = 0; d < 3; d++) { boxlo[d] = 0; boxhi[d] = 1; prd[d] = 1; periodic[d] = 1; }
    // development switches: option defaults from the environment (whole test-suite runs under an alternative kernel path)
    if (const char *e = getenv("MESO_PAIR_DEBUG")) pair_debug = atoi(e);
    if (const char *e = getenv("MESO_ASYNC_COUNTS")) async_counts = atoi(e);
    if (const char *e = getenv("MESO_OVERLAP_REBUILD")) overlap_rebuild = atoi(e);
    if (const char *e = getenv("MESO_GHOST_EPILOGUE")) ghost_epilogue = atoi(e);
    if (const char *e = getenv("MESO_PAIR_NPART")) pair_npart = atoi(e);
}

Engine::~Engine() { free_all(); }

template <typename T>
static hipError_t dalloc(T *&p, size_t n)
{
    p = nullptr;
    return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T));
}

template <typename T>
static void dfree(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

static void free_soa(AtomSoA &a)
{
    for (int d = 0; d < 3; d++) { dfree(a.x[d]); dfree(a.v[d]); dfree(a.f[d]); }
    dfree(a.tag); dfree(a.type); dfree(a.mask); dfree(a.image); dfree(a.mass);
    dfree(a.nbond); dfree(a.bond_tag); dfree(a.bond_type); dfree(a.nspecial); dfree(a.special);
    dfree(a.nangle); dfree(a.angle_tag);
}

void Engine::free_all()
{
    if (stream) (void)hipStreamSynchronize(stream);
    free_soa(cur); free_soa(alt);
    dfree(coord4); dfree(veloc4); dfree(coord4_next); dfree(veloc4_next);
    for (int k = 0; k < 6; k++) dfree(virial[k]);
    dfree(d_bond_kr0); dfree(e_bond); dfree(bond_idx); dfree(tagmap); dfree(tagc); dfree(tagbits);
    dfree(d_angle_cf); dfree(e_angle); dfree(angle_idx);
    dfree(e_pair); dfree(xhold); dfree(d_mass_type); dfree(d_dtfm_type); dfree(d_coeff64); dfree(d_coeff32); dfree(d_poly); dfree(d_ftab);
    dfree(pair_count); dfree(pair_nback); dfree(pair_table); dfree(pair_back);
    dfree(bin_key); dfree(bin_key_alt); dfree(bin_val); dfree(bin_val_alt); dfree(img_cnt); dfree(img); dfree(d_shift27);
    dfree(rkey); dfree(rkey_alt); dfree(rval); dfree(rval_alt);
    dfree(fr_bucket); dfree(fr_ovf); dfree(fr_novf); dfree(fr_scratch); dfree(senddir); dfree(fr_gorder); dfree(fr_gcnt);
    for (int k = 0; k < 2; k++) { dfree(fr_ttot[k]); dfree(fr_stot[k]); dfree(fr_gttot[k]); dfree(fr_gstot[k]); }
    if (sort_temp) (void)hipFree(sort_temp);
    sort_temp = nullptr;
    dfree(estart); dfree(gstart); dfree(gcount); dfree(rcount); dfree(gslot); dfree(mr_gcnt); dfree(mig_cnt); dfree(mig_lst); dfree(d_vofs); dfree(d_center27);
    dfree(binrange);
    dfree(brick_hoff); dfree(brick_hmap); dfree(brick_hdr); dfree(brick_own); dfree(brick_order2); dfree(tile_queue);
    if (debug_event) { (void)hipEventDestroy(debug_event); debug_event = nullptr; }
    if (debug_stream) { (void)hipStreamDestroy(debug_stream); debug_stream = nullptr; }
    dfree(sendlist); dfree(chunk_count); dfree(chunk_offset); dfree(d_dir_start);
    dfree(d_partial); dfree(d_scalar); dfree(d_flags); dfree(sendlist_aux); dfree(d_mr);
    if (stage_send) (void)hipFree(stage_send);
    if (stage_recv) (void)hipFree(stage_recv);
    stage_send = stage_recv = nullptr; stage_send_bytes = stage_recv_bytes = 0;
    if (stage2_send) (void)hipFree(stage2_send);
    if (stage2_recv) (void)hipFree(stage2_recv);
    stage2_send = stage2_recv = nullptr; stage2_send_bytes = stage2_recv_bytes = 0;
    free_fwd_tab();
    comm_free();
    if (ev_pack) (void)hipEventDestroy(ev_pack);
    if (ev_halo) (void)hipEventDestroy(ev_halo);
    ev_pack = ev_halo = nullptr;
    if (h_flags) (void)hipHostFree(h_flags);
    if (h_scalar) (void)hipHostFree(h_scalar);
    h_flags = nullptr; h_scalar = nullptr;
    for (auto &kv : timers)
        for (auto &pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    timers.clear();
    for (auto e : event_pool) (void)hipEventDestroy(e);
    event_pool.clear();
    if (stream) (void)hipStreamDestroy(stream);
    if (side) (void)hipStreamDestroy(side);
    stream = side = nullptr;
    nmax = 0;
}

// ------------------------------------------------------------------------------------------------
// timers (HIP events on the engine stream; only when profiling is on)
// ------------------------------------------------------------------------------------------------
void Engine::tbegin(const char *name)
{
    if (!profiling) return;
    hipEvent_t a, b;
    if (event_pool.size() >= 2) {
        a = event_pool.back(); event_pool.pop_back();
        b = event_pool.back(); event_pool.pop_back();
    } else {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
    }
    (void)hipEventRecord(a, stream);
    timers[name].pending.push_back(std::make_pair(a, b));
}

void Engine::tend(const char *name)
{
    if (!profiling) return;
    PhaseTimer &t = timers[name];
    if (t.pending.empty()) return;
    (void)hipEventRecord(t.pending.back().second, stream);
}

void Engine::tflush()
{
    if (!profiling) return;
    (void)hipStreamSynchronize(stream);
    for (auto &kv : timers) {
        for (auto &pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.ms += ms;
                kv.second.calls++;
            }
            event_pool.push_back(pr.first);
            event_pool.push_back(pr.second);
        }
        kv.second.pending.clear();
    }
}

int Engine::timer_reset()
{
    xchg_events_flush();
    xchg_stats.clear();
    tflush();
    for (auto &kv : timers) { kv.second.ms = 0.0; kv.second.calls = 0; }
    return 0;
}

int Engine::timer_get(const std::string &name, double *ms, int64_t *calls)
{
    // (a counter, not a timer: rebuilds that outgrew a capacity and were redone through the synchronous path, engine.hip prepare_redo)
    if (name == "rebuilds_redone") { if (ms) *ms = 0.0; if (calls) *calls = nredo; return 0; }
    tflush();
    auto it = timers.find(name);
    if (ms) *ms = it == timers.end() ? 0.0 : it->second.ms;
    if (calls) *calls = it == timers.end() ? 0 : it->second.calls;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// configuration
// ------------------------------------------------------------------------------------------------
int Engine::set_box(const double *lo, const double *hi, const int *per)
{
    for (int d = 0; d < 3; d++) {
        if (!(hi[d] > lo[d])) return fail(1, "Box bounds are invalid");
        boxlo[d] = lo[d]; boxhi[d] = hi[d]; prd[d] = hi[d] - lo[d]; periodic[d] = per ? per[d] : 1;
    }
    have_box = true;
    params_ready = false;
    return 0;
}

int Engine::set_mass(int nt, const double *m)
{
    if (nt < 1) return fail(1, "Invalid number of atom types");
    if (ntypes && nt != ntypes) return fail(1, "Number of atom types changed");
    ntypes = nt;
    mass_type.assign(m, m + nt + 1);
    for (int t = 1; t <= nt; t++)
        if (!(mass_type[t] > 0.0)) return fail(1, "Invalid mass value");
    if ((int)coeff.size() != nt * nt * N_COEFF) { coeff.assign((size_t)nt * nt * N_COEFF, 0.0); coeff_set.assign((size_t)nt * nt, 0); }
    params_ready = false;
    return 0;
}

int Engine::neighbor(double s, int ev, int dl, int chk)
{
    if (s < 0.0) return fail(1, "Illegal neighbor command");
    if (ev <= 0 || dl < 0) return fail(1, "Illegal neigh_modify command");
    skin = s; every = ev; delay = dl; dist_check = chk ? 1 : 0;
    params_ready = false;
    return 0;
}

// MesoPairDPD::settings (pair_dpd_meso.cu:272-288): pair_style dpd/meso rc seed
int Engine::pair_settings(int style, double cut, int sd)
{
    if (style < 0 || style > 4) return fail(1, "Illegal pair_style command");
    if (!(cut > 0.0)) return fail(1, "Illegal pair_style command");
    // MesoPairDPDMini (pair_dpd_minimal_meso.cu:239-265): one global coefficient set, cutoff fixed at 1
    if (style == 2 && cut != 1.0) return fail(1, "pair_style dpd/mini/meso has a fixed cutoff of 1");
    pair_rng = style == 2 ? 1 : style == 4 ? 2 : 0;
    pair_ftab = style == 4;          // MesoPairDPDTableForce: fp32 arithmetic, uniform TEA noise, tabulated conservative force
    ftab.clear(); ftab_len = 0;
    pair_poly = style == 3;          // MesoPairDPDPolyForce: fp32 arithmetic, TEA noise, polynomial conservative force
    poly.clear();
    pair_style = style >= 2 ? 1 : style; cut_global = cut; seed = sd;
    have_pair = true;
    params_ready = false;
    return 0;
}

// MesoPairDPD::coeff (pair_dpd_meso.cu:290-327): pair_coeff i j a0 gamma sigma s [rc]; init_one mirrors ij -> ji
int Engine::pair_coeff(int i, int j, double a0, double gamma, double sigma, double expw, double cut)
{
    if (!have_pair) return fail(3, "pair_coeff before pair_style");
    if (ntypes == 0) return fail(3, "pair_coeff before atom types are known");
    if (i < 1 || j < 1 || i > ntypes || j > ntypes) return fail(1, "Incorrect args for pair coefficients");
    if (pair_poly) return fail(1, "Incorrect args for pair coefficients: dpd/polyforce/meso takes gamma sigma order c_order ... c_0");
    if (pair_ftab) return fail(1, "Incorrect args for pair dpd/tableforce/meso: type1 type2 gamma sigma < fc_file_name | fc_table >");
    if (cut <= 0.0) cut = cut_global;
    if (pair_rng) {
        // MesoPairDPDMini::coeff: a0, gamma, sigma are scalars of the style - every pair of types gets them, s = 1, rc = 1
        for (int t = 0; t < ntypes * ntypes; t++) {
            double *c = &coeff[(size_t)t * N_COEFF];
            c[P_CUT] = 1.0; c[P_CUTSQ] = 1.0; c[P_CUTINV] = 1.0; c[P_EXPW] = 1.0;
            c[P_A0] = a0; c[P_GAMMA] = gamma; c[P_SIGMA] = sigma;
            coeff_set[t] = 1;
        }
        have_coeff = true;
        params_ready = false;
        return 0;
    }
    for (int k = 0; k < 2; k++) {
        int a = k ? j - 1 : i - 1, b = k ? i - 1 : j - 1;
        double *c = &coeff[((size_t)a * ntypes + b) * N_COEFF];
        c[P_CUT] = cut; c[P_CUTSQ] = cut * cut; c[P_CUTINV] = 1.0 / cut; c[P_EXPW] = expw;
        c[P_A0] = a0; c[P_GAMMA] = gamma; c[P_SIGMA] = sigma;
        coeff_set[(size_t)a * ntypes + b] = 1;
    }
    have_coeff = true;
    params_ready = false;
    return 0;
}

// MesoPairDPDPolyForce::coeff (pair_dpd_polyforce_meso.cu:290-335): pair_coeff i j gamma sigma order c_order ... c_0;
// cutoff = the style's global cutoff, weight exponent 1
int Engine::pair_coeff_poly(int i, int j, double gamma, double sigma, int order, const double *c)
{
    if (!have_pair || !pair_poly) return fail(3, "polynomial coefficients need pair_style dpd/polyforce/meso");
    if (ntypes == 0) return fail(3, "pair_coeff before atom types are known");
    if (i < 1 || j < 1 || i > ntypes || j > ntypes || !c) return fail(1, "Incorrect args for pair coefficients");
    if (order < 0 || order + 1 > MESO_POLY_MAXLEN) return fail(1, "Incorrect args for pair coefficients: polynomial order");
    if (poly.empty()) poly.assign((size_t)ntypes * ntypes * MESO_POLY_PITCH, 0.f);
    for (int k = 0; k < 2; k++) {
        int a = k ? j - 1 : i - 1, b = k ? i - 1 : j - 1;
        double *q = &coeff[((size_t)a * ntypes + b) * N_COEFF];
        q[P_CUT] = cut_global; q[P_CUTSQ] = cut_global * cut_global; q[P_CUTINV] = 1.0 / cut_global; q[P_EXPW] = 1.0;
        q[P_A0] = 0.0; q[P_GAMMA] = gamma; q[P_SIGMA] = sigma;
        coeff_set[(size_t)a * ntypes + b] = 1;
        float *row = &poly[((size_t)a * ntypes + b) * MESO_POLY_PITCH];
        row[0] = (float)order;
        for (int t = 0; t <= order; t++) row[1 + t] = (float)c[t];
    }
    have_coeff = true;
    params_ready = false;
    return 0;
}

// MesoPairDPDTableForce::coeff (pair_dpd_tableforce_meso.cu:306-356): pair_coeff i j gamma sigma < file | L values >; the
// table length L is a setting of the style (:290) and therefore the same for every pair of types
int Engine::pair_coeff_table(int i, int j, double gamma, double sigma, int len, const double *t)
{
    if (!have_pair || !pair_ftab) return fail(3, "force tables need pair_style dpd/tableforce/meso");
    if (ntypes == 0) return fail(3, "pair_coeff before atom types are known");
    if (i < 1 || j < 1 || i > ntypes || j > ntypes || !t || len < 2) return fail(1, "Incorrect args for pair coefficients");
    if (ftab_len && len != ftab_len) return fail(1, "Incorrect args for pair dpd/tableforce/meso: every table has table_length entries");
    ftab_len = len;
    if (ftab.empty()) ftab.assign((size_t)ntypes * ntypes * len, 0.f);
    for (int k = 0; k < 2; k++) {
        int a = k ? j - 1 : i - 1, b = k ? i - 1 : j - 1;
        double *q = &coeff[((size_t)a * ntypes + b) * N_COEFF];
        q[P_CUT] = cut_global; q[P_CUTSQ] = cut_global * cut_global; q[P_CUTINV] = 1.0 / cut_global; q[P_EXPW] = 1.0;
        q[P_A0] = 0.0; q[P_GAMMA] = gamma; q[P_SIGMA] = sigma;
        coeff_set[(size_t)a * ntypes + b] = 1;
        for (int e = 0; e < len; e++) ftab[((size_t)a * ntypes + b) * len + e] = (float)t[e];
    }
    have_coeff = true;
    params_ready = false;
    return 0;
}

int Engine::set_option(const std::string &key, double val)
{
    if (key == "fused_rebuild") { fused_rebuild = (int)val; return 0; }
    if (key == "row_part") { row_part = (int)val; return 0; }
    if (key == "check_launches") { check_launches = (int)val; return 0; }
    if (key == "xcd_balance") { xcd_balance = (int)val; return 0; }
    if (key == "fuse_count") { fuse_count = (int)val; return 0; }
    if (key == "lean_boundary") { lean_boundary = (int)val; return 0; }
    if (key == "merge_ghosts") { merge_ghosts = (int)val; return 0; }
    if (key == "split_gather") { split_gather = (int)val; return 0; }
    if (key == "brick2") { brick2 = (int)val; return 0; }
    if (key == "brick2_limit") { brick2_limit = (int)val; return 0; }
    if (key == "tile_persist") { tile_persist = (int)val; return 0; }
    if (key == "report_poll") { report_poll = (int)val; return 0; }
    if (key == "debug_early_reuse") { debug_early_reuse = (int)val; return 0; }
    if (key == "debug_ghost_cap") { debug_ghost_cap = (int)val; return 0; }      // tests: the next asynchronous rebuild reserves this many ghosts only
    if (key == "fused_cap") { fr_cap_user = (int)val; return 0; }       // tests: atoms per cell bucket (the rest takes the overflow list)
    if (key == "profile") { tflush(); profiling = val != 0.0; return 0; }
    if (key == "neigh_kernel") { neigh_kernel = (int)val; return 0; }
    if (key == "fuse_clear") { fuse_clear = (int)val; return 0; }
    if (key == "fuse_step") { fuse_step = (int)val; return 0; }
    if (key == "overlap") { overlap = (int)val; return 0; }
    if (key == "pair_kernel") {
        // 2 (= 5): ring kernel; 0: lane per atom (the reference-like kernel that also serves the energy/virial steps).
        // The tile, MLP and brick kernels of round 1 were retired.
        if (val != 0 && val != 2 && val != 5) return fail(1, "pair_kernel: 0 (lane per atom) or 2 (ring); the other kernels were retired");
        pair_kernel = (int)val;
        return 0;
    }
    if (key == "fuse_pair") { fuse_pair = (int)val; return 0; }
    if (key == "brick_margin") { if (val < 1.0) return fail(1, "brick_margin must be >= 1"); brick_margin = val; params_ready = false; return 0; }
    if (key == "pair_share") { pair_share = (int)val; return 0; }
    if (key == "ghost_sort") { ghost_sort = (int)val; return 0; }
    if (key == "reorder_sort") { reorder_sort = (int)val; return 0; }
    if (key == "tile_plan") { tile_plan = (int)val; return 0; }
    if (key == "refresh_epilogue") { refresh_epilogue = (int)val; return 0; }
    if (key == "refresh_direct") { refresh_direct = (int)val; return 0; }
    if (key == "border_fused") { border_fused = (int)val; return 0; }
    if (key == "mig_slim") { mig_slim = (int)val; return 0; }
    if (key == "border_runs") { border_runs = (int)val; return 0; }      // several ranks: ghosts in message order, cells as runs (0: unpack + binning chain)
    if (key == "reorder_cap") { reorder_cap_user = (int)val; return 0; }      // tests: force the ordering pass off its LDS stage
    if (key == "pair_npart") { pair_npart = (int)val; return 0; }
    if (key == "async_counts") { async_counts = (int)val; return 0; }
    if (key == "fuse_bonds") { fuse_bonds = (int)val; return 0; }
    if (key == "mr_cap_margin") { mr_cap_margin = val; return 0; }
    if (key == "mig_cap_floor") { mig_cap_floor = (int)val; return 0; }
    if (key == "overlap_rebuild") { overlap_rebuild = (int)val; return 0; }
    if (key == "ghost_epilogue") { ghost_epilogue = (int)val; return 0; }
    if (key == "async_grid_scale") { async_grid_scale = val; return 0; }      // tests: under-sized grids must still cover every ghost
    if (key == "pair_debug") { pair_debug = (int)val; return 0; }
    if (key == "layout") {     // kept for scripts of round 1: only the cell-ordered layout exists
        if (val != 2) return fail(1, "layout: only 2 (cell order = storage order); layouts 0 and 1 were retired");
        return 0;
    }
    if (key == "groupbit") { groupbit = (int)val; return 0; }
    return fail(1, "Unknown option '" + key + "'");
}

// ------------------------------------------------------------------------------------------------
// memory
// ------------------------------------------------------------------------------------------------
template <typename T>
static hipError_t regrow(T *&p, size_t keep, size_t cap, hipStream_t s)
{
    T *q = nullptr;
    hipError_t e = hipMalloc((void **)&q, (cap ? cap : 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (p && keep) {
        e = hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { (void)hipFree(q); return e; }
    }
    if (p) (void)hipFree(p);
    p = q;
    return hipSuccess;
}

int Engine::alloc_atoms(int cap)
{
    size_t keep = (size_t)nlocal, c = (size_t)cap;
    if (!stream) {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIPCHK(dalloc(d_partial, 1024));
        HIPCHK(dalloc(d_scalar, 16));
        HIPCHK(dalloc(d_flags, 16));
        HIPCHK(dalloc(d_dir_start, 32));
        HIPCHK(dalloc(sendlist_aux, 2 * 27 * 27 + 64));
        HIPCHK(hipHostMalloc((void **)&h_flags, 192 * sizeof(int)));      // [64..127]: report of the multi-rank border exchange, [128..159]: of the migration
        HIPCHK(hipHostMalloc((void **)&h_scalar, 16 * sizeof(double)));
        HIPCHK(hipMemsetAsync(d_flags, 0, 16 * sizeof(int), stream));
    }
    for (int d = 0; d < 3; d++) {
        HIPCHK(regrow(cur.x[d], keep, c, stream));
        HIPCHK(regrow(cur.v[d], keep, c, stream));
        HIPCHK(regrow(cur.f[d], keep, c, stream));
        HIPCHK(regrow(alt.x[d], 0, c, stream));
        HIPCHK(regrow(alt.v[d], 0, c, stream));
        HIPCHK(regrow(alt.f[d], 0, c, stream));
    }
    HIPCHK(regrow(cur.tag, keep, c, stream)); HIPCHK(regrow(alt.tag, 0, c, stream));
    HIPCHK(regrow(cur.type, keep, c, stream)); HIPCHK(regrow(alt.type, 0, c, stream));
    HIPCHK(regrow(cur.mask, keep, c, stream)); HIPCHK(regrow(alt.mask, 0, c, stream));
    HIPCHK(regrow(cur.image, keep, c, stream)); HIPCHK(regrow(alt.image, 0, c, stream));
    HIPCHK(regrow(cur.mass, keep, c, stream)); HIPCHK(regrow(alt.mass, 0, c, stream));
    if (bpa > 0 || msp > 0) {
        TRY(alloc_topology(cur, cap, nlocal));
        TRY(alloc_topology(alt, cap, 0));
        HIPCHK(regrow(bond_idx, 0, c * (size_t)std::max(bpa, 1), stream));
        HIPCHK(regrow(tagc, 0, c, stream));
        HIPCHK(regrow(e_bond, 0, c, stream));
        if (apa > 0) {
            HIPCHK(regrow(angle_idx, 0, c * 3 * (size_t)apa, stream));
            HIPCHK(regrow(e_angle, 0, c, stream));
        }
    }
    HIPCHK(regrow(coord4, 0, c, stream)); HIPCHK(regrow(veloc4, 0, c, stream));
    merged_in_reorder = false;
    HIPCHK(regrow(coord4_next, 0, c, stream)); HIPCHK(regrow(veloc4_next, 0, c, stream));

    for (int k = 0; k < 6; k++) HIPCHK(regrow(virial[k], 0, c, stream));
    HIPCHK(regrow(e_pair, 0, c, stream));
    HIPCHK(regrow(xhold, 0, 3 * c, stream));
    HIPCHK(regrow(pair_count, 0, c, stream)); HIPCHK(regrow(pair_nback, 0, c, stream));
    HIPCHK(regrow(bin_key, 0, c, stream)); HIPCHK(regrow(bin_key_alt, 0, c, stream));
    HIPCHK(regrow(bin_val, 0, c, stream)); HIPCHK(regrow(bin_val_alt, 0, c, stream));
    HIPCHK(regrow(rkey, 0, c, stream)); HIPCHK(regrow(rkey_alt, 0, c, stream));
    HIPCHK(regrow(rval, 0, c, stream)); HIPCHK(regrow(rval_alt, 0, c, stream));
    HIPCHK(regrow(sendlist, 0, c, stream));
    HIPCHK(regrow(gslot, 0, c, stream));
    HIPCHK(regrow(gtmp_placed, 0, c, stream)); HIPCHK(regrow(gtmp_code, 0, c, stream)); HIPCHK(regrow(perm_inverse, 0, c, stream));
    HIPCHK(regrow(img_cnt, 0, c, stream)); HIPCHK(regrow(img, 0, c * 8, stream));
    HIPCHK(regrow(senddir, 0, c, stream)); HIPCHK(regrow(fr_scratch, 0, c, stream));
    images_ready = false;
    mr_images_ready = false;
    img_alloc_gen++;            // (image counters cleared by a reorder before this are gone: halo_borders_multi_async)
    send_cap = cap;
    chunk_cap = (cap + 255) / 256 + 1;
    HIPCHK(regrow(chunk_count, 0, (size_t)27 * chunk_cap + 1, stream));
    HIPCHK(regrow(chunk_offset, 0, (size_t)27 * chunk_cap + 1, stream));
    size_t tb = std::max(sort_temp_bytes_u32(cap), sort_temp_bytes_u64(cap));
    tb = std::max(tb, scan_temp_bytes(27 * chunk_cap + 1));
    if (tb > scan_temp_side_bytes) {       // the side stream's scans (overlapped rebuild) need their own scratch
        if (scan_temp_side) (void)hipFree(scan_temp_side);
        scan_temp_side = nullptr;
        HIPCHK(hipMalloc(&scan_temp_side, tb));
        scan_temp_side_bytes = tb;
    }
    if (tb > sort_temp_bytes) {
        if (sort_temp) (void)hipFree(sort_temp);
        sort_temp = nullptr;
        HIPCHK(hipMalloc(&sort_temp, tb));
        sort_temp_bytes = tb;
    }
    if (n_col > 0) {
        table_tiles = ((size_t)cap + 63) / 64;
        dfree(pair_table); dfree(pair_back);
        HIPCHK(dalloc(pair_table, table_tiles * 64 * (size_t)n_col));      // (pair_back: when a build first writes two sections)
        dfree(brick_own);
        HIPCHK(dalloc(brick_own, table_tiles * 64));
    }
    nmax = cap;
    return 0;
}

int Engine::alloc_topology(AtomSoA &a, int cap, int keep)
{
    // arrays laid out with other per-atom widths cannot be carried over (a second bonds_upload / read_restart on a context
    // that already held topology): they are allocated fresh and filled again by the caller
    const size_t c = (size_t)cap;
    const size_t kb = (a.bpa == bpa && a.msp == msp) ? (size_t)keep : 0, ka = (a.apa == apa) ? (size_t)keep : 0;
    a.bpa = bpa; a.msp = msp;
    HIPCHK(regrow(a.nbond, kb, c, stream));
    HIPCHK(regrow(a.bond_tag, kb * std::max(bpa, 1), c * std::max(bpa, 1), stream));
    HIPCHK(regrow(a.bond_type, kb * std::max(bpa, 1), c * std::max(bpa, 1), stream));
    HIPCHK(regrow(a.nspecial, kb, c, stream));
    HIPCHK(regrow(a.special, kb * std::max(msp, 1), c * std::max(msp, 1), stream));
    a.apa = apa;
    if (apa > 0) {
        HIPCHK(regrow(a.nangle, ka, c, stream));
        HIPCHK(regrow(a.angle_tag, ka * 4 * apa, c * 4 * apa, stream));
    }
    return 0;
}

// special_bonds lj w12 w13 w14: like filter_exclusion_meso (neigh_build_meso.cu:546-569) a level is either kept or
// removed from the pair rows; fractional weights are not supported on this path
int Engine::special_bonds(double w12, double w13, double w14)
{
    if (have_bonds) return fail(3, "special_bonds must precede the bond list");
    special_w[0] = w12; special_w[1] = w13; special_w[2] = w14;
    for (int k = 0; k < 3; k++)
        if (special_w[k] != 0.0 && special_w[k] != 1.0) return fail(1, "special_bonds weights must be 0 or 1 on the meso path");
    return 0;
}

int Engine::bond_style(int nbt, int kind)
{
    if (nbt < 1) return fail(1, "Illegal bond_style command");
    if (kind != 0 && kind != 1) return fail(1, "Invalid bond style");
    nbondtypes = nbt;
    bond_kind = kind;
    bond_kr0.assign(4 * (size_t)(nbt + 1), 0.0);
    dfree(d_bond_kr0);
    return 0;
}

// BondHarmonic::coeff (src/MOLECULE/bond_harmonic.cpp): bond_coeff type K r0
// BondFENE::coeff (src/MOLECULE/bond_fene.cpp:143-167, bond_fene_meso.cu:36-50): bond_coeff type K R0 epsilon sigma
int Engine::bond_coeff(int type, double k, double r0, double eps, double sigma)
{
    if (nbondtypes == 0) return fail(3, "bond_coeff before bond_style");
    if (type < 1 || type > nbondtypes) return fail(1, "Incorrect args for bond coefficients");
    bond_kr0[type] = k;
    bond_kr0[nbondtypes + 1 + type] = r0;
    bond_kr0[2 * (nbondtypes + 1) + type] = eps;
    bond_kr0[3 * (nbondtypes + 1) + type] = sigma;
    dfree(d_bond_kr0);
    return 0;
}

// Bonds section of the data file: (tag_i, tag_j, type).  Every rank is handed the whole list and keeps, for the
// atoms it owns, the bond on BOTH atoms (newton off, atom_vec_dpd_bond_meso.cu) plus the 1-2/1-3/1-4 partner tags
// of the levels that special_bonds removes (Special::build, src/special.cpp).
int Engine::bonds_upload(int nb, const int *ti, const int *tj, const int *bt)
{
    if (nb < 0 || (nb && (!ti || !tj || !bt))) return fail(1, "Invalid bond arrays");
    if (nlocal > 0 && h_tags.empty()) return fail(3, "Bonds must follow the atoms");
    maxtag = 0;
    for (int b = 0; b < nb; b++) maxtag = std::max(maxtag, std::max(ti[b], tj[b]));
    for (int t : h_tags) maxtag = std::max(maxtag, t);
    std::vector<std::vector<std::pair<int, int>>> adj((size_t)maxtag + 1);
    for (int b = 0; b < nb; b++) {
        if (ti[b] < 1 || tj[b] < 1 || ti[b] == tj[b]) return fail(1, "Invalid atom ID in Bonds section of data file");
        if (nbondtypes && (bt[b] < 1 || bt[b] > nbondtypes)) return fail(1, "Invalid bond type in Bonds section of data file");
        adj[ti[b]].push_back(std::make_pair(tj[b], bt[b]));
        adj[tj[b]].push_back(std::make_pair(ti[b], bt[b]));
    }
    int excl_levels = 0;   // consecutive levels with weight 0, starting at 1-2
    while (excl_levels < 3 && special_w[excl_levels] == 0.0) excl_levels++;
    // special partners by breadth-first search to depth excl_levels
    auto specials = [&](int t) {
        std::vector<int> out, frontier(1, t), seen(1, t);
        for (int lev = 0; lev < excl_levels; lev++) {
            std::vector<int> next;
            for (int u : frontier)
                for (auto &pr : adj[u])
                    if (std::find(seen.begin(), seen.end(), pr.first) == seen.end()) { seen.push_back(pr.first); next.push_back(pr.first); out.push_back(pr.first); }
            frontier.swap(next);
        }
        return out;
    };
    // array widths come from the WHOLE list (every rank holds it), so all ranks agree on the message layout
    int new_bpa = 0, new_msp = 0;
    for (int t = 1; t <= maxtag; t++) {
        if (adj[t].empty()) continue;
        new_bpa = std::max(new_bpa, (int)adj[t].size());
        new_msp = std::max(new_msp, (int)specials(t).size());
    }
    int n = nlocal;
    std::vector<std::vector<int>> spec((size_t)n);
    for (int i = 0; i < n; i++) spec[i] = specials(h_tags[i]);
    bpa = std::max(new_bpa, 1); msp = std::max(new_msp, 1);
    TRY(alloc_atoms(std::max(nmax, 1024)));   // (re)allocates with the topology arrays
    std::vector<int> hn((size_t)n, 0), hbt((size_t)n * bpa, 0), hty((size_t)n * bpa, 0), hs((size_t)n, 0), hsp((size_t)n * msp, 0);
    for (int i = 0; i < n; i++) {
        auto &a = adj[h_tags[i]];
        hn[i] = (int)a.size();
        for (size_t b = 0; b < a.size(); b++) { hbt[(size_t)i * bpa + b] = a[b].first; hty[(size_t)i * bpa + b] = a[b].second; }
        hs[i] = (int)spec[i].size();
        for (size_t s = 0; s < spec[i].size(); s++) hsp[(size_t)i * msp + s] = spec[i][s];
    }
    if (n) {
        HIPCHK(hipMemcpy(cur.nbond, hn.data(), hn.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(cur.bond_tag, hbt.data(), hbt.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(cur.bond_type, hty.data(), hty.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(cur.nspecial, hs.data(), hs.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(cur.special, hsp.data(), hsp.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    dfree(tagmap);
    HIPCHK(dalloc(tagmap, (size_t)maxtag + 2));
    // the tags a rebuild has to find: the ends of the bonds (angles add theirs); every rank holds the whole list
    h_tagbits.assign(((size_t)maxtag + 32) / 32, 0u);
    for (int b = 0; b < nb; b++) { h_tagbits[ti[b] >> 5] |= 1u << (ti[b] & 31); h_tagbits[tj[b] >> 5] |= 1u << (tj[b] & 31); }
    TRY(upload_tagbits());
    have_bonds = true;
    is_setup = false;
    return 0;
}

int Engine::upload_tagbits()
{
    dfree(tagbits);
    if (h_tagbits.empty()) return 0;
    HIPCHK(dalloc(tagbits, h_tagbits.size()));
    HIPCHK(hipMemcpy(tagbits, h_tagbits.data(), h_tagbits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return 0;
}

// map_set_device + bond_all (atom_meso.cu:108-126, neighbor_meso.cu:130-): after every rebuild
int Engine::rebuild_topology()
{
    if (!have_bonds) return 0;
    const int *ng_dev = pending_nghost_dev();     // nghost is a launch bound while the counts travel
    launch_tag_cell(cur.tag, (fused_active || mr_runs) ? nullptr : gslot, nlocal, nghost, ng_dev, tagc, stream);
    HIPCHK(hipMemsetAsync(tagmap, 0x7f, ((size_t)maxtag + 2) * sizeof(int), stream));
    launch_set_map(tagc, nlocal, nghost, ng_dev, maxtag, tagbits, tagmap, stream);
    // (d_flags[4] counts partners that are neither local nor ghost; it stays set until check_overflow reports it)
    launch_map_bonds(cur.nbond, cur.bond_tag, bpa, tagmap, maxtag, nlocal, bond_idx, d_flags + 4, stream);
    if (have_angles)
        launch_map_angles(cur.tag, cur.nangle, cur.angle_tag, apa, tagmap, maxtag, nlocal, angle_idx, d_flags + 4, stream);
    return 0;
}

// BondHarmonic::compute / gpu_bond_harmonic (bond_harmonic_meso.cu:46-117), MesoBondFENE::compute (bond_fene_meso.cu:150-214)
int Engine::bond_compute(int eflag, int store)
{
    if (!have_bonds || nbondtypes == 0) return 0;
    if (!d_bond_kr0) {
        HIPCHK(dalloc(d_bond_kr0, bond_kr0.size()));
        HIPCHK(hipMemcpy(d_bond_kr0, bond_kr0.data(), bond_kr0.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    tbegin("bond");
    launch_bond(bond_kind, coord4, cur.nbond, bond_idx, cur.bond_type, bpa, d_bond_kr0, nbondtypes, prd, nlocal, cur.f[0], cur.f[1],
                cur.f[2], eflag ? e_bond : nullptr, store, stream);
    tend("bond");
    return 0;
}

int Engine::compute_ebond(double *e)
{
    if (!have_bonds || nbondtypes == 0 || !is_setup || !d_bond_kr0) { *e = 0.0; return 0; }
    // energy at the coordinates of the last force evaluation (what thermo prints on an eflag step, src/thermo.cpp ebond)
    launch_bond(bond_kind, coord4, cur.nbond, bond_idx, cur.bond_type, bpa, d_bond_kr0, nbondtypes, prd, nlocal, nullptr, nullptr,
                nullptr, e_bond, 0, stream);
    std::vector<double> h((size_t)nlocal);
    HIPCHK(hipMemcpyAsync(h.data(), e_bond, nlocal * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    double s = 0.0;
    for (double v : h) s += v;
    *e = reduce_global_sum(s);
    return 0;
}

// Angles section of the data file: (tag1, tag2, tag3, type), tag2 the apex.  Like the bonds, every rank is handed the whole
// list and keeps an angle on all three of its atoms (newton off, atom_vec_dpd_angle_meso.cu); needs the Bonds section first
// (tag map, special lists and message layout are set up there).
int Engine::angles_upload(int na, const int *t1, const int *t2, const int *t3, const int *ty)
{
    if (na < 0 || (na && (!t1 || !t2 || !t3 || !ty))) return fail(1, "Invalid angle arrays");
    if (!have_bonds) return fail(3, "Angles must follow the Bonds section");
    std::vector<std::vector<int>> per((size_t)maxtag + 1);
    for (int a = 0; a < na; a++) {
        const int t[3] = {t1[a], t2[a], t3[a]};
        for (int c = 0; c < 3; c++)
            if (t[c] < 1 || t[c] > maxtag) return fail(1, "Invalid atom ID in Angles section of data file");
        if (t[0] == t[1] || t[1] == t[2] || t[0] == t[2]) return fail(1, "Invalid atom ID in Angles section of data file");
        if (nangletypes && (ty[a] < 1 || ty[a] > nangletypes)) return fail(1, "Invalid angle type in Angles section of data file");
        for (int c = 0; c < 3; c++) per[t[c]].push_back(a);
        if (!h_tagbits.empty()) for (int c = 0; c < 3; c++) h_tagbits[t[c] >> 5] |= 1u << (t[c] & 31);
    }
    TRY(upload_tagbits());
    int new_apa = 0;
    for (auto &v : per) new_apa = std::max(new_apa, (int)v.size());
    apa = std::max(new_apa, 1);
    TRY(alloc_atoms(std::max(nmax, 1024)));
    const int n = nlocal;
    std::vector<int> hn((size_t)n, 0), ht((size_t)n * 4 * apa, 0);
    for (int i = 0; i < n; i++) {
        auto &v = per[h_tags[i]];
        hn[i] = (int)v.size();
        for (size_t q = 0; q < v.size(); q++) {
            int *o = &ht[((size_t)i * apa + q) * 4];
            o[0] = t1[v[q]]; o[1] = t2[v[q]]; o[2] = t3[v[q]]; o[3] = ty[v[q]];
        }
    }
    if (n) {
        HIPCHK(hipMemcpy(cur.nangle, hn.data(), hn.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(cur.angle_tag, ht.data(), ht.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    have_angles = true;
    is_setup = false;
    return 0;
}

int Engine::angle_style(int nat)
{
    if (nat < 1) return fail(1, "Illegal angle_style command");
    nangletypes = nat;
    angle_cf.assign(2 * (size_t)(nat + 1), 0.0);
    dfree(d_angle_cf);
    return 0;
}

// AngleHarmonic::coeff (src/MOLECULE/angle_harmonic.cpp:157-181): angle_coeff type K theta0[degrees]
int Engine::angle_coeff(int type, double k, double theta0_deg)
{
    if (nangletypes == 0) return fail(3, "angle_coeff before angle_style");
    if (type < 1 || type > nangletypes) return fail(1, "Incorrect args for angle coefficients");
    angle_cf[type] = k;
    angle_cf[nangletypes + 1 + type] = theta0_deg / 180.0 * 3.14159265358979323846;
    dfree(d_angle_cf);
    return 0;
}

// MesoAngleHarmonic::compute (angle_harmonic_meso.cu:174-236)
int Engine::angle_compute(int eflag)
{
    if (!have_angles || nangletypes == 0) return 0;
    if (!d_angle_cf) {
        HIPCHK(dalloc(d_angle_cf, angle_cf.size()));
        HIPCHK(hipMemcpy(d_angle_cf, angle_cf.data(), angle_cf.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    tbegin("angle");
    launch_angle_harmonic(coord4, cur.nangle, angle_idx, cur.angle_tag, apa, d_angle_cf, nangletypes, prd, nlocal, cur.f[0],
                          cur.f[1], cur.f[2], eflag ? e_angle : nullptr, stream);
    tend("angle");
    return 0;
}

int Engine::compute_eangle(double *e)
{
    if (!have_angles || nangletypes == 0 || !is_setup || !d_angle_cf) { *e = 0.0; return 0; }
    launch_angle_harmonic(coord4, cur.nangle, angle_idx, cur.angle_tag, apa, d_angle_cf, nangletypes, prd, nlocal, nullptr,
                          nullptr, nullptr, e_angle, stream);
    std::vector<double> h((size_t)nlocal);
    HIPCHK(hipMemcpyAsync(h.data(), e_angle, nlocal * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    double s = 0.0;
    for (double v : h) s += v;
    *e = reduce_global_sum(s);
    return 0;
}

int Engine::ensure_capacity(int need)
{
    if (need <= nmax) return 0;
    int cap = (int)(need * 1.15) + 4096;
    TRY(alloc_atoms(cap));
    // neighbour table contents are rebuilt by the caller
    return 0;
}

int Engine::atoms_upload(int n, const double *x, const double *v, const int *tag, const int *type, const int *mask,
                         const int *image)
{
    if (n < 0 || (n > 0 && (!x || !v || !tag || !type))) return fail(1, "Invalid atom arrays");      // a rank may start empty
    if (!have_box) return fail(3, "Box must be set before atoms are created");
    // a new deck: the estimates and message capacities that earlier rebuilds left behind say nothing about it
    nghost_prev = -1; n_bulk_prev = -1; mr_caps_ready = false; mig_caps_ready = false; counts_pending = false; mr_pending = false; bulk_pending = false;
    // capacity: locals + expected ghosts (periodic images within cutghost) with head-room
    double ext = 1.0;
    double cg = (cut_global > 0 ? cut_global : 1.0) + skin;
    for (int d = 0; d < 3; d++) ext *= (prd[d] / procgrid[d] + 2.0 * cg) / (prd[d] / procgrid[d]);
    int cap = (int)(1.25 * n * std::min(ext, 27.0)) + 8192;
    nlocal = 0;
    // several ranks: every rank is handed the whole deck and keeps the atoms of its own sub-box
    std::vector<double> fx, fv;
    std::vector<int> ftag, ftype, fmask, fimage;
    if (nranks > 1 && !upload_all) {
        for (int i = 0; i < n; i++) {
            if (!owns(x + 3 * (size_t)i)) continue;
            for (int d = 0; d < 3; d++) { fx.push_back(x[3 * (size_t)i + d]); fv.push_back(v[3 * (size_t)i + d]); }
            ftag.push_back(tag[i]); ftype.push_back(type[i]);
            if (mask) fmask.push_back(mask[i]);
            if (image) fimage.push_back(image[i]);
        }
        n = (int)ftag.size();
        x = fx.data(); v = fv.data(); tag = ftag.data(); type = ftype.data();
        if (mask) mask = fmask.data();
        if (image) image = fimage.data();
        cap = (int)(1.25 * n * std::min(ext, 27.0)) + 8192;
    }
    if (cap > nmax) TRY(alloc_atoms(cap));
    std::vector<double> tmp((size_t)std::max(n, 1));
    for (int d = 0; d < 3; d++) {
        for (int i = 0; i < n; i++) tmp[i] = x[3 * (size_t)i + d];
        HIPCHK(hipMemcpy(cur.x[d], tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < n; i++) tmp[i] = v[3 * (size_t)i + d];
        HIPCHK(hipMemcpy(cur.v[d], tmp.data(), n * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(hipMemsetAsync(cur.f[d], 0, n * sizeof(double), stream));
    }
    HIPCHK(hipMemcpy(cur.tag, tag, n * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(cur.type, type, n * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> itmp((size_t)std::max(n, 1), 1);
    HIPCHK(hipMemcpy(cur.mask, mask ? mask : itmp.data(), n * sizeof(int), hipMemcpyHostToDevice));
    // image flags: 10 bits per dimension, 512 = no wrap (LAMMPS' smallint packing)
    std::fill(itmp.begin(), itmp.end(), 512 | (512 << 10) | (512 << 20));
    HIPCHK(hipMemcpy(cur.image, image ? image : itmp.data(), n * sizeof(int), hipMemcpyHostToDevice));
    for (int i = 0; i < n; i++)
        if (type[i] < 1 || (ntypes && type[i] > ntypes)) return fail(1, "Invalid atom type in atom arrays");
    h_tags.assign(tag, tag + n);
    have_bonds = false;
    have_angles = false;
    restart_forces = false;
    nlocal = n;
    nghost = 0;
    n_bulk = 0;
    is_setup = false;
    params_ready = false;
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
}

int Engine::atoms_download(double *x, double *v, double *f, int *tag, int *type, int *image)
{
    HIPCHK(hipStreamSynchronize(stream));
    int n = nlocal;
    std::vector<double> tmp((size_t)n);
    double *dst[3] = {x, v, f};
    for (int a = 0; a < 3; a++) {
        if (!dst[a]) continue;
        for (int d = 0; d < 3; d++) {
            double *src = a == 0 ? cur.x[d] : (a == 1 ? cur.v[d] : cur.f[d]);
            HIPCHK(hipMemcpy(tmp.data(), src, n * sizeof(double), hipMemcpyDeviceToHost));
            for (int i = 0; i < n; i++) dst[a][3 * (size_t)i + d] = tmp[i];
        }
    }
    if (tag) HIPCHK(hipMemcpy(tag, cur.tag, n * sizeof(int), hipMemcpyDeviceToHost));
    if (type) HIPCHK(hipMemcpy(type, cur.type, n * sizeof(int), hipMemcpyDeviceToHost));
    if (image) HIPCHK(hipMemcpy(image, cur.image, n * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// parameters: sub-box, slabs, bins (MesoNeighbor::setup_bins neighbor_meso.cu:858-931), row width
// ------------------------------------------------------------------------------------------------
int Engine::init_params()
{
    if (params_ready) return 0;
    if (!have_box) return fail(3, "Box has not been set");
    if (!have_pair || !have_coeff) return fail(3, "Pair style/coefficients have not been set");
    if (ntypes < 1 || (int)mass_type.size() != ntypes + 1) return fail(3, "Masses have not been set");
    if (ntypes > 255) return fail(1, "More than 255 atom types are not supported (the force kernel queues partner types as bytes)");
    for (int i = 0; i < ntypes * ntypes; i++)
        if (!coeff_set[i]) return fail(3, "All pair coeffs are not set");
    cutmax = 0.0;
    for (int i = 0; i < ntypes * ntypes; i++) cutmax = std::max(cutmax, coeff[(size_t)i * N_COEFF + P_CUT]);
    cutghost = cutmax + skin;
    const double cutneighmax = cutmax + skin;

    {
        // sub-box, border slabs and the 26 neighbours (rank, periodic shift, centre of its sub-box): decomp_plan (comm.hip), the
        // same host function the C ABI exports as meso_decomp_plan for the CPU-side decomposition tests
        int act[27];
        if (decomp_plan(boxlo, boxhi, periodic, procgrid, myloc, cutghost, sublo, subhi, slab_lo, slab_hi, peer27, act, shift27, center27))
            return fail(1, "Sub-domain smaller than the ghost cutoff is not supported");
        for (int dir = 0; dir < 27; dir++) send_active[dir] = act[dir] != 0;
    }
    build_peer_tables();
    if (!d_shift27) HIPCHK(dalloc(d_shift27, 81));
    HIPCHK(hipMemcpy(d_shift27, shift27, 81 * sizeof(double), hipMemcpyHostToDevice));
    if (!d_center27) HIPCHK(dalloc(d_center27, 81));
    HIPCHK(hipMemcpy(d_center27, center27, 81 * sizeof(double), hipMemcpyHostToDevice));

    // bins aligned with the sub-box, one ghost layer each side
    double subvol = 1.0;
    for (int d = 0; d < 3; d++) {
        double dim = subhi[d] - sublo[d];
        subvol *= dim;
        geom.lo[d] = sublo[d]; geom.hi[d] = subhi[d];
        geom.mbin[d] = std::max((int)(dim * (1.0 / cutneighmax)), 1) + 2;
        geom.binsize[d] = dim / (geom.mbin[d] - 2);
        geom.bininv[d] = 1.0 / geom.binsize[d];
    }
    geom.nbin = geom.mbin[0] * geom.mbin[1] * geom.mbin[2];
    double density = nlocal / subvol;
    if (density < 3.0) density = 3.0;
    double expected = density * (4.0 / 3.0 * 3.142 * std::pow(cutneighmax, 3.0)) * 4.0;
    expected = std::max(expected, 32.0);
    int ncol = ((int)std::ceil(expected) + 31) / 32 * 32;
    if (ncol != n_col || !pair_table) {
        n_col = ncol;
        nb_col = std::max(8, (n_col / 2 + 7) & ~7);      // back section of a partitioned row: about half of the in-group partners, at most half a row
        table_tiles = ((size_t)nmax + 63) / 64;
        dfree(pair_table); dfree(pair_back);
        HIPCHK(dalloc(pair_table, table_tiles * 64 * (size_t)n_col));      // (pair_back: when a build first writes two sections)
        dfree(brick_own);
        HIPCHK(dalloc(brick_own, table_tiles * 64));
    }
    {
        int max_bin = std::max(geom.mbin[0], std::max(geom.mbin[1], geom.mbin[2]));
        l1bits = 0;
        while ((1 << (l1bits + 1)) <= max_bin * 2) l1bits++;
        size_t M = (size_t)1 << (3 * l1bits);
        if (2 * M + 1 > estart_cap) {
            dfree(estart); dfree(gstart);
            estart_cap = 2 * M + 1;
            HIPCHK(dalloc(estart, estart_cap));
            HIPCHK(dalloc(gstart, M + 1));
            // counts per code: zeroed once here, the ordering passes leave them clean for the next rebuild
            dfree(gcount);
            HIPCHK(dalloc(gcount, M + 1));
            HIPCHK(hipMemsetAsync(gcount, 0, (M + 1) * sizeof(int), stream));
            dfree(rcount);
            HIPCHK(dalloc(rcount, 2 * M + 1));
            HIPCHK(hipMemsetAsync(rcount, 0, (2 * M + 1) * sizeof(int), stream));
            dfree(binrange);
            HIPCHK(dalloc(binrange, 2 * M));
            size_t tb = scan_temp_bytes((int)(2 * M + 2));  // atoms per extended code (2M + 1) is the longest scan
            if (tb > scan_temp_side_bytes) {
                if (scan_temp_side) (void)hipFree(scan_temp_side);
                scan_temp_side = nullptr;
                HIPCHK(hipMalloc(&scan_temp_side, tb));
                scan_temp_side_bytes = tb;
            }
            if (tb > sort_temp_bytes) {
                if (sort_temp) (void)hipFree(sort_temp);
                sort_temp = nullptr;
                HIPCHK(hipMalloc(&sort_temp, tb));
                sort_temp_bytes = tb;
            }
        }
        // halo capacity of a brick (6x6x6 bins): mean + 6.5 sigma of a Poisson count at this density, times the option
        // brick_margin for inhomogeneous systems.  The brick-layout kernels have static LDS arrays; the tile builder
        // sizes its LDS at launch (occupancy drops from 3 to 2 workgroups per CU above ~2340 atoms) and falls back to the
        // lane-per-atom builder when even one workgroup per CU could not hold the neighbourhood.
        {
            const double binvol = geom.binsize[0] * geom.binsize[1] * geom.binsize[2];
            const double mean = density * 216.0 * binvol * brick_margin;
            int want = ((int)std::ceil(mean + 6.5 * std::sqrt(mean)) + 63) / 64 * 64;
            if (want < brick_static_maxh()) want = brick_static_maxh();
            // the fullest neighbourhood seen so far + 8 % (every 2.5 KB of LDS beyond the need can cost the third workgroup per CU)
            if (want < brick_maxh_floor) want = brick_maxh_floor;
            tile_fits = want <= tile_build_maxh_limit(n_col, have_bonds && msp > 0 ? 1 : 0);
            bargs.maxh = want;
            {
                // the 2x2x2 brick of small boxes: 64 bins, scaled like the 4-brick stage when that one had to grow
                const double mean2 = density * 64.0 * binvol * brick_margin;
                const double grow = std::max(1.0, (double)want / std::max(1.0, std::ceil(mean + 6.5 * std::sqrt(mean))));
                bargs.maxh2 = ((int)std::ceil((mean2 + 7.0 * std::sqrt(mean2)) * grow) + 63) / 64 * 64;
                // eight 4-wave workgroups fit a CU whatever they stage up to 20 KB each: the head-room is free (a 4^3-bin
                // neighbourhood feels a local compression more than a 6^3-bin one does)
                const int roomy = (20 * 1024 - 1024 - 4 * 4 * n_col * 2) / 16 / 64 * 64;      // (1 KB: the kernel's static LDS)
                if (bargs.maxh2 < roomy) bargs.maxh2 = roomy;
                if (bargs.maxh2 < brick2_floor) bargs.maxh2 = brick2_floor;      // (grown during the run, see reneighbor)
            }
            // LDS stage of the reorder's ordering pass: the atoms of 128 consecutive extended codes (mean + 6.5 sigma, margin)
            const double m128 = density * 128.0 * binvol * brick_margin * std::max(1.0, (double)brick_maxh_floor / std::max(1.0, mean + 6.5 * std::sqrt(mean)));
            reorder_cap = std::min(7680, std::max(2048, ((int)std::ceil(m128 + 6.5 * std::sqrt(m128)) + 63) / 64 * 64));
            if (reorder_cap_user > 0) reorder_cap = reorder_cap_user;
            // bucket of one code in the fused rebuild: atoms of one bin, mean + 6.5 sigma + 8 (rarer: the overflow list)
            const double m1 = density * binvol * brick_margin * std::max(1.0, (double)brick_maxh_floor / std::max(1.0, mean + 6.5 * std::sqrt(mean)));
            fr_cap_want = ((int)std::ceil(m1 + 6.5 * std::sqrt(m1)) + 8 + 7) / 8 * 8;
            bargs.maxown = 0;
        }
        if ((M / brick_codes() + 8 > brick_cap || bargs.maxh != brick_maxh_alloc)) {
            dfree(brick_hoff); dfree(brick_hmap); dfree(brick_hdr);
            brick_cap = M / brick_codes() + 8;
            brick_maxh_alloc = bargs.maxh;
            HIPCHK(dalloc(brick_hoff, brick_cap * brick_hoff_pitch()));
            HIPCHK(dalloc(brick_hmap, brick_cap * (size_t)bargs.maxh));
            HIPCHK(dalloc(brick_hdr, brick_cap * brick_hdr_pitch()));
        }
        {
            // launch order of the 2x2x2 bricks: those that own real cells (coordinates 1 .. mbin - 2), fullest first, Morton order
            // inside a class.  A brick at a face of the bin grid owns half the cells of an interior one, at an edge a quarter: as
            // the launch's tail they cost a fraction of a brick time (32^3: 2197 bricks on 2048 workgroup slots - the 149 that
            // wait for a slot were whole bricks before: 50 -> 3x us per build)
            std::vector<std::pair<int, int>> ord;
            const int nb2 = (int)(M / 8);
            for (int B = 0; B < nb2; B++) {
                const u32 c0 = (u32)B * 8u;
                int b0[3] = {0, 0, 0};
                for (int bit = 0; bit < 10; bit++)
                    for (int d = 0; d < 3; d++) b0[d] |= (int)((c0 >> (3 * bit + d)) & 1u) << bit;
                int w = 1;
                for (int d = 0; d < 3; d++) {
                    int nd = 0;
                    for (int k = 0; k < 2; k++) nd += (b0[d] + k >= 1 && b0[d] + k <= geom.mbin[d] - 2) ? 1 : 0;
                    w *= nd;
                }
                if (w > 0) ord.push_back(std::make_pair(-w, B));
            }
            std::stable_sort(ord.begin(), ord.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
            std::vector<int> h(ord.size());
            for (size_t k = 0; k < ord.size(); k++) h[k] = ord[k].second;
            dfree(brick_order2);
            HIPCHK(dalloc(brick_order2, std::max<size_t>(h.size(), 1)));
            HIPCHK(hipMemcpy(brick_order2, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
            bargs.order2 = brick_order2; bargs.norder2 = (int)h.size();
            // tiles of the fused rebuild's ghost kernel that hold a ghost cell of the bin grid (a coordinate 0 or mbin - 1)
            std::vector<int> gt;
            const int gtile = fused_gtile_codes();
            for (int t = 0; t < (int)(M / gtile); t++) {
                bool any = false;
                for (int k = 0; k < gtile && !any; k++) {
                    const u32 c = (u32)t * (u32)gtile + (u32)k;
                    int b[3] = {0, 0, 0};
                    for (int bit = 0; bit < 10; bit++)
                        for (int d = 0; d < 3; d++) b[d] |= (int)((c >> (3 * bit + d)) & 1u) << bit;
                    const bool inside = b[0] < geom.mbin[0] && b[1] < geom.mbin[1] && b[2] < geom.mbin[2];
                    any = inside && (b[0] == 0 || b[0] == geom.mbin[0] - 1 || b[1] == 0 || b[1] == geom.mbin[1] - 1 || b[2] == 0 || b[2] == geom.mbin[2] - 1);
                }
                if (any) gt.push_back(t);
            }
            dfree(fr_gorder); dfree(fr_gcnt);
            HIPCHK(dalloc(fr_gorder, std::max<size_t>(gt.size(), 1)));
            HIPCHK(hipMemcpy(fr_gorder, gt.data(), gt.size() * sizeof(int), hipMemcpyHostToDevice));
            fr_ngorder = (int)gt.size();
            HIPCHK(dalloc(fr_gcnt, M + 1));
            HIPCHK(hipMemset(fr_gcnt, 0, (M + 1) * sizeof(int)));
        }
        bargs.estart = estart; bargs.gstart = gstart; bargs.M = (int)M; bargs.nbricks = (int)(M / brick_codes());
        bargs.hoff = brick_hoff; bargs.hmap = brick_hmap; bargs.hdr = brick_hdr; bargs.own_info = brick_own;
        for (int d = 0; d < 3; d++) {
            bargs.mbin[d] = geom.mbin[d];
            bargs.org[d] = (float)(geom.lo[d] - geom.binsize[d]); bargs.binw[d] = (float)geom.binsize[d];
        }
    }
    img_ok = true;
    for (int d = 0; d < 3; d++)
        if (periodic[d] && !(prd[d] > 2.0 * cutghost)) img_ok = false;
    // coefficient tables (prepare_coeff pair_dpd_meso.cu:68-89)
    dfree(d_coeff64); dfree(d_coeff32); dfree(d_mass_type); dfree(d_dtfm_type);
    dtfm_for = -1.0;
    HIPCHK(dalloc(d_coeff64, coeff.size()));
    HIPCHK(dalloc(d_coeff32, coeff.size()));
    HIPCHK(dalloc(d_mass_type, mass_type.size()));
    HIPCHK(dalloc(d_dtfm_type, mass_type.size()));
    std::vector<float> c32(coeff.begin(), coeff.end());
    HIPCHK(hipMemcpy(d_coeff64, coeff.data(), coeff.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_coeff32, c32.data(), c32.size() * sizeof(float), hipMemcpyHostToDevice));
    dfree(d_ftab);
    if (pair_ftab) {
        if (ftab.size() != (size_t)ntypes * ntypes * ftab_len || !ftab_len) return fail(3, "All pair coeffs are not set");
        HIPCHK(dalloc(d_ftab, ftab.size()));
        HIPCHK(hipMemcpy(d_ftab, ftab.data(), ftab.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    dfree(d_poly);
    if (pair_poly) {
        if (poly.size() != (size_t)ntypes * ntypes * MESO_POLY_PITCH) return fail(3, "All pair coeffs are not set");
        HIPCHK(dalloc(d_poly, poly.size()));
        HIPCHK(hipMemcpy(d_poly, poly.data(), poly.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(d_mass_type, mass_type.data(), mass_type.size() * sizeof(double), hipMemcpyHostToDevice));
    launch_unpack_mass(cur.type, d_mass_type, ntypes, cur.mass, 0, nlocal, stream);
    // (a capacity regrowth re-runs this on the ranks that need it only: no collective then - the atom count is conserved)
    if (!regrow_only || natoms_total == 0) natoms_total = (long)reduce_global_sum((double)nlocal);
    regrow_only = false;
    params_ready = true;
    return 0;
}



void Engine::range(int r, int &beg, int &end) const
{
    beg = 0; end = nlocal;
    if (r == 1) end = n_bulk;
    else if (r == 2) beg = n_bulk;
}

// ------------------------------------------------------------------------------------------------
// rebuild pieces
// ------------------------------------------------------------------------------------------------

// MesoAtom::sort_local (atom_meso.cu:343-384) + transfer_post_sort, all device resident
// the locals' reorder as count + place/gather of the fused rebuild (n: atoms the count walks)
bool Engine::reorder_fuses(long n) const { return !reorder_sort && fused_rebuild && neigh_kernel == 1 && n < (1L << 27); }

int Engine::reorder_locals()
{
    merged_in_reorder = false;
    if (nlocal == 0) {
        mig_holes = false;
        n_bulk = 0;
        launch_estart(rkey, 0, reorder_sub_bits(geom), 2 * bargs.M, estart, stream);   // an empty rank: all zero
        return 0;
    }
    tbegin("reorder");
    int bits = reorder_key_bits(geom);
    bool gathered = false;
    if (!reorder_sort) {
        // counting per extended code instead of a comparison sort (kernels.hip): ~8 launches instead of ~28; estart - first
        // index of every extended code ([border][Morton(bin)]), the border section starts at estart[M] = n_bulk - is the scan
        if (reorder_fuses(mig_holes ? mig_span : nlocal)) {
            // count + place/gather (rebuild.hip, the locals' half of the fused rebuild): two launches for count, scan x2, place,
            // order and gather; estart and the merged pairs come with it (several ranks, and the first rebuild of one rank)
            TRY(fused_alloc());
            FusedArgs a;
            fused_locals_args(a);
            fr_epoch++;
            // (several ranks without a host round trip in the ghost stage: the border count that follows clears the overflow count)
            a.novf_later = (nranks > 1 && mr_async_ok()) ? 1 : 0;
            launch_fused_rebuild(a, stream);
            novf_pending = a.novf_later != 0;
            mig_holes = false;
            if (!async_ok()) HIPCHK(hipMemcpyAsync(d_flags + 1, estart + bargs.M, sizeof(int), hipMemcpyDeviceToDevice, stream));
            gathered = true;
        } else {
        const int ncodes = 2 * bargs.M;
        launch_reorder_count(cur, geom, slab_lo, slab_hi, rkey, rval_alt, rcount, nlocal, wrap_in_reorder ? boxlo : nullptr, boxhi, periodic,
                             stream);
        HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, rcount, estart, ncodes + 1, stream));
        launch_reorder_place(rkey, rval_alt, estart, geom, ncodes, nlocal, reorder_cap, (int *)rkey_alt, rval, (uint32_t *)bin_key_alt, rcount,
                             stream);
        std::swap(rkey, bin_key_alt);            // sorted keys (the lane-per-atom list builder reads them)
        // n_bulk = estart[M]: the asynchronous rebuild reads it where it is (halo_borders), the others through d_flags[1]
        if (!async_ok()) HIPCHK(hipMemcpyAsync(d_flags + 1, estart + bargs.M, sizeof(int), hipMemcpyDeviceToDevice, stream));
        }
    } else {
        // option reorder_sort: the former path - one rocPRIM radix_sort_pairs on the full key (same resulting order)
        launch_reorder_keys(cur, geom, slab_lo, slab_hi, nullptr, rkey, rval, nlocal, stream);
        HIPCHK(sort_pairs_u32(sort_temp, sort_temp_bytes, rkey, rkey_alt, rval, rval_alt, nlocal, bits, stream));
        launch_estart(rkey, nlocal, reorder_sub_bits(geom), 2 * bargs.M, estart, stream);
        HIPCHK(hipMemcpyAsync(d_flags + 1, estart + bargs.M, sizeof(int), hipMemcpyDeviceToDevice, stream));
    }
    if (!gathered) {
        // the gather also writes the merged float4 pair of the new order, with the signatures of the current step
        launch_permute_merge(cur, alt, rval, nlocal, permute_forces ? 1 : 0, coord4, veloc4, 0.5 * (subhi[0] + sublo[0]),
                             0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]), premix_tea<64>((u32)seed, (u32)ntimestep), nullptr,
                             images_on() ? img_cnt : nullptr, stream);
    }
    merged_in_reorder = true;      // (alloc_atoms clears it: a regrown coord4 has lost the values)
    std::swap(cur, alt); ck_swapped = !ck_swapped;
    if (!(nranks == 1 && async_ok())) HIPCHK(hipMemcpyAsync(h_flags, d_flags, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
    tend("reorder");
    if ((nranks == 1 && (nlocal <= 524288 || async_ok())) || (nranks > 1 && nlocal <= 524288)) {
        // one rank, small box: the ghost-list pass below scans every local atom (bulk atoms have no flags), so the bulk
        // count is not needed yet; it arrives with that pass's own host round trip - one synchronisation per rebuild,
        // not two (+4 % at 25^3-32^3; above ~0.5 M atoms the longer scan costs more than the round trip).  With
        // async_counts neither round trip happens: the border scan starts from a bound (halo_borders)
        bulk_pending = true;
        return 0;
    }
    HIPCHK(hipStreamSynchronize(stream));
    if (h_flags[0]) return check_overflow();
    n_bulk = h_flags[1];
    return 0;
}

// One rank, no host round trip, two streams.  Which atoms become ghosts depends on the (wrapped) positions only, not on the
// storage order, so the two halves of a rebuild are independent until the very end:
//   main stream: reorder of the locals - count per extended code (wraps the coordinates), scan, place + order, gather
//                (which also leaves the inverse permutation);
//   side stream: border lists on the OLD order, ghost creation into the arrays of the NEW order (behind the locals), ghost
//                binning (count, scan, order -> gstart, gslot);
//   join:        the send list is translated to the new order (one small kernel) - the per-step refresh reads it.
// Ghost numbering differs from the serial path (ascending old instead of new index inside a direction); ghost SLOTS,
// neighbour sets and forces do not (fixed-point force sums are order independent).
int Engine::rebuild_overlapped()
{
    tbegin("reorder");
    const int ncodes = 2 * bargs.M;
    if (!ev_wrap) { HIPCHK(hipEventCreateWithFlags(&ev_wrap, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_ghosts, hipEventDisableTiming)); }
    // `bound` sizes the grids of the ghost kernels (they loop, so any count is covered); the hard limit is the capacity
    const int bound = (int)(nghost_prev * async_grid_scale) + 1024;
    TRY(ensure_capacity(nlocal + bound + bound / 2));
    // ---- main: key, rank and count per extended code; the periodic wrap happens here
    launch_reorder_count(cur, geom, slab_lo, slab_hi, rkey, rval_alt, rcount, nlocal, wrap_in_reorder ? boxlo : nullptr, boxhi, periodic, stream);
    HIPCHK(hipEventRecord(ev_wrap, stream));
    // ---- side: borders and ghosts from the old order
    HIPCHK(hipStreamWaitEvent(side, ev_wrap, 0));
    const int end = nlocal, nchunk = (end + 255) / 256;
    launch_border_count(cur, 0, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, side);
    if (!h_flags_dev) HIPCHK(hipHostGetDevicePointer((void **)&h_flags_dev, h_flags, 0));
    const int cap_bound = std::min(nmax - nlocal - 1, send_cap);
    if (!launch_border_scan(chunk_count, chunk_offset, nchunk, d_dir_start, d_flags + 3 /* any value >= 0: no range check */, 0, cap_bound,
                            d_flags, h_flags_dev, side)) {
        HIPCHK(exclusive_scan_i32(scan_temp_side, scan_temp_side_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, side));
        launch_dir_starts_check(chunk_offset, nchunk, d_dir_start, d_flags + 3, 0, cap_bound, d_flags, side);
        HIPCHK(hipMemcpyAsync(h_flags + 16, d_dir_start, 28 * sizeof(int), hipMemcpyDeviceToHost, side));
        HIPCHK(hipMemcpyAsync(h_flags + 8, d_flags, sizeof(int), hipMemcpyDeviceToHost, side));
    }
    nsend = nghost = bound;            // launch bounds until resolve_counts() has the numbers
    launch_border_fill(cur, 0, end, slab_lo, slab_hi, nullptr, chunk_offset, nchunk, sendlist, side);
    // ghosts go behind the locals of the arrays the gather below fills (alt becomes cur)
    launch_pack_border(cur, sendlist, nsend, d_dir_start, shift27, alt.x[0] + nlocal, alt.x[1] + nlocal, alt.x[2] + nlocal,
                       alt.tag + nlocal, alt.type + nlocal, alt.mask + nlocal, side);
    const int *ng_dev = d_dir_start + 27;
    launch_ghost_count(alt, geom, nlocal, nghost, bin_key, bin_val, gcount, ng_dev, side);
    HIPCHK(exclusive_scan_i32(scan_temp_side, scan_temp_side_bytes, gcount, gstart, bargs.M + 1, side));
    launch_ghost_order(bin_key, bin_val, gstart, bargs.M, nghost, reorder_cap, gtmp_placed, bin_val_alt, gtmp_code, gslot, gcount, ng_dev, side);
    HIPCHK(hipEventRecord(ev_ghosts, side));
    // ---- main: scan, placement, ordering, gather
    HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, rcount, estart, ncodes + 1, stream));
    launch_reorder_place(rkey, rval_alt, estart, geom, ncodes, nlocal, reorder_cap, (int *)rkey_alt, rval, (uint32_t *)bin_key_alt, rcount, stream);
    std::swap(rkey, bin_key_alt);
    launch_permute_merge(cur, alt, rval, nlocal, permute_forces ? 1 : 0, coord4, veloc4, 0.5 * (subhi[0] + sublo[0]),
                         0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]), premix_tea<64>((u32)seed, (u32)ntimestep), perm_inverse,
                         images_on() ? img_cnt : nullptr, stream);
    merged_in_reorder = true;
    std::swap(cur, alt); ck_swapped = !ck_swapped;
    // ---- join
    HIPCHK(hipStreamWaitEvent(stream, ev_ghosts, 0));
    launch_translate_list(sendlist, perm_inverse, nsend, ng_dev, estart + bargs.M, h_flags_dev, stream);
    if (!ev_counts) HIPCHK(hipEventCreateWithFlags(&ev_counts, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_counts, stream));
    counts_by_seq = false;
    counts_pending = true;
    bulk_pending = true;
    ghosts_binned = true;
    tend("reorder");
    return 0;
}

// ------------------------------------------------------------------------------------------------
// one rank: the whole rebuild in front of the list builder in three launches (rebuild.hip)
// ------------------------------------------------------------------------------------------------
bool Engine::fused_ok() const
{
    return fused_rebuild && async_ok() && !overlap_rebuild && nlocal > 0 && (long)nlocal < (1L << 27) && neigh_kernel == 1;
}

int Engine::fused_alloc()
{
    const size_t M = (size_t)bargs.M;
    const int cap = fr_cap_user > 0 ? fr_cap_user : std::max(fr_cap_want, 16);
    if (M != fr_M || cap > fr_cap || (fr_cap_user > 0 && cap != fr_cap)) {
        dfree(fr_bucket);
        for (int k = 0; k < 2; k++) { dfree(fr_ttot[k]); dfree(fr_stot[k]); dfree(fr_gttot[k]); dfree(fr_gstot[k]); }
        fr_M = M; fr_cap = cap; fr_gcap = cap;
        const size_t ntl = 2 * M / fused_tile_codes(), nsl = ntl / fused_super_tiles() + 1;
        const size_t ntg = M / fused_gtile_codes(), nsg = ntg / fused_super_tiles() + 1;
        HIPCHK(dalloc(fr_bucket, 2 * M * (size_t)cap));
        for (int k = 0; k < 2; k++) {
            HIPCHK(dalloc(fr_ttot[k], ntl)); HIPCHK(dalloc(fr_stot[k], nsl));
            HIPCHK(dalloc(fr_gttot[k], ntg)); HIPCHK(dalloc(fr_gstot[k], nsg));      // (stot: [0] only, written afresh by k_fr_super)
        }
        fused_dirty = true;
    }
    if (!fr_ovf) {
        HIPCHK(dalloc(fr_ovf, 2 * (size_t)fr_ovf_cap));
        HIPCHK(dalloc(fr_novf, 1));
        fused_dirty = true;
    }
    if (fused_dirty) {
        const size_t ntl = 2 * M / fused_tile_codes(), nsl = ntl / fused_super_tiles() + 1;
        const size_t ntg = M / fused_gtile_codes(), nsg = ntg / fused_super_tiles() + 1;
        for (int k = 0; k < 2; k++) {
            HIPCHK(hipMemsetAsync(fr_ttot[k], 0, ntl * sizeof(int), stream)); HIPCHK(hipMemsetAsync(fr_stot[k], 0, nsl * sizeof(int), stream));
            HIPCHK(hipMemsetAsync(fr_gttot[k], 0, ntg * sizeof(int), stream)); HIPCHK(hipMemsetAsync(fr_gstot[k], 0, nsg * sizeof(int), stream));
        }
        HIPCHK(hipMemsetAsync(rcount, 0, (2 * M + 1) * sizeof(int), stream));
        HIPCHK(hipMemsetAsync(fr_novf, 0, sizeof(int), stream));
        fused_dirty = false;
    }
    return 0;
}

// arguments of the count + place kernels (the locals' half of the fused rebuild)
void Engine::fused_locals_args(FusedArgs &a, bool ghost_stage)
{
    memset(&a, 0, sizeof a);
    a.src = cur; a.dst = alt;
    a.n = mig_holes ? mig_span : nlocal;
    a.skip = mig_holes ? gslot : nullptr; a.skip_n = mig_nold;      // (gslot: the migration's code array, see migrate)
    a.with_f = permute_forces ? 1 : 0;
    a.wrap = wrap_in_reorder ? 1 : 0;
    for (int d = 0; d < 3; d++) {
        a.boxlo[d] = boxlo[d]; a.boxhi[d] = boxhi[d]; a.per[d] = periodic[d];
        a.sl.lo[d] = slab_lo[d]; a.sl.hi[d] = slab_hi[d];
    }
    a.g = geom;
    a.sub_bits = reorder_sub_bits(geom);
    a.M = bargs.M;
    a.cnt = rcount; a.bucket = fr_bucket; a.cap = fr_cap;
    a.ovf = fr_ovf; a.novf = fr_novf; a.ovf_cap = fr_ovf_cap;
    const int par = (int)(fr_epoch & 1u);
    a.ttot = fr_ttot[par]; a.ttot_next = fr_ttot[par ^ 1];
    a.stot = 2 * bargs.M / fused_tile_codes() > fused_direct_tiles() ? fr_stot[0] : nullptr;
    a.estart = estart;
    if (ghost_stage) {
        // one rank: the count books every border atom's periodic images per tile of ghost cells (k_fr_ghosts then needs nothing
        // from the gather: the two run as one launch)
        a.gttot = fr_gttot[par];
        a.dir_mask = 0;
        for (int d = 0; d < 27; d++)
            if (d != 13 && send_active[d]) a.dir_mask |= 1u << d;
        a.img_booked = 1;
    }
    a.perm = rval;
    // large boxes: order first, then a streaming gather (64^3: the fused form moves its 190 MB at 2.4 TB/s, a streaming pass at 5)
    a.split_gather = (nranks == 1 && (split_gather == 1 || (split_gather < 0 && nlocal >= 50000))) ? 1 : 0;
    a.scratch = fr_scratch;
    a.lds_cap = reorder_cap;
    a.mg.coord4 = coord4; a.mg.veloc4 = veloc4;
    a.mg.cx = 0.5 * (subhi[0] + sublo[0]); a.mg.cy = 0.5 * (subhi[1] + sublo[1]); a.mg.cz = 0.5 * (subhi[2] + sublo[2]);
    a.mg.seed = premix_tea<64>((u32)seed, (u32)ntimestep);
    a.mg.inverse = nullptr;
    a.mg.zero = (images_on() || mr_img_wanted()) ? img_cnt : nullptr;
    if (a.mg.zero) img_zero_gen = img_alloc_gen;      // (only a gather that clears the counters vouches for them)
    a.flags = d_flags;
}

// several ranks: the rebuild's border kernel records every border atom's slots in the per-step refresh messages, and the force
// kernel's step boundary writes the refresh (sub-boxes at least two ghost cutoffs wide: at most 7 directions per atom)
bool Engine::mr_img_wanted() const
{
    if (!(refresh_epilogue && nranks > 1 && border_fused && mr_async_ok() && fuse_pair && fuse_step && ring_selected() && reorder_fuses(nlocal))) return false;
    for (int d = 0; d < 3; d++)
        if (!(subhi[d] - sublo[d] > 2.0 * cutghost)) return false;
    return true;
}

int Engine::rebuild_fused()
{
    tbegin("reorder");
    // the ghost count of the previous rebuild (+ head-room) sizes the arrays and the later per-step launches; the ghost kernel
    // itself covers every ghost cell whatever the count is, and reports a count beyond the capacity
    const int bound = (int)(nghost_prev * async_grid_scale) + 1024;
    TRY(ensure_capacity(nlocal + bound + bound / 2));
    TRY(fused_alloc());
    if (!h_flags_dev) HIPCHK(hipHostGetDevicePointer((void **)&h_flags_dev, h_flags, 0));
    FusedArgs a;
    fused_locals_args(a, true);
    const int par = (int)(fr_epoch & 1u);
    a.gttot = fr_gttot[par]; a.gttot_next = fr_gttot[par ^ 1];
    // (the separate plan kernel - option tile_plan, and rows narrower than 64 entries, whose stage cannot lend the inline plan its
    // run tables - reads gstart by differences: every tile runs then.  The same predicate as plan_inline in build_cells_and_table.)
    const bool skip_tiles = tile_plan == 0 && n_col >= 64 && fr_ngorder > 0;
    a.gorder = skip_tiles ? fr_gorder : nullptr; a.ngorder = fr_ngorder; a.gcnt_out = skip_tiles ? fr_gcnt : nullptr;
    // few tiles run: each adds up the tile totals in front of it directly (no supertile launch) up to 16 Ki tiles
    a.gstot = bargs.M / fused_gtile_codes() > (skip_tiles ? 16384 : fused_direct_tiles()) ? fr_gstot[0] : nullptr;
    fused_gcnt_valid = skip_tiles;
    fr_epoch++;
    a.gstart = gstart;
    a.dir_mask = 0;
    for (int d = 0; d < 27; d++)
        if (d != 13 && send_active[d]) a.dir_mask |= 1u << d;
    for (int d = 0; d < 27; d++)
        for (int k = 0; k < 3; k++) { a.sh.s[d][k] = shift27[3 * d + k]; a.ce.c[d][k] = center27[3 * d + k]; }
    a.sendlist = sendlist; a.senddir = senddir;
    a.img_cnt = images_on() ? img_cnt : nullptr; a.img = img;
    a.ghost_cap = std::min(nmax - nlocal - 1, send_cap);
    if (debug_ghost_cap > 0) { a.ghost_cap = std::min(a.ghost_cap, debug_ghost_cap); debug_ghost_cap = 0; }
    a.dir_start = d_dir_start;
    a.report = h_flags_dev;
    a.merged_ghosts = merge_ghosts ? 1 : 0;      // (honoured with the order-only placing kernel: launch_fused_rebuild)
    // the report names this rebuild: the host polls the word in pinned memory when it next needs the counts (resolve_counts)
    counts_by_seq = report_poll != 0 && a.gttot != nullptr;      // (the ghost tiles write the report)
    if (counts_by_seq) { report_seq = report_seq % 1000000 + 1; a.report_seq = report_seq; }
    launch_fused_rebuild(a, stream, count_in_epilogue);
    count_in_epilogue = false;
    std::swap(cur, alt); ck_swapped = !ck_swapped;
    merged_in_reorder = true;
    if (!counts_by_seq) {
        if (!ev_counts) HIPCHK(hipEventCreateWithFlags(&ev_counts, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_counts, stream));
    }
    counts_pending = true;
    bulk_pending = true;
    ghosts_binned = true;
    fused_active = true;
    nsend = nghost = bound;            // launch bounds until resolve_counts() has the numbers
    tend("reorder");
    return 0;
}

// MesoComm::borders (comm_meso.cu:41-186) as device list building + device pack
int Engine::halo_borders()
{
    if (nranks > 1) return halo_borders_multi();
    tbegin("halo");
    if (async_ok() && bulk_pending) {
        // No host round trip: the ghost count of the previous rebuild (+ 12.5 % + 1024) bounds this one's launches, every consumer
        // masks with the device-side count (d_dir_start[27]), and the counts travel to the host behind an event that
        // resolve_counts() waits for when the host next needs them (the next rebuild, the end of run(), any query).
        // `bound` sizes the grids of the ghost kernels (they loop, so any count is covered); the hard limit is the capacity
        int bound = (int)(nghost_prev * async_grid_scale) + 1024;
        TRY(ensure_capacity(nlocal + bound + bound / 2));
        int cap_bound = std::min(nmax - nlocal - 1, send_cap);
        if (debug_ghost_cap > 0) { cap_bound = std::min(cap_bound, debug_ghost_cap); debug_ghost_cap = 0; }
        // border scan: every local atom in small boxes; in large ones from a little before the previous border section
        int beg = 0;
        if (nlocal > 524288) beg = std::max(0, n_bulk_prev - n_bulk_prev / 64 - 4096) & ~255;
        const int end = nlocal, nchunk = (end - beg + 255) / 256;
        launch_border_count(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, stream);
        const int *nb_dev = estart + bargs.M;       // n_bulk where the reorder's scan left it
        if (!h_flags_dev) HIPCHK(hipHostGetDevicePointer((void **)&h_flags_dev, h_flags, 0));
        if (!launch_border_scan(chunk_count, chunk_offset, nchunk, d_dir_start, nb_dev, beg, cap_bound, d_flags, h_flags_dev, stream)) {
            HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
            launch_dir_starts_check(chunk_offset, nchunk, d_dir_start, nb_dev, beg, cap_bound, d_flags, stream);
            HIPCHK(hipMemcpyAsync(h_flags + 16, d_dir_start, 28 * sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_flags + 8, d_flags, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_flags + 9, nb_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
        }
        if (!ev_counts) HIPCHK(hipEventCreateWithFlags(&ev_counts, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_counts, stream));
        counts_by_seq = false;
        counts_pending = true;
        nsend = nghost = bound;            // launch bounds until resolve_counts() has the numbers
        launch_border_fill(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_offset, nchunk, sendlist, stream);
        launch_pack_border(cur, sendlist, nsend, d_dir_start, shift27, cur.x[0] + nlocal, cur.x[1] + nlocal, cur.x[2] + nlocal,
                           cur.tag + nlocal, cur.type + nlocal, cur.mask + nlocal, stream);
        tend("halo");
        return 0;
    }
    int beg = bulk_pending ? 0 : n_bulk, end = nlocal;
    int nchunk = (end - beg + 255) / 256;
    nsend = 0;
    for (int k = 0; k < 28; k++) h_dir_start[k] = 0;
    if (nchunk > 0) {
        launch_border_count(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, stream);
        HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
        // direction starts = offsets of chunk 0 of each direction (+ total)
        launch_dir_starts(chunk_offset, nchunk, d_dir_start, stream);
        HIPCHK(hipMemcpyAsync(h_flags + 16, d_dir_start, 28 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int k = 0; k < 28; k++) h_dir_start[k] = h_flags[16 + k];
        nsend = h_dir_start[27];
    } else {
        HIPCHK(hipMemsetAsync(d_dir_start, 0, 28 * sizeof(int), stream));
        if (bulk_pending) HIPCHK(hipStreamSynchronize(stream));
    }
    if (bulk_pending) {
        bulk_pending = false;
        if (h_flags[0]) return check_overflow();
        n_bulk = h_flags[1];
    }
    // single rank: every send is my own ghost
    nghost = nsend;
    nghost_prev = nghost; n_bulk_prev = n_bulk;
    if (nlocal + nghost > nmax || nsend > send_cap) {
        TRY(ensure_capacity(nlocal + nghost));
        // lists are rebuilt below on the new buffers; chunk arrays were regrown, so recount
        launch_border_count(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, stream);
        HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
    }
    if (nsend > 0) {
        launch_border_fill(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_offset, nchunk, sendlist, stream);
        launch_pack_border(cur, sendlist, nsend, d_dir_start, shift27, cur.x[0] + nlocal, cur.x[1] + nlocal,
                           cur.x[2] + nlocal, cur.tag + nlocal, cur.type + nlocal, cur.mask + nlocal, stream);
    }
    tend("halo");
    return 0;
}

// Comm::forward_comm + gpu_merge_xvt(ghost range): ghosts arrive as merged float4 pairs
int Engine::halo_forward_seed(uint32_t sd, bool async)
{
    if (nranks > 1) return halo_forward_multi_begin(sd, async);
    if (nsend <= 0) return 0;
    tbegin("halo");
    // at a rebuild (merged_in_reorder was just consumed: the counters were cleared by the gather) the kernel also records every
    // source atom's images for the step-boundary epilogue
    const bool rec = build_images_now && images_on();
    launch_pack_forward(cur, sendlist, nsend, d_dir_start, shift27, center27, sd, coord4 + nlocal, veloc4 + nlocal,
                        fused_active ? nullptr : gslot, rec ? img_cnt : nullptr, img, nlocal, fused_active ? senddir : nullptr, stream);
    if (rec) images_ready = true;
    tend("halo");
    return 0;
}

int Engine::merge_locals(uint32_t sd)
{
    tbegin("merge");
    launch_merge_xvt(cur, coord4, veloc4, 0.5 * (subhi[0] + sublo[0]), 0.5 * (subhi[1] + sublo[1]),
                     0.5 * (subhi[2] + sublo[2]), sd, 0, nlocal, stream);
    tend("merge");
    return 0;
}

// binning_meso (neighbor_meso.cu:535-711) + full_bin_meso (neigh_build_meso.cu:254-417)
int Engine::build_cells_and_table()
{
    float rc2 = (float)((cutmax + skin) * (cutmax + skin));
    {
        // locals are already cell-ordered by the reorder; only the ghosts need binning
        tbegin("bin");
        // (the lane-per-atom builder and the sorting variants want the counts on the host)
        if (counts_pending && mr_pending && !(neigh_kernel == 1 && n_col <= tile_build_rowcap() && tile_fits && !ghost_sort)) TRY(resolve_counts());
        bargs.active = nullptr;          // every brick, in workgroup-id order (brick_slot)
        bargs.nactive = bargs.nbricks;
        bargs.nactive_dev = nullptr;
        if (ghosts_binned) {
            // (rebuild_overlapped binned this rebuild's ghosts on the side stream)
        } else if (ghost_sort) {
            launch_ghost_morton(cur, geom, nlocal, nghost, bin_key, bin_val, stream);
            if (nghost > 0)
                HIPCHK(sort_pairs_u32(sort_temp, sort_temp_bytes, bin_key, bin_key_alt, bin_val, bin_val_alt, nghost,
                                      std::max(1, 3 * l1bits), stream));
            launch_code_starts_u32(bin_key, nghost, bargs.M, gstart, stream);
            launch_invert_perm(bin_val, gslot, nghost, stream);
        } else {
            // counting instead of sorting: 6 launches instead of ~18 (the ghosts' comparison sort was launch-bound)
            const int *ng_dev = pending_nghost_dev();     // nghost is a launch bound while the counts travel
            launch_ghost_count(cur, geom, nlocal, nghost, bin_key, bin_val, gcount, ng_dev, stream);
            HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, gcount, gstart, bargs.M + 1, stream));
            launch_ghost_order(bin_key, bin_val, gstart, bargs.M, nghost, reorder_cap, (int *)bin_key_alt, bin_val_alt, (uint32_t *)rkey_alt, gslot,
                               gcount, ng_dev, stream);
        }
        bargs.ghost_base = nlocal;
        tend("bin");
        // merged arrays with the signatures of the CURRENT step: the force kernel of this step uses them as they are
        const u32 sd_now = premix_tea<64>((u32)seed, (u32)ntimestep);
        images_ready = false;
        build_images_now = merged_in_reorder;      // the gather of this rebuild cleared the image counters
        if (!merged_in_reorder) TRY(merge_locals(sd_now));
        merged_in_reorder = false;
        {
            // (several ranks: the border message carried the velocities, the ghosts' merged pairs are built locally)
            // (fused rebuild: the ghosts' merged pairs and the image table were written with the ghosts)
            const int rc_fwd = (fused_active || mr_runs) ? 0 : (nranks > 1 ? merge_new_ghosts(sd_now) : halo_forward_seed(sd_now));
            if (fused_active && images_on()) images_ready = true;
            build_images_now = false;
            if (rc_fwd) return rc_fwd;
        }
        TRY(rebuild_topology());
        {
            tbegin("neigh");
            ExclArgs ex = {nullptr, nullptr, nullptr, 0};
            if (have_bonds && msp > 0) { ex.tagc = tagc; ex.nspecial = cur.nspecial; ex.special = cur.special; ex.msp = msp; }
            if (neigh_kernel == 1 && n_col <= tile_build_rowcap() && tile_fits) {
                // wave-per-bin ballot builder on LDS-staged neighbourhoods (every brick: empty ones exit at once)
                bargs.hoff = brick_hoff; bargs.hmap = brick_hmap; bargs.hdr = brick_hdr; bargs.own_info = brick_own;
                // (the inline plan borrows the row stage for its run tables: 5 x 216 ints)
                bargs.plan_inline = (tile_plan == 0 && bargs.active == nullptr && n_col >= 64) ? 1 : 0;
                if (!bargs.plan_inline) launch_brick_plan(bargs, d_flags, stream);
                BrickArgs bb = bargs;
                bb.gcnt = (fused_active && fused_gcnt_valid) ? fr_gcnt : (mr_runs ? mr_gcnt : nullptr);
                if (brick2_off || !brick2) bb.maxh2 = 0;
                bb.brick2_limit = brick2_limit;
                if (tile_persist && !tile_queue) {
                    HIPCHK(dalloc(tile_queue, (size_t)tile_build_queue_ints()));
                    HIPCHK(hipMemsetAsync(tile_queue, 0, (size_t)tile_build_queue_ints() * sizeof(int), stream));
                }
                bb.queue = tile_persist ? tile_queue : nullptr;
                // partitioned rows (RowPartArgs): the pairing group is the ring kernel's workgroup for a launch over this rank's atoms
                // (every force launch of the interval is then made with the same lanes per atom, launch_pair)
                // (not for the wide records of more than 2^25 atoms on a rank, whose launches walk one row per atom)
                // option row_part -1 (default): the sections cost the builder ~12 % and save every force launch of the interval 6 % (fp32
                // style) or 10 % (fp64 style): they pay from a rebuild every 4 (2) steps on; with check yes the interval is long
                const bool part_pays = row_part > 0 || (row_part < 0 && (dist_check || every >= (pair_style == 0 ? 2 : 4)));
                rows_part = part_pays && pair_share != 0 && ring_selected() && pair_debug != 9 &&
                            (counts_pending ? (long)nmax : (long)nlocal + nghost) <= (1L << 25);
                part_group = rows_part ? pair_ring_group_for(nlocal, pair_npart) : 0;
                // the back table (half a row per atom) exists only for decks that partition: plain rows, the wide records of very
                // large systems and the lane-per-atom builder never pay for it (21 GB at 256^3)
                if (rows_part && !pair_back) HIPCHK(dalloc(pair_back, table_tiles * 64 * (size_t)nb_col));
                RowPartArgs pt = {part_group, nlocal, pair_nback, pair_back, nb_col};
                launch_tile_build(bb, coord4, rc2, n_col, pair_count, pair_table, d_flags, have_bonds ? &ex : nullptr, nlocal,
                                  pair_debug >= 10 ? pair_debug - 10 : 0, stream, &pt);
                tend("neigh");
                nbuild++;
                return 0;
            }
            rows_part = false; part_group = 0;
            launch_bin_ranges(estart, gstart, bargs.M, nlocal, binrange, stream);
            launch_cell_build(coord4, rkey, reorder_sub_bits(geom), binrange, bargs.M, geom.mbin, rc2, nlocal, n_col, pair_count, pair_table,
                              d_flags, have_bonds ? &ex : nullptr, stream);
            tend("neigh");
        }
        nbuild++;
        return 0;
    }
}

bool Engine::async_ok() const
{
    return async_counts && nranks == 1 && !ghost_sort && !reorder_sort && nghost_prev >= 0 &&
           neigh_kernel == 1 && tile_fits;
}

// the host's copy of the counts a rebuild left on the device (see halo_borders)
int Engine::resolve_counts()
{
    if (!counts_pending) { counts_by_seq = false; return 0; }
    if (counts_by_seq) {
        // the ghost tiles wrote the counts and then the rebuild's number (system-scope release): long since there when the host asks -
        // four steps later, as a rule; a stream synchronisation only if it does not show up
        volatile int *seq = (volatile int *)h_flags + 12;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
        while (*seq != report_seq && std::chrono::steady_clock::now() < t_end) std::this_thread::yield();
        if (*seq != report_seq) {
            HIPCHK(hipStreamSynchronize(stream));
            if (*seq != report_seq) return fail(2, "The rebuild's report did not arrive");
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        counts_by_seq = false;
    } else HIPCHK(hipEventSynchronize(ev_counts));
    counts_pending = false;
    bulk_pending = false;
    if (h_flags[8]) {
        mr_pending = false;
        // One rank, inside run(): an outgrown ghost list / border range / cell bucket poisoned every launch behind the rebuild that
        // reported it (PairArgs::poison, the NVE kernels): the state is still what that rebuild was given.  run() goes back to it.
        const int code = h_flags[8];
        if (nranks == 1 && redo_armed && (code == 200000 || code == 200001 || code >= 300000)) return prepare_redo(code);
        return check_overflow();
    }
    n_bulk = h_flags[9];
    for (int k = 0; k < 28; k++) h_dir_start[k] = (fused_active && k < 27) ? 0 : h_flags[16 + k];      // (fused rebuild: no direction segments)
    if (mr_pending) return mr_resolve();
    nsend = nghost = h_dir_start[27];
    nghost_prev = nghost; n_bulk_prev = n_bulk;
    return 0;
}

// The rebuild at checkpoint `ck` reported an outgrown capacity (resolve_counts).  Every launch behind it was a no-op; the atoms it
// reordered from are still in `alt` (the reorder gathers, it does not move).  Back to them, and the rebuild runs again through the
// synchronous path (exact counts, capacities regrown) - MesoComm::borders grows its buffers on the fly too (comm_meso.cu:122,138,179-181).
int Engine::prepare_redo(int code)
{
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
    h_flags[0] = h_flags[8] = 0;
    counts_pending = bulk_pending = false;
    count_in_epilogue = false;
    nghost_prev = -1;            // the redone rebuild (and only it) takes the synchronous path
    fused_dirty = true;
    if (code >= 300000) {
        // a cell bucket and the overflow list together were too small: twice the bucket from here on
        fr_cap_user = 0;
        fr_cap_want = std::max(fr_cap_want, fr_cap) * 2;
    }
    if (ck_swapped) { std::swap(cur, alt); ck_swapped = false; }
    nlocal = ck_nlocal;
    nredo++;
    return MESO_REDO;
}

int Engine::check_overflow()
{
    HIPCHK(hipMemcpyAsync(h_flags, d_flags, 8 * sizeof(int), hipMemcpyDeviceToHost, stream));      // [5]: fullest brick neighbourhood so far, [6]: 2-brick near its stage, [7]: fp32 force sums out of range
    HIPCHK(hipStreamSynchronize(stream));
    if (h_flags[7]) {
        h_flags[7] = 0;
        HIPCHK(hipMemsetAsync(d_flags + 7, 0, sizeof(int), stream));
        return fail(4, "Force on an atom beyond the range of the fp32 styles' 32-bit fixed-point sums (|F| >= 16384 force units: overlapping "
                       "atoms, or a deck far from reduced units): the forces of this run are not valid - use pair_style dpd/meso (64-bit sums)");
    }
    if (have_bonds && h_flags[4]) {
        HIPCHK(hipMemsetAsync(d_flags + 4, 0, sizeof(int), stream));
        return fail(4, "Bond atoms missing: a bonded (or angle) partner is outside the ghost cutoff");
    }
    if (h_flags[0]) {
        char buf[200];
        if (h_flags[0] == 200002) {
            HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
            counts_pending = mr_pending = false;
            mr_caps_ready = false;
            return fail(4, "A border message outgrew the capacity both ranks derived from the previous rebuild (ghost count up by more than a "
                           "quarter within one rebuild interval): run again with option async_counts 0");
        }
        if (h_flags[0] == 200000 || h_flags[0] == 200001) {
            HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
            counts_pending = false;
            nghost_prev = -1;      // the next rebuild takes the synchronous path again
            fused_dirty = true;
            return fail(4, h_flags[0] == 200000 ? "Ghost list outgrew the capacity reserved from the previous rebuild (the ghost count rose by more "
                                                  "than two thirds within one rebuild interval): run again with option async_counts 0"
                                                : "Border section moved in front of the scanned range: run again with option async_counts 0");
        }
        if (h_flags[0] >= 300000 && nranks == 1 && in_reneighbor) {
            // seen by the rebuild itself (synchronous path): buckets twice as deep, and reneighbor() builds again from the arrays the
            // reorder gathered from
            HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
            h_flags[0] = 0;
            counts_pending = bulk_pending = false;
            nghost_prev = -1;
            fused_dirty = true;
            fr_cap_user = 0;
            fr_cap_want = std::max(fr_cap_want, fr_cap) * 2;
            return MESO_DEEPER;
        }
        if (h_flags[0] >= 300000) {
            HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
            counts_pending = false;
            nghost_prev = -1;
            fused_dirty = true;
            return fail(4, "Fused rebuild: a cell holds more atoms than its bucket and the overflow list together can take (local density far "
                           "above the mean): run again with option fused_rebuild 0");
        }
        if (h_flags[0] >= 100000)
            snprintf(buf, sizeof buf, "Brick halo overflow: %d atoms in one brick neighbourhood (capacity %d); local density too "
                     "high - raise option brick_margin or use neigh_kernel 0", h_flags[0] - 100000, bargs.maxh);
        else if (h_flags[0] == n_col + 1 && rows_part)
            snprintf(buf, sizeof buf, "Pair table overflow: a row's back section (mirrored in-group partners) holds more than nb_col = %d entries "
                     "or the row more than n_col = %d; local density too high - option row_part 0 stores plain rows", nb_col, n_col);
        else
            snprintf(buf, sizeof buf, "Pair table overflow: %d > %d; local density too high", h_flags[0], n_col);
        return fail(4, buf);
    }
    return 0;
}

// option check_launches (debugging): every stage of a rebuild is waited for and asked for launch / execution errors, so that a fault
// is reported with the stage that raised it instead of at the end of run()
int Engine::launch_check(const char *stage)
{
    if (!check_launches) return 0;
    hipError_t e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) {
        char msg[256];
        snprintf(msg, sizeof msg, "HIP error after the rebuild stage '%s' (timestep %ld): %s", stage, (long)ntimestep, hipGetErrorString(e));
        return fail(2, msg);
    }
    return 0;
}

int Engine::reneighbor()
{
    TRY(resolve_counts());       // the previous rebuild's counts (long since arrived) size this one
    int rc = 0;
    for (int attempt = 0; attempt < 8; attempt++) {
        in_reneighbor = true;
        rc = reneighbor_once();
        in_reneighbor = false;
        if (rc != MESO_DEEPER) break;
        if (ck_swapped) { std::swap(cur, alt); ck_swapped = false; }      // (the reorder gathered from intact arrays)
    }
    return rc == MESO_DEEPER ? fail(4, "Fused rebuild: a cell holds more atoms than eight doublings of its bucket can take") : rc;
}

int Engine::reneighbor_once()
{
    TRY(resolve_counts());
    ck_swapped = false; ck_nlocal = nlocal;      // (what prepare_redo needs to undo THIS rebuild, should its report ask for it)
    {
        // denser than expected (chains, phase separation): the LDS stage of a brick neighbourhood grows BEFORE it overflows - the
        // high-water mark of the earlier list builds came with the count report (or with the last check_overflow)
        if ((h_flags[6] || h_flags[11]) && !brick2_off) {
            // a 2-brick neighbourhood passed three quarters of its LDS stage: the stage grows by a quarter (eight workgroups per CU
            // up to 20 KB, fewer beyond: still ahead of the 4-brick, 176 against 273 us per build at 64^3) as long as five
            // workgroups fit a CU; only then the 4-brick takes over, whose stage keeps growing on its own
            const int next = (bargs.maxh2 * 5 / 4 + 63) / 64 * 64;
            if ((size_t)next * 16 + (size_t)4 * 4 * n_col * 2 + 2048 <= 32 * 1024) {
                brick2_floor = next;
                params_ready = false;
                regrow_only = true;
            } else brick2_off = true;
            h_flags[6] = h_flags[11] = 0;
            HIPCHK(hipMemsetAsync(d_flags + 6, 0, sizeof(int), stream));
        }
        const int hwm = std::max(h_flags[5], h_flags[10]);
        if (params_ready && neigh_kernel == 1 && (long)hwm * 100 > (long)bargs.maxh * 93) {
            brick_maxh_floor = ((int)(hwm * 1.08) + 63) / 64 * 64;
            params_ready = false;
            regrow_only = true;
            h_flags[5] = h_flags[10] = 0;
            HIPCHK(hipMemsetAsync(d_flags + 5, 0, sizeof(int), stream));
        }
    }
    TRY(init_params());
    // one rank: nothing happens between the wrap and the reorder, which reads the coordinates anyway - wrapped there
    wrap_in_reorder = nranks == 1 && !reorder_sort && nlocal > 0;
    if (!wrap_in_reorder && !mig_slim_now()) launch_pbc(cur, boxlo, boxhi, periodic, nlocal, stream);      // (the slim migration front wraps)
    TRY(migrate());
    TRY(launch_check("migrate"));
    ghosts_binned = false;
    fused_active = false;
    mr_runs = false;
    mr_images_ready = false;
    fwd_packed = false;
    if (count_in_epilogue && !fused_ok()) {
        // the force kernel's epilogue counted for a fused rebuild that does not happen: the counters are cleared before their next use
        count_in_epilogue = false;
        fused_dirty = true;
        if (rcount) HIPCHK(hipMemsetAsync(rcount, 0, (2 * (size_t)bargs.M + 1) * sizeof(int), stream));
    }
    if (fused_ok()) {
        TRY(rebuild_fused());
        TRY(launch_check("fused rebuild (count, place, gather, ghosts)"));
    } else if (async_ok() && overlap_rebuild && nlocal > 0) {
        TRY(rebuild_overlapped());
        TRY(launch_check("overlapped rebuild"));
    } else {
        TRY(reorder_locals());
        TRY(launch_check("reorder"));
        TRY(halo_borders());
        TRY(launch_check("borders"));
    }
    // the gathers of the force kernel address the merged arrays through 32-bit byte offsets (16 bytes per atom)
    if ((long)nlocal + nghost >= (1L << 28))
        return fail(4, "Too many atoms on one rank: local + ghost atoms exceed 268435456 (32-bit byte offsets of the gathers); use more ranks");
    TRY(build_cells_and_table());
    TRY(launch_check("ghost binning, merged arrays, list builder"));
    if (dist_check) launch_copy_hold(cur, xhold, nlocal, nmax, stream);
    ago = 0;
    return 0;
}

// Neighbor::decide (src/neighbor.cpp:1216-1231) with check_distance on the device
int Engine::decide(int *rebuild)
{
    ago++;
    *rebuild = 0;
    if (ago >= delay && ago % every == 0) {
        if (!dist_check) { *rebuild = 1; return 0; }
        launch_max_disp2(cur, xhold, nlocal, nmax, d_partial, d_scalar, stream);
        HIPCHK(hipMemcpyAsync(h_scalar, d_scalar, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        double trig = 0.5 * skin;
        *rebuild = reduce_global_sum(h_scalar[0] > trig * trig ? 1.0 : 0.0) > 0.0 ? 1 : 0;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// per-step pieces
// ------------------------------------------------------------------------------------------------
int Engine::nve_initial()
{
    tbegin("nve");
    launch_nve_initial(cur, 0.5 * dt, dt, groupbit, nlocal, stream, d_flags);
    tend("nve");
    return 0;
}

int Engine::nve_final()
{
    tbegin("nve");
    launch_nve_final(cur, 0.5 * dt, groupbit, nlocal, stream, d_flags);
    tend("nve");
    return 0;
}

int Engine::halo_forward() { return halo_forward_seed(premix_tea<64>((u32)seed, (u32)ntimestep)); }

int Engine::force_clear(int r)
{
    int beg, end;
    range(r, beg, end);
    for (int d = 0; d < 3; d++) launch_fill_f64(cur.f[d] + beg, 0.0, end - beg, stream);
    return 0;
}

// MesoPairDPD::compute / compute_bulk / compute_border (pair_dpd_meso.cu:241-266).  The merged arrays of the
// local range are refreshed for LOCAL and BULK calls, the ghost range for LOCAL and BORDER calls, exactly the
// split the reference uses to hide its host round trip.
// kernel choice (option pair_kernel): 0 lane per atom, 1 tile, 2 auto, 3 mlpc, 4 mlp, 5 ring
bool Engine::ring_selected() const { return pair_kernel == 5 || pair_kernel == 2; }

void Engine::launch_pair(PairArgs &p, int ev)
{
    const bool cell_ring = !ev && ring_selected();
    if (!cell_ring) p.fuse_nve = 0;              // only the ring kernel has the epilogue
    // bound of the buffer-addressed gathers (the ghost count may still be an estimate: then the capacity); 32-bit byte offsets,
    // 16 bytes per atom: never beyond 2^28 - 1 atoms (reneighbor refuses more atoms than that on a rank)
    p.nall = (int)std::min<long>(counts_pending ? (long)nmax : (long)nlocal + nghost, (1L << 28) - 1);
    p.rng = pair_rng;
    p.npart = pair_npart;
    p.bulk_hint = xcd_balance ? std::max(n_bulk_prev, 0) : 0;      // (the previous rebuild's bulk count: a scheduling hint)
    p.nback = rows_part ? pair_nback : nullptr;
    p.table_back = pair_back; p.nb_col = nb_col;
    p.part_group = part_group;
    p.range_flag = d_flags + 7;
    p.poison = d_flags;
    if (rows_part) p.npart = 4 * 64 / part_group;      // every launch of the interval pairs inside the groups the rows were partitioned for
    p.poly = pair_poly ? d_poly : nullptr;
    p.ftab = pair_ftab ? d_ftab : nullptr;
    p.ftab_len = ftab_len;
    p.all_expw_one = 1;
    p.share = (pair_share && (p.end == nlocal || (p.end & (pair_ring_group() - 1)) == 0)) ? 1 : 0;
    for (int t = 0; t < ntypes * ntypes; t++) p.all_expw_one &= coeff[(size_t)t * 7 + 3] == 1.0 ? 1 : 0;
    p.uniform_cut = 1;
    for (int t = 0; t < ntypes * ntypes; t++) p.uniform_cut &= coeff[(size_t)t * 7 + P_CUT] == coeff[P_CUT] ? 1 : 0;
    // two kernels: the ring kernel (both styles) and the lane-per-atom kernel that also books energy and virial
    if (ev || pair_kernel == 0) launch_pair_dpd(p, pair_style, ev, stream);
    else if (!launch_pair_dpd_ring(p, pair_style, stream, pair_variant)) launch_refused = true;
}

int Engine::pair_compute(int r, int eflag, int vflag)
{
    if (!is_setup && !params_ready) return fail(3, "pair_compute before setup");
    u32 sd = premix_tea<64>((u32)seed, (u32)ntimestep);
    if (r == 0 || r == 1) TRY(merge_locals(sd));
    if (r == 0 || r == 2) TRY(halo_forward_seed(sd));
    int beg, end;
    range(r, beg, end);
    PairArgs p = {};
    p.bond.nbond = nullptr;
    p.coord4 = coord4; p.veloc4 = veloc4; p.count = pair_count; p.table = pair_table; p.n_col = n_col;
    for (int d = 0; d < 3; d++) p.f[d] = cur.f[d];
    int ev = (eflag || vflag) ? 1 : 0;
    // ev_setup: the per-atom virial of the atoms in this range starts from zero (the kernel accumulates, like the forces)
    if (ev) for (int k = 0; k < 6; k++) launch_fill_f64(virial[k] + beg, 0.0, end - beg, stream);
    p.e_pair = ev ? e_pair : nullptr;
    for (int k = 0; k < 6; k++) p.virial[k] = ev ? virial[k] : nullptr;
    p.coeff64 = d_coeff64; p.coeff32 = d_coeff32; p.ntypes = ntypes;
    for (int k = 0; k < 7; k++) p.cf1[k] = coeff[k];
    p.dt_inv_sqrt = 1.0 / std::sqrt(dt);
    p.beg = beg; p.end = end;
    p.accumulate = 1;
    p.fuse_nve = 0;
    p.debug = 0;
    p.chunked = 1;
    tbegin("pair");
    launch_pair(p, ev);
    tend("pair");
    if (launch_refused) { launch_refused = false; return fail(2, "The force kernel has no form for this combination of row layout and record format"); }
    if (ev) ev_valid = true;
    return 0;
}

int Engine::setup()
{
    if (nlocal <= 0 && nranks == 1) return fail(3, "No atoms have been uploaded");
    if ((pair_rng || pair_poly || pair_ftab) && !ring_selected())
        return fail(3, "pair styles dpd/mini/meso, dpd/polyforce/meso and dpd/tableforce/meso run on the default force kernel only (pair_kernel=2)");
    TRY(init_params());
    for (int attempt = 0;; attempt++) {
        TRY(reneighbor());
        // the first list build sizes the brick stage from the mean density: a start configuration denser than that somewhere
        // (polymer decks) gets a larger stage and a second build instead of an error
        HIPCHK(hipMemcpyAsync(h_flags, d_flags, 6 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        const bool grow = h_flags[0] >= 100000 && h_flags[0] < 200000 && attempt < 8;
        // (the rebuild exchanges atoms and ghosts: every rank repeats it when one has to)
        if (reduce_global_sum(grow ? 1.0 : 0.0) == 0.0) break;
        if (grow) {
            brick_maxh_floor = ((int)((h_flags[0] - 100000) * 1.08) + 63) / 64 * 64;
            params_ready = false;
            regrow_only = true;
            HIPCHK(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
        }
    }
    nbuild = 0;
    if (restart_forces) {
        // continuing from a restart file: the forces of the interrupted step came with the atoms (restart.hip) and were
        // carried through the reorder above; energies and virial are tallied on demand (tally_ev)
        restart_forces = false;
        ev_valid = false;
        TRY(check_overflow());
        is_setup = true;
        return 0;
    }
    TRY(force_clear(0));
    for (int k = 0; k < 6; k++) launch_fill_f64(virial[k], 0.0, nlocal, stream);
    TRY(pair_compute(0, 1, 1));
    TRY(bond_compute(1));
    TRY(angle_compute(1));
    TRY(check_overflow());
    is_setup = true;
    return 0;
}

// The rebuild's count in the epilogue of the force launch in front of it (FrCountArgs, kernels.h; option fuse_count).  Everything the
// rebuild would do before its first kernel happens here, ahead of that force launch: the previous rebuild's counts are read, the arrays
// get the capacity the rebuild asks for (they may move: the caller builds its launch arguments afterwards), the fused rebuild's buffers
// are there and clean.  ok = false: this rebuild keeps its own count kernel (several ranks, a rebuild that will not take the fused
// path, a list builder stage that is about to grow).
int Engine::prepare_count_in_epilogue(FrCountArgs &c, bool &ok)
{
    ok = false;
    if (!fuse_count || nranks != 1 || reorder_sort || dist_check || mig_holes || fused_dirty) return 0;
    TRY(resolve_counts());
    if (!fused_ok() || !params_ready) return 0;
    if ((h_flags[6] || h_flags[11]) && !brick2_off) return 0;                                   // the 2-brick stage is about to change
    if (neigh_kernel == 1 && (long)std::max(h_flags[5], h_flags[10]) * 100 > (long)bargs.maxh * 93) return 0;      // ... or the 4-brick's
    const int bound = (int)(nghost_prev * async_grid_scale) + 1024;
    // No regrow here: alloc_atoms() keeps the atoms but not the neighbour table nor the merged records, and this step's force launch
    // still reads both.  A rebuild that needs more room runs its own count kernel and regrows where the table is rebuilt anyway
    // (rebuild_fused); tests/test_gpu_parity.py::test_count_in_epilogue_does_not_regrow_ahead_of_the_force_launch.
    if (nlocal + bound + bound / 2 > nmax) return 0;
    TRY(fused_alloc());
    wrap_in_reorder = true;
    FusedArgs a;
    fused_locals_args(a, true);
    c = fused_count_args(a);
    ok = true;
    return 0;
}

int Engine::run(int nsteps)
{
    if (!is_setup) return fail(3, "run before setup");
    tbegin("total_steps");
    bool initial_done = false, merged = false, ghosts_by_epilogue = false;
    // checkpoint of the last rebuild of this run (host side; the device side is the untouched state itself, see prepare_redo)
    int ck_it = -1, ck_ago = 0, redo_pending = 0;
    int64_t ck_step = 0;
    // TRY inside the loop: a rebuild of this run that has to be redone (MESO_REDO from resolve_counts) restarts the loop at its step
#define TRY_STEP(call)                                                              \
    do {                                                                            \
        int _rc = (call);                                                           \
        if (_rc == MESO_REDO && ck_it >= 0) { redo_pending = 1; goto redo_rebuild; } \
        if (_rc) { redo_armed = false; return _rc == MESO_REDO ? fail(4, "A rebuild outgrew its capacities outside a run") : _rc; } \
    } while (0)
    // (armed for the duration of this call only, whichever way it returns: outside run() an outgrown capacity is an error)
    struct Disarm { bool &b; ~Disarm() { b = false; } } disarm{redo_armed};
    redo_armed = true;
    for (int it = 0; it <= nsteps; it++) {
        if (it == nsteps) {
            // the end of the run: the last rebuild's report is read here, while it can still be redone
            TRY_STEP(resolve_counts());
            break;
        }
        if (false) {
        redo_rebuild:
            // back to the step of the rebuild that failed: its initial integration is part of the state, the rebuild is due again
            it = ck_it; ntimestep = ck_step - 1; ago = ck_ago;
            initial_done = true; merged = false; ghosts_by_epilogue = false;
        }
        profile_tick(it, nsteps);
        ntimestep++;
        if (!initial_done) {
            // the first step of a run (or the step behind an unfused boundary): when it provably keeps the neighbour table, the
            // initial integration and the merge of the locals are one pass
            const bool keeps = !dist_check && !redo_pending && !(ago + 1 >= delay && (ago + 1) % every == 0);
            if (keeps && fuse_step && !merged) {
                tbegin("nve");
                launch_nve_initial_merge(cur, 0.5 * dt, dt, groupbit, nlocal, coord4, veloc4, 0.5 * (subhi[0] + sublo[0]),
                                         0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]), premix_tea<64>((u32)seed, (u32)ntimestep),
                                         stream, d_flags);
                tend("nve");
                merged = true;
            } else TRY(nve_initial());
        }
        int rebuild = 0;
        if (redo_pending) { rebuild = 1; ago++; }
        else TRY(decide(&rebuild));
        bool ghosts_fresh = false;
        if (rebuild) {
            const int ago_before = ago - 1;
            permute_forces = false;                     // this step's force kernel overwrites them
            int rr = reneighbor();
            permute_forces = true;
            // (MESO_REDO here: the report of the PREVIOUS rebuild, read at the top of reneighbor - the checkpoint still names it)
            if (rr == MESO_REDO && !redo_pending && ck_it >= 0) { redo_pending = 1; goto redo_rebuild; }
            if (rr) { redo_armed = false; return rr == MESO_REDO ? fail(4, "A rebuild outgrew its capacities twice in a row") : rr; }
            ck_it = it; ck_step = ntimestep; ck_ago = ago_before;
            redo_pending = 0;
            merged = true; ghosts_fresh = true;    // the rebuild merged with this step's seed
        }
        // the step in front of a rebuild inside this run: its force launch can run the rebuild's count in its epilogue (arrays may be
        // regrown here, before any launch argument of this step is taken)
        bool count_here = false;
        FrCountArgs frc_args;
        {
            const int a1_ = ago + 1;
            const bool next_rebuild_ = dist_check || (a1_ >= delay && a1_ % every == 0);
            const bool fusable = fuse_pair && fuse_step && it + 1 < nsteps && (!have_bonds || nbondtypes > 0) && ring_selected() && nranks == 1;
            if (next_rebuild_ && fusable) TRY_STEP(prepare_count_in_epilogue(frc_args, count_here));
        }
        u32 sd = premix_tea<64>((u32)seed, (u32)ntimestep);
        if (!merged) TRY(merge_locals(sd));
        // (several ranks: the counts of the last rebuild have arrived long ago - the split point and the refresh tables need them)
        if (mr_pending && !rebuild) TRY(resolve_counts());
        // bulk/border split point, rounded down to the force kernel's 256-atom groups (Newton pairing needs whole groups);
        // the few bulk atoms behind it simply wait for the ghosts too
        const int n_split = n_bulk & ~(pair_ring_group() - 1);
        const bool split = nranks > 1 && overlap && n_split > 0 && n_split < nlocal && !mr_pending;      // (pending: ghosts are fresh)
        if (!ghosts_fresh && !ghosts_by_epilogue) TRY(halo_forward_seed(sd, split));
        ghosts_by_epilogue = false;
        PairArgs p = {};
        p.bond.nbond = nullptr;
        p.coord4 = coord4; p.veloc4 = veloc4; p.count = pair_count; p.table = pair_table; p.n_col = n_col;
        for (int d = 0; d < 3; d++) p.f[d] = cur.f[d];
        p.e_pair = nullptr;
        for (int k = 0; k < 6; k++) p.virial[k] = nullptr;
        p.coeff64 = d_coeff64; p.coeff32 = d_coeff32; p.ntypes = ntypes;
        for (int k = 0; k < 7; k++) p.cf1[k] = coeff[k];
        p.dt_inv_sqrt = 1.0 / std::sqrt(dt);
        p.accumulate = fuse_clear ? 0 : 1;
        p.debug = pair_debug;
        p.chunked = 1;
        if (!fuse_clear) TRY(force_clear(0));
        // step boundary in the force kernel's epilogue: final(s) + initial(s+1) + merge(s+1) of the atoms a launch owns;
        // the merged arrays of step s+1 go to the second buffer (this step's are still being gathered from)
        const int a1 = ago + 1;
        const bool next_rebuild = dist_check || (a1 >= delay && a1 % every == 0);
        // bonded systems: bond and angle forces depend on this step's merged coordinates only, so they are computed FIRST (the
        // bond kernel opens the force arrays) and the force kernel's epilogue adds them to its sums before the step boundary.
        // Not with the bulk/border overlap of several ranks: bonds across a face need the ghosts the bulk launch does not wait for.
        const bool bonded_first = have_bonds && nbondtypes > 0 && !split;
        const bool boundary_in_pair = fuse_pair && fuse_step && it + 1 < nsteps && (!have_bonds || bonded_first) && ring_selected();
        p.fuse_nve = boundary_in_pair ? 1 : 0;
        p.frc_on = 0;
        if (count_here && boundary_in_pair && next_rebuild && !split) { p.frc_on = 1; p.frc = frc_args; count_in_epilogue = true; }
        if (boundary_in_pair && bonded_first) {
            if (fuse_bonds && !have_angles) {
                // ... or, without angles, inside the force kernel's epilogue: each atom's few bonds are evaluated there (the same
                // device function the bond kernel calls) - no bond launch, no force arrays written and read back
                if (!d_bond_kr0) {
                    HIPCHK(dalloc(d_bond_kr0, bond_kr0.size()));
                    HIPCHK(hipMemcpy(d_bond_kr0, bond_kr0.data(), bond_kr0.size() * sizeof(double), hipMemcpyHostToDevice));
                }
                p.bond.nbond = cur.nbond; p.bond.bond_idx = bond_idx; p.bond.bond_type = cur.bond_type;
                p.bond.bpa = bpa; p.bond.nbt = nbondtypes; p.bond.style = bond_kind; p.bond.cf = d_bond_kr0;
                for (int d = 0; d < 3; d++) p.bond.prd[d] = prd[d];
            } else {
                TRY(bond_compute(0, 1));
                TRY(angle_compute(0));
                p.accumulate = 1;
            }
        }
        if (boundary_in_pair)
            p.nve = make_nve_args(cur, 0.5 * dt, dt, groupbit, next_rebuild ? 0 : 1, coord4_next, veloc4_next,
                                  0.5 * (subhi[0] + sublo[0]), 0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]),
                                  premix_tea<64>((u32)seed, (u32)(ntimestep + 1)));
        if (boundary_in_pair && lean_boundary) {
            // (type from the merged record, mass - as dtf / m, evaluated once per type - from the per-type table, no mask of group "all")
            if (dtfm_for != 0.5 * dt) { launch_dtfm_table(d_mass_type, ntypes, 0.5 * dt, d_dtfm_type, stream); dtfm_for = 0.5 * dt; }
            p.nve.mass_type = d_mass_type; p.nve.dtfm_type = d_dtfm_type;
        }
        // small boxes on one rank: the epilogue also writes the merged pairs of the atom's periodic images for step s+1
        const bool img_step = boundary_in_pair && !next_rebuild && images_ready && images_on() && !split;
        if (img_step) { p.nve.img_cnt = img_cnt; p.nve.img = img; p.nve.img_shift = d_shift27; }
        // (fused rebuild: the order is [bulk][border] and estart[M] is the first border atom - only those can have images)
        if (img_step && fused_active && lean_boundary) p.nve.img_first = estart + bargs.M;
        // several ranks: the epilogue writes the next step's refresh messages into the send staging (tables from the rebuild's
        // border kernel); bulk atoms have no entries, so with the bulk/border split only the border launch writes
        const bool mr_img_step = boundary_in_pair && !next_rebuild && nranks > 1 && mr_images_ready && stage_send == mr_img_stage;
        if (mr_img_step) {
            p.nve.img_cnt = img_cnt; p.nve.img = img; p.nve.img_shift = d_shift27;
            p.nve.img_c4 = (float4 *)stage_send; p.nve.img_v4 = (float4 *)stage_send; p.nve.img_vofs = d_vofs; p.nve.img_center = d_center27;
        }
        // bulk atoms have no ghost partners: their forces are computed while the ghosts are in flight
        for (int part = 0; part < (split ? 2 : 1); part++) {
            p.beg = split ? (part == 0 ? 0 : n_split) : 0;
            p.end = split ? (part == 0 ? n_split : nlocal) : nlocal;
            if (split && part == 1 && !ghosts_fresh) TRY(halo_wait());
            {
                tbegin("pair");
                launch_pair(p, 0);
                tend("pair");
            }
        }
        if (!(boundary_in_pair && bonded_first)) {
            TRY(bond_compute(0));
            TRY(angle_compute(0));
        }
        if (boundary_in_pair) {
            if (!next_rebuild) { std::swap(coord4, coord4_next); std::swap(veloc4, veloc4_next); }
            initial_done = true;
            merged = !next_rebuild;
            ghosts_by_epilogue = img_step;
            fwd_packed = mr_img_step;
        } else if (fuse_step && it + 1 < nsteps) {
            // one pass for final(s) + initial(s+1); the merge for s+1 rides along when s+1 provably keeps the table
            tbegin("nve");
            launch_nve_boundary(cur, 0.5 * dt, dt, groupbit, nlocal, next_rebuild ? 0 : 1, coord4, veloc4,
                                0.5 * (subhi[0] + sublo[0]), 0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]),
                                premix_tea<64>((u32)seed, (u32)(ntimestep + 1)), stream, d_flags);
            tend("nve");
            initial_done = true;
            merged = !next_rebuild;
            fwd_packed = false;
        } else {
            fwd_packed = false;
            TRY(nve_final());
            initial_done = false;
            merged = false;
        }
        ev_valid = false;
        // a launch that the runtime refused (configuration, resources) is reported at the step that made it, not at the end of the run
        // (a host-side query, no synchronisation; faults inside kernels still surface with the next synchronisation - option
        // check_launches 1 waits behind every stage of a rebuild)
        if (launch_refused) { launch_refused = false; return fail(2, "The force kernel has no form for this combination of row layout and record format"); }
        if (hipError_t le = hipPeekAtLastError(); le != hipSuccess && le != hipErrorNotReady) {      // (NotReady: a stream / event query that found work in flight)
            char msg[200];
            snprintf(msg, sizeof msg, "HIP launch error at timestep %ld: %s", (long)ntimestep, hipGetErrorString(le));
            (void)hipGetLastError();
            return fail(2, msg);
        }
    }
#undef TRY_STEP
    redo_armed = false;
    tend("total_steps");
    profile_tick(nsteps, nsteps);
    TRY(resolve_counts());
    TRY(check_overflow());
    HIPCHK(hipGetLastError());
    return 0;
}

int Engine::sync()
{
    HIPCHK(hipStreamSynchronize(stream));
    return 0;
}

// float4 copy rate on this device (GB/s, read + write): the measured peak quoted beside the nominal 8 TB/s
int Engine::membw_probe(size_t nbytes, int reps, double *gbs)
{
    const size_t n = nbytes / 16;
    float4 *a = nullptr, *b = nullptr;
    HIPCHK(hipMalloc(&a, n * 16));
    if (hipMalloc(&b, n * 16) != hipSuccess) { (void)hipFree(a); return fail(2, "membw_probe: out of device memory"); }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemsetAsync(a, 0, n * 16, stream);
    launch_copy_f4(a, b, n, stream);      // warm-up (page mapping)
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        (void)hipEventRecord(e0, stream);
        launch_copy_f4(a, b, n, stream);
        (void)hipEventRecord(e1, stream);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms > 0.f && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    HIPCHK(hipGetLastError());
    *gbs = 2.0 * (double)n * 16.0 / ((double)best * 1e-3) / 1e9;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// computes
// ------------------------------------------------------------------------------------------------
int Engine::compute_temp(double *t)
{
    TRY(init_params());
    launch_sum_mv2(cur, groupbit, nlocal, d_partial, d_scalar, stream);
    HIPCHK(hipMemcpyAsync(h_scalar, d_scalar, sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    double sum = reduce_global_sum(h_scalar[0]);
    double dof = 3.0 * natoms_total - 3.0;   // extra_dof = 3 (src/compute_temp.cpp dof_compute)
    *t = dof > 0.0 ? sum / dof : 0.0;        // units lj: mvv2e = boltz = 1
    return 0;
}

static int host_sum(double *d, int n, hipStream_t s, double &out)
{
    std::vector<double> h((size_t)n);
    if (hipMemcpyAsync(h.data(), d, n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return 2;
    if (hipStreamSynchronize(s) != hipSuccess) return 2;
    long double acc = 0.0L;
    for (int i = 0; i < n; i++) acc += h[i];
    out = (double)acc;
    return 0;
}

// Energy and virial of the current configuration without disturbing the stored forces (thermo output between run chunks)
int Engine::tally_ev()
{
    if (!is_setup) return fail(3, "tally before setup");
    if (ev_valid) return 0;
    for (int d = 0; d < 3; d++)
        HIPCHK(hipMemcpyAsync(alt.f[d], cur.f[d], (size_t)nlocal * sizeof(double), hipMemcpyDeviceToDevice, stream));
    TRY(force_clear(0));
    TRY(pair_compute(0, 1, 1));
    TRY(bond_compute(1));
    TRY(angle_compute(1));
    for (int d = 0; d < 3; d++)
        HIPCHK(hipMemcpyAsync(cur.f[d], alt.f[d], (size_t)nlocal * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return 0;
}

int Engine::compute_pe(double *pe)
{
    if (!ev_valid) {
        // recompute with energy/virial tallies on the current configuration (no force change: scratch f)
        return fail(3, "Energy was not tallied on this timestep");
    }
    double s = 0.0;
    if (host_sum(e_pair, nlocal, stream, s)) return fail(2, "HIP error in compute_pe");
    *pe = reduce_global_sum(s);
    return 0;
}

int Engine::compute_pressure(double *p)
{
    if (!ev_valid) return fail(3, "Virial was not tallied on this timestep");
    double t = 0.0;
    TRY(compute_temp(&t));
    double vsum = 0.0;
    for (int k = 0; k < 3; k++) {
        double s = 0.0;
        if (host_sum(virial[k], nlocal, stream, s)) return fail(2, "HIP error in compute_pressure");
        vsum += s;
    }
    vsum = reduce_global_sum(vsum);
    double dof = 3.0 * natoms_total - 3.0;
    double vol = prd[0] * prd[1] * prd[2];
    *p = (dof * t + vsum) / 3.0 / vol;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// introspection / tests
// ------------------------------------------------------------------------------------------------
// (partitioned rows, RowPartArgs in kernels.h: an atom's neighbours are its front row plus its back row)
int Engine::neigh_info(int *ncol, int *max_count, double *avg, int64_t *nb)
{
    HIPCHK(hipStreamSynchronize(stream));
    std::vector<int> h((size_t)nlocal), hb;
    if (nlocal) HIPCHK(hipMemcpy(h.data(), pair_count, nlocal * sizeof(int), hipMemcpyDeviceToHost));
    if (rows_part && nlocal) { hb.resize(nlocal); HIPCHK(hipMemcpy(hb.data(), pair_nback, nlocal * sizeof(int), hipMemcpyDeviceToHost)); }
    long tot = 0;
    int mx = 0;
    for (int i = 0; i < nlocal; i++) {
        const int n = h[i] + (rows_part ? hb[i] : 0);
        tot += n; mx = std::max(mx, n);
    }
    if (ncol) *ncol = n_col;
    if (max_count) *max_count = mx;
    if (avg) *avg = nlocal ? (double)tot / nlocal : 0.0;
    if (nb) *nb = nbuild;
    return 0;
}

static void unchunk_rows(const std::vector<int> &h, int pitch, int nlocal, bool raw, const int *count, int *table, int stride, int col0)
{
    for (int i = 0; i < nlocal; i++) {
        const int n = raw ? (count[i] + 7) & ~7 : count[i];      // (raw: the tail slots of the last chunk too)
        for (int p = 0; p < n && col0 + p < stride; p++)
            table[(size_t)i * stride + col0 + p] = h[((((size_t)(i >> 6)) * (pitch >> 3) + (p >> 3)) * 64 + (i & 63)) * 8 + (p & 7)];
    }
}

// sections of the table in use as stored (padding of the last chunk included): front[i * stride + p], back[i * stride + p]
int Engine::neigh_parts(int *parted, int *group, int *nfront, int *nback, int *front, int *back, int stride)
{
    HIPCHK(hipStreamSynchronize(stream));
    *parted = rows_part ? 1 : 0; *group = part_group;
    if (!nlocal) return 0;
    std::vector<int> cf(nlocal), cb(nlocal, 0);
    HIPCHK(hipMemcpy(cf.data(), pair_count, nlocal * sizeof(int), hipMemcpyDeviceToHost));
    if (rows_part) HIPCHK(hipMemcpy(cb.data(), pair_nback, nlocal * sizeof(int), hipMemcpyDeviceToHost));
    if (nfront) std::copy(cf.begin(), cf.end(), nfront);
    if (nback) std::copy(cb.begin(), cb.end(), nback);
    const size_t tiles = ((size_t)nlocal + 63) / 64;
    if (front) {
        std::vector<int> h(tiles * 64 * (size_t)n_col);
        HIPCHK(hipMemcpy(h.data(), pair_table, h.size() * sizeof(int), hipMemcpyDeviceToHost));
        unchunk_rows(h, n_col, nlocal, true, cf.data(), front, stride, 0);
    }
    if (back && rows_part) {
        std::vector<int> h(tiles * 64 * (size_t)nb_col);
        HIPCHK(hipMemcpy(h.data(), pair_back, h.size() * sizeof(int), hipMemcpyDeviceToHost));
        unchunk_rows(h, nb_col, nlocal, true, cb.data(), back, stride, 0);
    }
    return 0;
}

// the neighbours of every atom, front section first
int Engine::neigh_download(int *count, int *table, int stride)
{
    HIPCHK(hipStreamSynchronize(stream));
    if (!nlocal) return 0;
    HIPCHK(hipMemcpy(count, pair_count, nlocal * sizeof(int), hipMemcpyDeviceToHost));
    const size_t tiles = ((size_t)nlocal + 63) / 64;
    std::vector<int> h(tiles * 64 * (size_t)n_col);
    HIPCHK(hipMemcpy(h.data(), pair_table, h.size() * sizeof(int), hipMemcpyDeviceToHost));
    unchunk_rows(h, n_col, nlocal, false, count, table, stride, 0);
    if (rows_part) {
        std::vector<int> cb(nlocal), hb(tiles * 64 * (size_t)nb_col);
        HIPCHK(hipMemcpy(cb.data(), pair_nback, nlocal * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hb.data(), pair_back, hb.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < nlocal; i++) {
            for (int p = 0; p < cb[i] && count[i] + p < stride; p++)
                table[(size_t)i * stride + count[i] + p] = hb[((((size_t)(i >> 6)) * (nb_col >> 3) + (p >> 3)) * 64 + (i & 63)) * 8 + (p & 7)];
            count[i] += cb[i];
        }
    }
    return 0;
}

int Engine::merged_download(float *c4, float *v4, int nall)
{
    HIPCHK(hipStreamSynchronize(stream));
    if (nall > nlocal + nghost) return fail(1, "merged_download: nall exceeds nlocal+nghost");
    if (c4) HIPCHK(hipMemcpy(c4, coord4, (size_t)nall * sizeof(float4), hipMemcpyDeviceToHost));
    if (v4) HIPCHK(hipMemcpy(v4, veloc4, (size_t)nall * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}

int Engine::test_tea(int n, int rounds, const uint32_t *u, const uint32_t *v, uint32_t *o0, uint32_t *o1)
{
    if (!stream) TRY(alloc_atoms(1024));
    uint32_t *du, *dv, *d0, *d1;
    HIPCHK(dalloc(du, n)); HIPCHK(dalloc(dv, n)); HIPCHK(dalloc(d0, n)); HIPCHK(dalloc(d1, n));
    HIPCHK(hipMemcpy(du, u, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv, v, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    launch_test_tea(du, dv, n, rounds, d0, d1, stream);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(o0, d0, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(o1, d1, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    dfree(du); dfree(dv); dfree(d0); dfree(d1);
    return 0;
}

int Engine::test_gaussian(int n, const uint32_t *u, const uint32_t *v, double *odp, float *osp)
{
    if (!stream) TRY(alloc_atoms(1024));
    uint32_t *du, *dv;
    double *dd;
    float *ds;
    HIPCHK(dalloc(du, n)); HIPCHK(dalloc(dv, n)); HIPCHK(dalloc(dd, n)); HIPCHK(dalloc(ds, n));
    HIPCHK(hipMemcpy(du, u, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv, v, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    launch_test_gaussian(du, dv, n, dd, ds, stream);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(odp, dd, n * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(osp, ds, n * sizeof(float), hipMemcpyDeviceToHost));
    dfree(du); dfree(dv); dfree(dd); dfree(ds);
    return 0;
}

int Engine::test_logistic(int n, const uint32_t *u, const uint32_t *v, float *out)
{
    if (!stream) TRY(alloc_atoms(1024));
    uint32_t *du, *dv;
    float *ds;
    HIPCHK(dalloc(du, n)); HIPCHK(dalloc(dv, n)); HIPCHK(dalloc(ds, n));
    HIPCHK(hipMemcpy(du, u, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dv, v, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    launch_test_logistic(du, dv, n, ds, stream);
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipMemcpy(out, ds, n * sizeof(float), hipMemcpyDeviceToHost));
    dfree(du); dfree(dv); dfree(ds);
    return 0;
}

} // namespace meso
