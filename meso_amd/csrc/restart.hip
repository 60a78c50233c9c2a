// Restart files of the device-resident state and the profiler window.
//
// Inside a LAMMPS tree the restart file is LAMMPS' own: MesoHipIntegrate::download hands the atoms back and the glue's
// Pair::write_restart / read_restart keep the byte layout of MesoPairDPD::write_restart (pair_dpd_meso.cu:363-447).  For the
// stand-alone driver (commands write_restart / read_restart) the engine writes its own file, one per rank: settings, pair
// and bonded coefficients, and the per-atom arrays exactly as they live on the device (fp64 x and v, tags, image flags,
// bond / special / angle lists) plus the forces of the interrupted step.  Reading a file written on a rebuild step and
// calling setup continues the run bit for bit (the force sums do not depend on the storage order; tests/test_gpu_restart.py);
// between rebuilds the positions are unwrapped and round differently in the fp32 merged coordinates after setup's wrap.
//
// Profiler window: MesoDevice::configure_profiler (engine_meso.cu:155-177) starts and stops the CUDA profiler at chosen
// timesteps (-profile all | core | loop | interval a b).  Here the same windows pause / resume rocprofv3 through the roctx
// control calls, looked up at run time so that the library does not depend on the profiler being installed.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>

#include "engine.h"
#include "meso_device.h"

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

namespace {
struct FileCloser {      // closes on every return path (the HIPCHK / TRY macros return early)
    FILE *f;
    ~FileCloser() { if (f) fclose(f); }
    int close() { int rc = f ? fclose(f) : 0; f = nullptr; return rc; }
};
const char MAGIC[8] = {'M', 'E', 'S', 'O', 'H', 'I', 'P', '1'};

struct Writer {
    FILE *f;
    bool ok = true;
    template <class T> void pod(const T &v) { ok = ok && fwrite(&v, sizeof(T), 1, f) == 1; }
    template <class T> void vec(const std::vector<T> &v)
    {
        uint64_t n = v.size();
        pod(n);
        if (n) ok = ok && fwrite(v.data(), sizeof(T), n, f) == n;
    }
};
struct Reader {
    FILE *f;
    bool ok = true;
    template <class T> void pod(T &v) { ok = ok && fread(&v, sizeof(T), 1, f) == 1; }
    template <class T> void vec(std::vector<T> &v)
    {
        uint64_t n = 0;
        pod(n);
        if (!ok || n > (1ull << 33)) { ok = false; return; }
        v.resize(n);
        if (n) ok = ok && fread(v.data(), sizeof(T), n, f) == n;
    }
};
template <class T> hipError_t d2h(std::vector<T> &h, const T *d, size_t n)
{
    h.resize(n);
    return n ? hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) : hipSuccess;
}
} // namespace

static std::string rank_path(const std::string &path, int nranks, int rank)
{
    return nranks > 1 ? path + "." + std::to_string(rank) : path;
}

int Engine::write_restart(const std::string &path)
{
    if (!have_box || !stream) return fail(3, "write_restart before the box and the atoms exist");
    HIPCHK(hipStreamSynchronize(stream));
    FILE *f = fopen(rank_path(path, nranks, rank).c_str(), "wb");
    if (!f) return fail(2, "Cannot open restart file " + path);
    FileCloser fc{f};
    Writer w{f};
    fwrite(MAGIC, 1, 8, f);
    w.pod(nranks); w.pod(rank);
    for (int d = 0; d < 3; d++) w.pod(procgrid[d]);
    w.pod(ntimestep); w.pod(dt); w.pod(skin); w.pod(every); w.pod(delay); w.pod(dist_check);
    for (int d = 0; d < 3; d++) { w.pod(boxlo[d]); w.pod(boxhi[d]); w.pod(periodic[d]); }
    w.pod(ntypes); w.vec(mass_type);
    int flags[4] = {have_pair ? 1 : 0, pair_rng, pair_poly ? 1 : 0, pair_ftab ? 1 : 0};
    for (int k = 0; k < 4; k++) w.pod(flags[k]);
    w.pod(pair_style); w.pod(cut_global); w.pod(seed);
    w.vec(coeff); w.vec(coeff_set); w.vec(poly); w.pod(ftab_len); w.vec(ftab);
    int topo[8] = {have_bonds ? 1 : 0, bpa, msp, maxtag, nbondtypes, bond_kind, have_angles ? 1 : 0, apa};
    for (int k = 0; k < 8; k++) w.pod(topo[k]);
    for (int k = 0; k < 3; k++) w.pod(special_w[k]);
    w.vec(bond_kr0); w.pod(nangletypes); w.vec(angle_cf);
    // atoms, as stored (cell order)
    const size_t n = (size_t)nlocal;
    w.pod(nlocal);
    std::vector<double> hd;
    std::vector<int> hi;
    for (int d = 0; d < 3; d++) { HIPCHK(d2h(hd, cur.x[d], n)); w.vec(hd); }
    for (int d = 0; d < 3; d++) { HIPCHK(d2h(hd, cur.v[d], n)); w.vec(hd); }
    // the forces too: they were computed from the velocities of the half step, which no longer exist - recomputing them in
    // setup (what LAMMPS does after read_restart) would not reproduce the uninterrupted trajectory of a DPD fluid
    for (int d = 0; d < 3; d++) { HIPCHK(d2h(hd, cur.f[d], n)); w.vec(hd); }
    HIPCHK(d2h(hi, cur.tag, n)); w.vec(hi);
    HIPCHK(d2h(hi, cur.type, n)); w.vec(hi);
    HIPCHK(d2h(hi, cur.mask, n)); w.vec(hi);
    HIPCHK(d2h(hi, cur.image, n)); w.vec(hi);
    if (have_bonds) {
        HIPCHK(d2h(hi, cur.nbond, n)); w.vec(hi);
        HIPCHK(d2h(hi, cur.bond_tag, n * bpa)); w.vec(hi);
        HIPCHK(d2h(hi, cur.bond_type, n * bpa)); w.vec(hi);
        HIPCHK(d2h(hi, cur.nspecial, n)); w.vec(hi);
        HIPCHK(d2h(hi, cur.special, n * msp)); w.vec(hi);
    }
    if (have_angles) {
        HIPCHK(d2h(hi, cur.nangle, n)); w.vec(hi);
        HIPCHK(d2h(hi, cur.angle_tag, n * 4 * apa)); w.vec(hi);
    }
    const bool ok = w.ok && fc.close() == 0;
    return ok ? 0 : fail(2, "Error while writing restart file " + path);
}

int Engine::read_restart(const std::string &path)
{
    // (the estimates and message capacities of earlier rebuilds say nothing about the state in the file)
    nghost_prev = -1; n_bulk_prev = -1; mr_caps_ready = false; mig_caps_ready = false; counts_pending = false; mr_pending = false; bulk_pending = false;
    FILE *f = fopen(rank_path(path, nranks, rank).c_str(), "rb");
    if (!f) return fail(2, "Cannot open restart file " + path);
    FileCloser fc{f};
    Reader r{f};
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, MAGIC, 8)) { return fail(2, "Not a meso-hip restart file: " + path); }
    int nr = 0, rk = 0, pg[3] = {1, 1, 1};
    r.pod(nr); r.pod(rk);
    for (int d = 0; d < 3; d++) r.pod(pg[d]);
    if (!r.ok || nr != nranks || rk != rank || pg[0] != procgrid[0] || pg[1] != procgrid[1] || pg[2] != procgrid[2]) {
        return fail(2, "Restart file was written by a different decomposition (the files are per rank)");
    }
    int64_t step = 0;
    double lo[3], hi[3];
    int per[3];
    r.pod(step); r.pod(dt); r.pod(skin); r.pod(every); r.pod(delay); r.pod(dist_check);
    for (int d = 0; d < 3; d++) { r.pod(lo[d]); r.pod(hi[d]); r.pod(per[d]); }
    int nt = 0;
    std::vector<double> mt;
    r.pod(nt); r.vec(mt);
    if (!r.ok || nt < 1 || (int)mt.size() != nt + 1) { return fail(2, "Restart file is damaged: " + path); }
    TRY(set_box(lo, hi, per));
    ntypes = 0;
    TRY(set_mass(nt, mt.data()));
    int flags[4];
    for (int k = 0; k < 4; k++) r.pod(flags[k]);
    r.pod(pair_style); r.pod(cut_global); r.pod(seed);
    have_pair = flags[0] != 0; pair_rng = flags[1]; pair_poly = flags[2] != 0; pair_ftab = flags[3] != 0;
    r.vec(coeff); r.vec(coeff_set); r.vec(poly); r.pod(ftab_len); r.vec(ftab);
    have_coeff = have_pair;
    int topo[8];
    for (int k = 0; k < 8; k++) r.pod(topo[k]);
    for (int k = 0; k < 3; k++) r.pod(special_w[k]);
    r.vec(bond_kr0); r.pod(nangletypes); r.vec(angle_cf);
    int n = 0;
    r.pod(n);
    if (!r.ok || n < 0 || (int)coeff.size() != nt * nt * N_COEFF) { return fail(2, "Restart file is damaged: " + path); }
    std::vector<double> x[3], v[3], fr[3];
    std::vector<int> tag, type, mask, image;
    for (int d = 0; d < 3; d++) r.vec(x[d]);
    for (int d = 0; d < 3; d++) r.vec(v[d]);
    for (int d = 0; d < 3; d++) r.vec(fr[d]);
    r.vec(tag); r.vec(type); r.vec(mask); r.vec(image);
    if (!r.ok || (int)tag.size() != n || (int)x[2].size() != n || (int)image.size() != n) { return fail(2, "Restart file is damaged: " + path); }
    std::vector<double> xa((size_t)3 * n), va((size_t)3 * n);
    for (int i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) { xa[3 * (size_t)i + d] = x[d][i]; va[3 * (size_t)i + d] = v[d][i]; }
    // every atom of the file is mine, wherever it has drifted since the last rebuild: no ownership filter
    upload_all = true;
    bpa = msp = apa = 0;
    int rc = atoms_upload(n, xa.data(), va.data(), tag.data(), type.data(), mask.data(), image.data());
    upload_all = false;
    if (rc) return rc;
    for (int d = 0; d < 3; d++)
        if ((int)fr[d].size() == n && n) HIPCHK(hipMemcpy(cur.f[d], fr[d].data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    dfree(d_bond_kr0); dfree(d_angle_cf);
    nbondtypes = topo[4]; bond_kind = topo[5];
    if (topo[0]) {
        bpa = topo[1]; msp = topo[2]; maxtag = topo[3];
        apa = topo[6] ? topo[7] : 0;
        rc = alloc_atoms(std::max(nmax, 1024));
        std::vector<int> a, b, c, d, e;
        r.vec(a); r.vec(b); r.vec(c); r.vec(d); r.vec(e);
        if (!rc && r.ok && (int)a.size() == n && (int)b.size() == n * bpa && (int)e.size() == n * msp && n) {
            HIPCHK(hipMemcpy(cur.nbond, a.data(), a.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(cur.bond_tag, b.data(), b.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(cur.bond_type, c.data(), c.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(cur.nspecial, d.data(), d.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(cur.special, e.data(), e.size() * sizeof(int), hipMemcpyHostToDevice));
        } else if (n) r.ok = false;
        if (topo[6]) {
            r.vec(a); r.vec(b);
            if (!rc && r.ok && (int)a.size() == n && (int)b.size() == n * 4 * apa && n) {
                HIPCHK(hipMemcpy(cur.nangle, a.data(), a.size() * sizeof(int), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(cur.angle_tag, b.data(), b.size() * sizeof(int), hipMemcpyHostToDevice));
            } else if (n) r.ok = false;
        }
        dfree(tagmap);
        HIPCHK(dalloc(tagmap, (size_t)maxtag + 2));
        // (the file holds this rank's lists only: every tag enters the tag map after a restart)
        h_tagbits.clear();
        dfree(tagbits);
        have_bonds = true;
        have_angles = topo[6] != 0;
    }
    fc.close();
    if (rc) return rc;
    if (!r.ok) return fail(2, "Restart file is damaged: " + path);
    ntimestep = step;
    restart_forces = true;      // setup keeps the forces of the file instead of recomputing them
    is_setup = false;
    params_ready = false;
    return 0;
}

// -profile all | core | loop | interval a b (engine_meso.cu:155-177)
int Engine::profile_window(int mode, int64_t start, int64_t end)
{
    if (mode < 0 || mode > 4) return fail(1, "Illegal profile window");
    if (mode == 4 && end < start) return fail(1, "Illegal profile window");
    prof_mode = mode; prof_start = start; prof_end = end;
    if (mode && !prof_pause) {
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            prof_pause = (int (*)(uint64_t))dlsym(h, "roctxProfilerPause");
            prof_resume = (int (*)(uint64_t))dlsym(h, "roctxProfilerResume");
        }
    }
    if (getenv("MESO_DEBUG_PROFILE")) fprintf(stderr, "profile_window: mode %d, roctxProfilerPause %s, roctxProfilerResume %s\n", mode, prof_pause ? "found" : "MISSING", prof_resume ? "found" : "MISSING");
    // modes with a window inside the run start with collection paused ("all" brackets the whole program instead)
    if (mode >= 2 && prof_pause) prof_pause(0);
    return 0;
}

// called by run() before step `it` of `nsteps` (it == nsteps: after the last one)
void Engine::profile_tick(int it, int nsteps)
{
    if (prof_mode < 2 || !prof_resume || !prof_pause) return;
    int64_t a, b;
    if (prof_mode == 2) { a = nsteps / 4; b = nsteps * 3 / 4; }          // core: the middle half of the run
    else if (prof_mode == 3) { a = 0; b = nsteps; }                          // loop
    else { a = prof_start - (ntimestep - it); b = prof_end - (ntimestep - it); }   // interval: absolute timesteps
    if (it == a) { (void)hipStreamSynchronize(stream); prof_resume(0); prof_windows++; }
    if (it == b) { (void)hipStreamSynchronize(stream); prof_pause(0); }
}

} // namespace meso
