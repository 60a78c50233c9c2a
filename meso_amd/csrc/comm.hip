// Spatial decomposition over several ranks (one rank per GPU): migration, ghost creation and the per-step
// ghost refresh.  Replaces MesoComm::exchange / borders (/root/reference/src/USER-MESO/comm_meso.cu:256-550, 41-186)
// and Comm::forward_comm (/root/reference/src/comm.cpp), which stage everything through pinned host arrays and
// six serialized MPI_Sendrecv swaps.  Here each rank packs on the device straight from the SoA arrays and talks
// to its (up to 26) brick neighbours directly: one message per peer per phase, all peers in one grouped RCCL
// call over xGMI.  The per-step payload is the merged float4 pair already in the receiver's frame (32 B/ghost;
// the reference sends 48 B of fp64 x,v and re-merges on the receiver).
//
// Transports: RCCL (ncclSend/ncclRecv grouped), LOCAL (several ranks inside one process, device-to-device
// copies: how the multi-rank path is exercised on a single GPU), HOST (caller-supplied exchange on host buffers).
#include "engine.h"
#include <chrono>
#include <cstdlib>
#include "meso_device.h"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <rccl/rccl.h>

namespace meso {

// MESO_RCCL_TIMEOUT (seconds) bounds the two waits of a communicator's first use; unset, empty, unparsable or not positive: 180
static double rccl_timeout_s()
{
    const char *te = getenv("MESO_RCCL_TIMEOUT");
    const double v = (te && *te) ? atof(te) : 0.0;
    return v > 0.0 ? v : 180.0;
}


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

// ------------------------------------------------------------------------------------------------
// in-process transport: ranks are Engine objects driven by different host threads
// ------------------------------------------------------------------------------------------------
struct LocalPost {
    int npeer = 0;
    const int *peer = nullptr;
    void *const *sbuf = nullptr;
    const size_t *sbytes = nullptr;
    double value = 0.0;
};

struct LocalGroup {
    std::mutex m;
    std::condition_variable cv;
    int n = 0, arrived = 0;
    int joined = 0;          // contexts that have taken their place (a full group is never joined again)
    long gen = 0;
    std::vector<LocalPost> post;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        long g = gen;
        if (++arrived == n) { arrived = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

static std::mutex g_groups_mutex;
static std::map<long, LocalGroup *> g_groups;

static LocalGroup *local_group(long id, int n)
{
    std::lock_guard<std::mutex> lk(g_groups_mutex);
    // a group id names the NEXT n contexts that ask for it: once a group is complete a later run with the same id (or one that
    // follows a run which ended with a rank stuck in a barrier) starts a fresh group instead of inheriting stale barrier state
    // (complete groups stay allocated for their members; they are a few hundred bytes)
    LocalGroup *&g = g_groups[id];
    if (!g || g->joined >= g->n || g->n != n) { g = new LocalGroup; g->n = n; g->post.resize(n); }
    g->joined++;
    return g;
}

int comm_unique_id(void *uid, size_t bytes)
{
    if (bytes < sizeof(ncclUniqueId)) return 1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    memcpy(uid, &id, sizeof id);
    return 0;
}

// brick procgrid with minimal surface for the given box (what Comm::set_procs / procmap.cpp picks)
void decomp_procgrid(int nranks, const double *prd, int *pg)
{
    double best = 1e300;
    pg[0] = nranks; pg[1] = 1; pg[2] = 1;
    for (int a = 1; a <= nranks; a++) {
        if (nranks % a) continue;
        for (int b = 1; b <= nranks / a; b++) {
            if ((nranks / a) % b) continue;
            int c = nranks / a / b;
            double surf = prd[0] * prd[1] / (a * b) + prd[0] * prd[2] / (a * c) + prd[1] * prd[2] / (b * c);
            if (surf < best - 1e-12) { best = surf; pg[0] = a; pg[1] = b; pg[2] = c; }
        }
    }
}

// The decomposition a rank works with (Comm::setup / Domain::set_local_box, src/comm.cpp, src/domain.cpp): its sub-box, the
// slabs within cutghost of the faces that have a neighbour, and for each of the 26 directions (dir = (sx+1) + 3(sy+1) + 9(sz+1)) the
// rank that owns the neighbouring sub-box, the periodic shift an atom gets on the way there and the centre of that sub-box (the
// origin of the receiver's fp32 coordinates).  Host only; returns 1 when a sub-box is thinner than the ghost cutoff.
int decomp_plan(const double *boxlo, const double *boxhi, const int *periodic, const int *procgrid, const int *myloc, double cutghost,
                double *sublo, double *subhi, double *slab_lo, double *slab_hi, int *peer27, int *active27, double *shift27,
                double *center27)
{
    const double BIG = 1.0e20;
    double prd[3];
    for (int d = 0; d < 3; d++) prd[d] = boxhi[d] - boxlo[d];
    for (int d = 0; d < 3; d++) {
        sublo[d] = boxlo[d] + prd[d] * myloc[d] / procgrid[d];
        subhi[d] = (myloc[d] == procgrid[d] - 1) ? boxhi[d] : boxlo[d] + prd[d] * (myloc[d] + 1) / procgrid[d];
        if (subhi[d] - sublo[d] < cutghost) return 1;
        bool has_lo = periodic[d] || myloc[d] > 0, has_hi = periodic[d] || myloc[d] < procgrid[d] - 1;
        slab_lo[d] = has_lo ? sublo[d] + cutghost : -BIG;
        slab_hi[d] = has_hi ? subhi[d] - cutghost : BIG;
    }
    for (int dir = 0; dir < 27; dir++) {
        int s[3] = {dir % 3 - 1, (dir / 3) % 3 - 1, dir / 9 - 1};
        bool active = dir != 13;
        int loc[3];
        for (int d = 0; d < 3; d++) {
            shift27[3 * dir + d] = 0.0;
            loc[d] = myloc[d] + s[d];
            if (loc[d] < 0 || loc[d] >= procgrid[d]) {
                if (!periodic[d]) active = false;
                shift27[3 * dir + d] = -s[d] * prd[d];
                loc[d] = (loc[d] + procgrid[d]) % procgrid[d];
            }
            double lo = boxlo[d] + prd[d] * loc[d] / procgrid[d];
            double hi = (loc[d] == procgrid[d] - 1) ? boxhi[d] : boxlo[d] + prd[d] * (loc[d] + 1) / procgrid[d];
            center27[3 * dir + d] = 0.5 * (hi + lo);
        }
        active27[dir] = active ? 1 : 0;
        peer27[dir] = loc[0] + procgrid[0] * (loc[1] + procgrid[1] * loc[2]);
    }
    return 0;
}

int Engine::comm_init(int nr, int rk, const int *pg, int tr, const void *uid, size_t uid_bytes)
{
    if (nr < 1 || rk < 0 || rk >= nr) return fail(1, "Invalid rank layout");
    if (pg[0] < 1 || pg[1] < 1 || pg[2] < 1 || pg[0] * pg[1] * pg[2] != nr) return fail(1, "Bad grid of processors");
    if (nlocal > 0) return fail(3, "comm_init must precede atom creation");
    nranks = nr; rank = rk; transport = nr == 1 ? 0 : tr;
    for (int d = 0; d < 3; d++) procgrid[d] = pg[d];
    myloc[0] = rk % pg[0]; myloc[1] = (rk / pg[0]) % pg[1]; myloc[2] = rk / (pg[0] * pg[1]);
    params_ready = false;
    // one rank needs no transport; MESO_FORCE_RCCL=1 still creates the communicator (library coexistence test)
    if (nr == 1 && !(transport == 1 && getenv("MESO_FORCE_RCCL"))) return 0;
    if (transport == 1) {
        if (!uid || uid_bytes < sizeof(ncclUniqueId)) return fail(1, "RCCL transport needs the 128-byte unique id of rank 0");
        ncclUniqueId id;
        memcpy(&id, uid, sizeof id);
        HIPCHK(hipSetDevice(device));
        // ncclCommInitRank blocks until every rank has joined: a rank that never arrives (a crashed peer, a wrong unique id) would
        // hang the job silently.  The call runs on a helper thread and is given MESO_RCCL_TIMEOUT seconds (default 180); past that
        // the rank reports itself and its place in the grid and the caller ends the process with a non-zero status (the helper
        // thread cannot be cancelled: a retry is a fresh process)
        struct Join { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t rc = ncclSuccess; ncclComm_t c = nullptr; };
        auto join = std::make_shared<Join>();
        const int dev = device;
        std::thread([join, nr, id, rk, dev]() {
            (void)hipSetDevice(dev);
            ncclComm_t c = nullptr;
            const ncclResult_t rc = ncclCommInitRank(&c, nr, id, rk);
            std::lock_guard<std::mutex> lk(join->mu);
            join->rc = rc; join->c = c; join->done = true;
            join->cv.notify_all();
        }).detach();
        {
            const double limit = rccl_timeout_s();
            std::unique_lock<std::mutex> lk(join->mu);
            if (!join->cv.wait_for(lk, std::chrono::duration<double>(limit), [&] { return join->done; })) {
                char msg[256];
                snprintf(msg, sizeof msg, "rank %d of %d (grid cell %d %d %d, device %d): ncclCommInitRank did not return within %.0f s - a peer "
                         "has not joined the communicator", rk, nr, myloc[0], myloc[1], myloc[2], device, limit);
                return fail(5, msg);
            }
            if (join->rc != ncclSuccess) {
                char msg[200];
                snprintf(msg, sizeof msg, "rank %d of %d: ncclCommInitRank failed (%s)", rk, nr, ncclGetErrorString(join->rc));
                return fail(5, msg);
            }
        }
        nccl = (ncclComm *)join->c;
        rccl_first_done = false;
    } else if (transport == 3) {
        long gid = 0;
        if (uid && uid_bytes >= sizeof(long)) memcpy(&gid, uid, sizeof(long));
        local = local_group(gid, nr);
    } else if (transport == 2) {
        if (!host_exchange) return fail(3, "HOST transport needs meso_comm_set_host_exchange first");
    } else {
        return fail(1, "Unknown transport");
    }
    return 0;
}

int Engine::comm_count(int *n)
{
    *n = nranks;
    if (transport == 1 && nccl) {
        int c = 0;
        if (ncclCommCount((ncclComm_t)nccl, &c) != ncclSuccess) return fail(5, "ncclCommCount failed");
        *n = c;
    }
    return 0;
}

void Engine::comm_free()
{
    if (nccl) ncclCommDestroy((ncclComm_t)nccl);
    nccl = nullptr;
}

// ------------------------------------------------------------------------------------------------
// exchange primitive: device buffers, one message per peer each way
// ------------------------------------------------------------------------------------------------
// rbuf2 != null: every message is two halves of equal size (the per-step ghost refresh: coordinates, then velocities) and its
// second half is delivered to rbuf2[k] - straight into the two merged arrays, no scatter kernel behind the exchange.  With RCCL
// the halves travel as two send/recv pairs of the same group (matched in order per peer).
// The pieces of one message: the whole of it, or - rbuf2 given: the per-step refresh received straight into the merged arrays - its two
// halves (coordinates, velocities), which land in two different arrays on the receiving side and therefore travel as two transfers.
// ONE definition for every transport: the in-process and host transports copy exactly the pieces RCCL sends and receives, so the
// multi-rank tests (which cannot run RCCL with two ranks on the one GPU of the test box) exercise the same sizes and offsets.
struct XPiece { void *p; size_t n; };
static inline int xchg_pieces(void *buf, void *buf2, size_t bytes, bool split, XPiece out[2])
{
    if (!bytes) return 0;
    if (!split) { out[0] = {buf, bytes}; return 1; }
    out[0] = {buf, bytes / 2};
    out[1] = {buf2 ? buf2 : (void *)((char *)buf + bytes / 2), bytes / 2};
    return 2;
}

int Engine::xchg(int np, const int *peer, void *const *sbuf, const size_t *sbytes, void *const *rbuf,
                 const size_t *rbytes, void *const *rbuf2)
{
    hipStream_t stream = xs ? xs : this->stream;   // exchange stream (the side stream during overlapped refreshes)
    const bool split = rbuf2 != nullptr;
    if (transport == 1) {
        ncclComm_t c = (ncclComm_t)nccl;
        bool any = false;
        for (int k = 0; k < np; k++) {
            if (peer[k] == rank) {
                XPiece sp[2], rp[2];
                const int ns = xchg_pieces(sbuf[k], nullptr, rbytes[k], split, sp), nr = xchg_pieces(rbuf[k], split ? rbuf2[k] : nullptr, rbytes[k], split, rp);
                for (int q = 0; q < nr && q < ns; q++) HIPCHK(hipMemcpyAsync(rp[q].p, sp[q].p, rp[q].n, hipMemcpyDeviceToDevice, stream));
            } else if (sbytes[k] || rbytes[k]) any = true;
        }
        if (any) {
            // (option profile: the group between two events on the exchange stream - what a kernel trace shows as the RCCL kernel)
            XchgEvent xe;
            const bool timed = profiling;
            if (timed) {
                if (hipEventCreate(&xe.a) != hipSuccess || hipEventCreate(&xe.b) != hipSuccess) return fail(2, "hipEventCreate failed");
                xe.what = xchg_what; xe.bytes = 0;
                for (int k = 0; k < np; k++) if (peer[k] != rank) xe.bytes += (double)sbytes[k];
                HIPCHK(hipEventRecord(xe.a, stream));
            }
            if (ncclGroupStart() != ncclSuccess) return fail(5, "ncclGroupStart failed");
            for (int k = 0; k < np; k++) {
                if (peer[k] == rank) continue;
                XPiece sp[2], rp[2];
                const int ns = xchg_pieces(sbuf[k], nullptr, sbytes[k], split, sp), nr = xchg_pieces(rbuf[k], split ? rbuf2[k] : nullptr, rbytes[k], split, rp);
                for (int q = 0; q < ns; q++)
                    if (ncclSend(sp[q].p, sp[q].n, ncclChar, peer[k], c, stream) != ncclSuccess) return fail(5, "ncclSend failed");
                for (int q = 0; q < nr; q++)
                    if (ncclRecv(rp[q].p, rp[q].n, ncclChar, peer[k], c, stream) != ncclSuccess) return fail(5, "ncclRecv failed");
            }
            if (ncclGroupEnd() != ncclSuccess) return fail(5, "ncclGroupEnd failed");
            if (timed) {
                HIPCHK(hipEventRecord(xe.b, stream));
                xchg_events.push_back(xe);
                // (a long profiled run: the events of finished groups are booked and destroyed every few hundred groups)
                if (xchg_events.size() >= 512) xchg_events_flush();
            }
            if (!rccl_first_done) {
                // the first group of a communicator sets up the connections to every peer: it is waited for here, for at most
                // MESO_RCCL_TIMEOUT seconds, so that a peer that never posts its side is reported by name instead of hanging the job
                const double limit = rccl_timeout_s();
                const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(limit);
                hipError_t q;
                // (connection set-up takes milliseconds to seconds: the first hundred polls yield, later ones sleep)
                for (int polls = 0; (q = hipStreamQuery(stream)) == hipErrorNotReady && std::chrono::steady_clock::now() < t_end; polls++) {
                    if (polls < 100) std::this_thread::yield();
                    else std::this_thread::sleep_for(std::chrono::microseconds(200));
                }
                if (q == hipErrorNotReady) {
                    std::string peers;
                    for (int k = 0; k < np; k++) if (peer[k] != rank && (sbytes[k] || rbytes[k])) peers += (peers.empty() ? "" : " ") + std::to_string(peer[k]);
                    char msg[320];
                    snprintf(msg, sizeof msg, "rank %d of %d (grid cell %d %d %d): the first RCCL exchange (%s) with peers [%s] did not complete "
                             "within %.0f s", rank, nranks, myloc[0], myloc[1], myloc[2], xchg_what, peers.c_str(), limit);
                    return fail(5, msg);
                }
                if (q != hipSuccess) return fail(2, "HIP error in the first RCCL exchange");
                rccl_first_done = true;
            }
        }
        return 0;
    }
    // host-side account of an exchange (option profile): time until the device has finished what the messages hold, time on the
    // "wire" (for the host and in-process transports that includes waiting for the slowest peer), time until the received
    // bytes are back on the device.  RCCL exchanges are enqueued on the stream and show up in the kernel trace instead.
    typedef std::chrono::steady_clock clk;
    const bool acct = profiling && (transport == 2 || transport == 3);
    clk::time_point t0, t1, t2;
    if (acct) t0 = clk::now();
    auto book = [&](clk::time_point a, clk::time_point b, clk::time_point c, clk::time_point d) {
        XchgStat &st = xchg_stats[xchg_what];
        st.calls++;
        st.ms_device += std::chrono::duration<double, std::milli>(b - a).count();
        st.ms_wire += std::chrono::duration<double, std::milli>(c - b).count();
        st.ms_back += std::chrono::duration<double, std::milli>(d - c).count();
        size_t bytes = 0;
        for (int k = 0; k < np; k++) bytes += sbytes[k];
        st.bytes += (double)bytes;
    };
    if (transport == 3) {
        HIPCHK(hipStreamSynchronize(stream));
        if (acct) t1 = clk::now();
        LocalPost &me = local->post[rank];
        me.npeer = np; me.peer = peer; me.sbuf = sbuf; me.sbytes = sbytes;
        local->barrier();
        for (int k = 0; k < np; k++) {
            if (!rbytes[k]) continue;
            const LocalPost &src = local->post[peer[k]];
            const void *from = nullptr;
            for (int q = 0; q < src.npeer; q++)
                if (src.peer[q] == rank) {
                    from = src.sbuf[q];
                    if (src.sbytes[q] != rbytes[k]) {
                        char msg[160];
                        snprintf(msg, sizeof msg, "local transport: size mismatch in the %s exchange (rank %d sends %zu bytes, rank %d expects %zu)",
                                 xchg_what, peer[k], src.sbytes[q], rank, rbytes[k]);
                        return fail(5, msg);
                    }
                }
            if (!from) return fail(5, "local transport: peer did not post a message");
            XPiece sp[2], rp[2];
            const int ns = xchg_pieces(const_cast<void *>(from), nullptr, rbytes[k], split, sp), nr = xchg_pieces(rbuf[k], split ? rbuf2[k] : nullptr, rbytes[k], split, rp);
            if (ns != nr) return fail(5, "local transport: a message and its receive are cut differently");
            for (int q = 0; q < nr; q++) HIPCHK(hipMemcpyAsync(rp[q].p, sp[q].p, rp[q].n, hipMemcpyDeviceToDevice, stream));
        }
        if (acct) t2 = clk::now();
        HIPCHK(hipStreamSynchronize(stream));
        local->barrier();
        if (acct) book(t0, t1, t2, clk::now());
        return 0;
    }
    if (transport == 2) {
        std::vector<std::vector<char>> hs(np), hr(np);
        std::vector<const void *> sp(np);
        std::vector<void *> rp(np);
        for (int k = 0; k < np; k++) {
            hs[k].resize(sbytes[k]); hr[k].resize(rbytes[k]);
            if (sbytes[k]) HIPCHK(hipMemcpyAsync(hs[k].data(), sbuf[k], sbytes[k], hipMemcpyDeviceToHost, stream));
            sp[k] = hs[k].data(); rp[k] = hr[k].data();
        }
        HIPCHK(hipStreamSynchronize(stream));
        if (acct) t1 = clk::now();
        if (host_exchange(host_exchange_user, np, peer, sp.data(), sbytes, rp.data(), rbytes)) return fail(5, "host exchange failed");
        if (acct) t2 = clk::now();
        for (int k = 0; k < np; k++) {
            if (!rbytes[k]) continue;
            XPiece sp[2], rp[2];
            const int ns = xchg_pieces(hr[k].data(), nullptr, rbytes[k], split, sp), nr = xchg_pieces(rbuf[k], split ? rbuf2[k] : nullptr, rbytes[k], split, rp);
            for (int q = 0; q < nr && q < ns; q++) HIPCHK(hipMemcpyAsync(rp[q].p, sp[q].p, rp[q].n, hipMemcpyHostToDevice, stream));
        }
        HIPCHK(hipStreamSynchronize(stream));
        if (acct) book(t0, t1, t2, clk::now());
        return 0;
    }
    return fail(5, "exchange without a transport");
}

// "what calls ms_device ms_wire ms_back bytes" per kind of exchange, one per line (meso_xchg_stats)
void Engine::xchg_events_flush()
{
    for (auto &xe : xchg_events) {
        float ms = 0.f;
        if (hipEventSynchronize(xe.b) == hipSuccess && hipEventElapsedTime(&ms, xe.a, xe.b) == hipSuccess) {
            XchgStat &st = xchg_stats[xe.what];
            st.calls++;
            st.ms_wire += ms;          // on the device: the grouped sends and receives, including the wait for the slowest peer
            st.bytes += xe.bytes;
        }
        (void)hipEventDestroy(xe.a); (void)hipEventDestroy(xe.b);
    }
    xchg_events.clear();
}

std::string Engine::xchg_report()
{
    xchg_events_flush();
    std::string out;
    char buf[256];
    for (const auto &kv : xchg_stats) {
        snprintf(buf, sizeof buf, "%s|%ld|%.6f|%.6f|%.6f|%.0f\n", kv.first.c_str(), kv.second.calls, kv.second.ms_device, kv.second.ms_wire,
                 kv.second.ms_back, kv.second.bytes);
        out += buf;
    }
    return out;
}

double Engine::reduce_global_sum(double v)
{
    if (nranks == 1) return v;
    if (transport == 3) {
        local->post[rank].value = v;
        local->barrier();
        double s = 0.0;
        for (int r = 0; r < nranks; r++) s += local->post[r].value;
        local->barrier();
        return s;
    }
    // RCCL / HOST: all-gather through the pairwise exchange on a small device buffer (thermo steps only)
    std::vector<int> pr;
    for (int r = 0; r < nranks; r++) if (r != rank) pr.push_back(r);
    int np = (int)pr.size();
    double *d = d_scalar + 2;   // [2] mine, [3..] theirs
    if (np + 3 > 16) {
        // d_scalar holds 16 doubles: fall back to a ring of single exchanges for big groups
        double s = v;
        for (int r : pr) {
            (void)hipMemcpyAsync(d, &v, sizeof(double), hipMemcpyHostToDevice, stream);
            void *sb = d, *rb = d + 1;
            size_t nb = sizeof(double);
            int p = r;
            if (xchg(1, &p, &sb, &nb, &rb, &nb)) return s;
            double t = 0.0;
            (void)hipMemcpyAsync(&t, d + 1, sizeof(double), hipMemcpyDeviceToHost, stream);
            (void)hipStreamSynchronize(stream);
            s += t;
        }
        return s;
    }
    (void)hipMemcpyAsync(d, &v, sizeof(double), hipMemcpyHostToDevice, stream);
    std::vector<void *> sb(np, d), rb(np);
    std::vector<size_t> nb(np, sizeof(double));
    for (int k = 0; k < np; k++) rb[k] = d + 1 + k;
    if (xchg(np, pr.data(), sb.data(), nb.data(), rb.data(), nb.data())) return v;
    std::vector<double> h(np);
    (void)hipMemcpyAsync(h.data(), d + 1, np * sizeof(double), hipMemcpyDeviceToHost, stream);
    (void)hipStreamSynchronize(stream);
    double s = v;
    for (double t : h) s += t;
    return s;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
struct Decomp {
    double boxlo[3], boxhi[3], prd[3];
    int pg[3], myloc[3];
};

__host__ __device__ inline double sub_bound(const Decomp &D, int d, int l)
{
    return l >= D.pg[d] ? D.boxhi[d] : D.boxlo[d] + D.prd[d] * l / D.pg[d];
}

// owner location in dim d of a coordinate inside [boxlo, boxhi)
__host__ __device__ inline int owner_loc(const Decomp &D, int d, double x)
{
    int o = (int)((x - D.boxlo[d]) / D.prd[d] * D.pg[d]);
    o = o < 0 ? 0 : (o > D.pg[d] - 1 ? D.pg[d] - 1 : o);
    while (o > 0 && x < sub_bound(D, d, o)) o--;
    while (o < D.pg[d] - 1 && x >= sub_bound(D, d, o + 1)) o++;
    return o;
}

// direction code (0..26, 13 = stays) of the rank that owns each atom after the PBC wrap
__global__ void __launch_bounds__(256) k_migrate_code(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, Decomp D, int n,
                                                      int *__restrict__ code, int *__restrict__ lost)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double c[3] = {x[i], y[i], z[i]};
    int s[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int o = owner_loc(D, d, c[d]);
        int t = o - D.myloc[d];
        if (t > 1) t -= D.pg[d];
        if (t < -1) t += D.pg[d];
        if (t > 1 || t < -1) { atomicAdd(lost, 1); t = 0; }
        s[d] = t;
    }
    code[i] = (s[0] + 1) + 3 * (s[1] + 1) + 9 * (s[2] + 1);
}

// ms doubles per migrant: x,y,z,vx,vy,vz,(tag,type),(mask,image) [+ nbond, nspecial, bond tags/types, special tags]
#define MIG_HEAD 11      // doubles of a migration record before the topology words: x, v, (tag,type), (mask,image), f
__device__ inline void pack_migrant(const AtomSoA &a, int j, int ms, double *o)
{
    o[0] = a.x[0][j]; o[1] = a.x[1][j]; o[2] = a.x[2][j];
    o[3] = a.v[0][j]; o[4] = a.v[1][j]; o[5] = a.v[2][j];
    int2 p = make_int2(a.tag[j], a.type[j]), r = make_int2(a.mask[j], a.image[j]);
    o[6] = *reinterpret_cast<double *>(&p);
    o[7] = *reinterpret_cast<double *>(&r);
    o[8] = a.f[0][j]; o[9] = a.f[1][j]; o[10] = a.f[2][j];
    if (a.bpa > 0 || a.msp > 0) {
        int *t = reinterpret_cast<int *>(o + MIG_HEAD);
        t[0] = a.nbond[j]; t[1] = a.nspecial[j];
        for (int b = 0; b < a.bpa; b++) { t[2 + 2 * b] = a.bond_tag[(size_t)j * a.bpa + b]; t[3 + 2 * b] = a.bond_type[(size_t)j * a.bpa + b]; }
        for (int s = 0; s < a.msp; s++) t[2 + 2 * a.bpa + s] = a.special[(size_t)j * a.msp + s];
        if (a.apa > 0) {
            int *u = t + 2 + 2 * a.bpa + a.msp;
            u[0] = a.nangle[j];
            for (int q = 0; q < 4 * a.apa; q++) u[1 + q] = a.angle_tag[(size_t)j * 4 * a.apa + q];
        }
    }
}

__global__ void __launch_bounds__(256) k_pack_migrate(AtomSoA a, const int *__restrict__ list, const int *__restrict__ dir_start,
                                                      const int *__restrict__ dir_dst, int n0, int n, int ms,
                                                      double *__restrict__ buf)
{
    // entries [n0, n0+n) of the direction-major list, skipping the stay segment handled by the caller
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    int k = n0 + q;
    int d = 0;
    for (int t = 1; t < 27; t++) d += (k >= dir_start[t]) ? 1 : 0;
    if (dir_dst[d] < 0) return;                 // (resend of single messages: the other directions go nowhere)
    int dst = dir_dst[d] + (k - dir_start[d]);
    pack_migrant(a, list[k], ms, buf + (size_t)ms * dst);
}

__device__ inline void unpack_migrant(AtomSoA &a, int i, const double *o, const double *__restrict__ mass_type)
{
    a.x[0][i] = o[0]; a.x[1][i] = o[1]; a.x[2][i] = o[2];
    a.v[0][i] = o[3]; a.v[1][i] = o[4]; a.v[2][i] = o[5];
    a.f[0][i] = o[8]; a.f[1][i] = o[9]; a.f[2][i] = o[10];
    double t6 = o[6], t7 = o[7];
    int2 p = *reinterpret_cast<int2 *>(&t6), r = *reinterpret_cast<int2 *>(&t7);
    a.tag[i] = p.x; a.type[i] = p.y; a.mask[i] = r.x; a.image[i] = r.y;
    a.mass[i] = mass_type[p.y];
    if (a.bpa > 0 || a.msp > 0) {
        const int *t = reinterpret_cast<const int *>(o + MIG_HEAD);
        a.nbond[i] = t[0]; a.nspecial[i] = t[1];
        for (int b = 0; b < a.bpa; b++) { a.bond_tag[(size_t)i * a.bpa + b] = t[2 + 2 * b]; a.bond_type[(size_t)i * a.bpa + b] = t[3 + 2 * b]; }
        for (int s = 0; s < a.msp; s++) a.special[(size_t)i * a.msp + s] = t[2 + 2 * a.bpa + s];
        if (a.apa > 0) {
            const int *u = t + 2 + 2 * a.bpa + a.msp;
            a.nangle[i] = u[0];
            for (int q = 0; q < 4 * a.apa; q++) a.angle_tag[(size_t)i * 4 * a.apa + q] = u[1 + q];
        }
    }
}

__global__ void __launch_bounds__(256) k_unpack_migrate(AtomSoA a, const double *__restrict__ buf, const double *__restrict__ mass_type,
                                                        int base, int n, int ms)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    unpack_migrant(a, base + q, buf + (size_t)ms * q, mass_type);
}

// ---- migration with the counts in band (same scheme as the fixed-capacity border messages further down: header of 27
// per-direction counts + capacity derived from the previous rebuild's count).  Migration still has its host round trip - the
// local atom count changes with it - but no separate count exchange any more; a message that outgrows its capacity is simply sent
// again, exactly, in a second exchange that costs nothing when nobody needs it (zero-size messages are not posted).
#define MIG_HDR_DOUBLES 16
struct MigPlan {
    int np;
    int pidx[27];                  // peer index of each direction (-1: stays / nothing goes that way)
    int cap_s[26], cap_r[26];      // atoms
    long base_s[26], base_r[26];   // first double of each peer's block in stage_send / stage_recv
};

// header of every peer's block + slot of each direction's segment inside its peer's payload (dst[27])
__global__ void __launch_bounds__(64) k_mig_hdr(const int *__restrict__ ds, MigPlan P, int *__restrict__ dst, double *__restrict__ stage_send)
{
    __shared__ int cnt[27];
    const int t = threadIdx.x;
    if (t < 27) cnt[t] = ds[t + 1] - ds[t];
    __syncthreads();
    if (t < P.np) {
        int fill = 0;
        int *hdr = reinterpret_cast<int *>(stage_send + P.base_s[t]);
        for (int d = 0; d < 27; d++) {
            const bool mine = P.pidx[d] == t;
            hdr[d] = mine ? cnt[d] : 0;
            if (mine) { dst[d] = fill; fill += cnt[d]; }
        }
    }
}

__global__ void __launch_bounds__(256) k_pack_migrate_fixed(AtomSoA a, const int *__restrict__ list, const int *__restrict__ dir_start,
                                                            MigPlan P, const int *__restrict__ dst_dev, int ms, double *__restrict__ stage_send)
{
    __shared__ int ds[28], dst[27];
    if (threadIdx.x < 28) ds[threadIdx.x] = dir_start[threadIdx.x];
    if (threadIdx.x < 27) dst[threadIdx.x] = dst_dev[threadIdx.x];
    __syncthreads();
    // the list holds every local atom, direction-major; the segment of direction 13 are the atoms that stay
    const int nleave = ds[13] + (ds[27] - ds[14]);
    for (int kk = blockDim.x * blockIdx.x + threadIdx.x; kk < nleave; kk += gridDim.x * blockDim.x) {
        const int k = kk < ds[13] ? kk : kk - ds[13] + ds[14];
        const int d = dir_of_entry(ds, k), p = P.pidx[d];
        if (p < 0) continue;
        const int q = dst[d] + (k - ds[d]);
        if (q >= P.cap_s[p]) continue;                      // (the whole message follows in the second exchange)
        pack_migrant(a, list[k], ms, stage_send + P.base_s[p] + MIG_HDR_DOUBLES + (size_t)ms * q);
    }
}

// ---- the migration's sending side in two launches (code, count, scan x 2, direction starts, fill, header, pack: 8 before) ----
// Few atoms leave (a few hundred of 131 k per rebuild): the kernel that finds them appends them to their direction's list with an
// atomic, and the packing kernel restores the order of the counting chain by ranking every leaver among its direction's list
// (index order), so messages - and with them the arrival order on the peer - are what they were, run after run.  A list holds
// lc entries; a direction that overflows it belongs to a message that overflows its capacity (lc >= every capacity) and is sent
// again exactly, from the lists of the counting chain, which are then built on demand (build_mig_lists).
// (pb.on: MesoDomain::pbc of the atom first - the k_pbc launch of the rebuild; lost atoms are counted in cnt[28], which
// k_mig_pack hands to the flag word the host reads and k_mig_read_hdr clears with the counters: no memset launch)
struct PbcArgs { int on; double lo[3], hi[3]; int per[3]; int *image; };
__global__ void __launch_bounds__(256) k_mig_scan(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                                  Decomp D, int n, int *__restrict__ code, int *__restrict__ cnt,
                                                  int *__restrict__ lst, int lc, PbcArgs pb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int *lost = cnt + 28;
    double c[3] = {x[i], y[i], z[i]};
    if (pb.on) {
        const int img = pb.image[i];
        int im[3] = {img & 1023, (img >> 10) & 1023, img >> 20};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (!pb.per[d]) continue;
            const double p = pb.hi[d] - pb.lo[d];
            if (c[d] < pb.lo[d]) { c[d] += p; im[d] = (im[d] - 1) & 1023; }
            if (c[d] >= pb.hi[d]) { c[d] -= p; c[d] = fmax(c[d], pb.lo[d]); im[d] = (im[d] + 1) & 1023; }
        }
        x[i] = c[0]; y[i] = c[1]; z[i] = c[2];
        pb.image[i] = im[0] | (im[1] << 10) | (im[2] << 20);
    }
    int s[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int o = owner_loc(D, d, c[d]);
        int t = o - D.myloc[d];
        if (t > 1) t -= D.pg[d];
        if (t < -1) t += D.pg[d];
        if (t > 1 || t < -1) { atomicAdd(lost, 1); t = 0; }
        s[d] = t;
    }
    const int cd = (s[0] + 1) + 3 * (s[1] + 1) + 9 * (s[2] + 1);
    code[i] = cd;
    if (cd != 13) {
        const int pos = atomicAdd(&cnt[cd], 1);
        if (pos < lc) lst[(size_t)cd * lc + pos] = i;
    }
}

// grid (blocks, 27): direction blockIdx.y.  Block (0, 0) also writes the peers' headers, the direction slots (dst_dev) and the
// direction starts the host reads after the exchange (the segment of direction 13 = the atoms that stay)
__global__ void __launch_bounds__(256) k_mig_pack(AtomSoA a, const int *__restrict__ cnt_dev, const int *__restrict__ lst, int lc, int n,
                                                  MigPlan P, int *__restrict__ dst_dev, int *__restrict__ dir_start, int ms,
                                                  double *__restrict__ stage_send, int *__restrict__ lost_out)
{
    __shared__ int cnt[27], dst[27];
    const int t = threadIdx.x;
    if (t < 27) cnt[t] = t == 13 ? 0 : cnt_dev[t];
    __syncthreads();
    if (t < P.np) {
        int fill = 0;
        int *hdr = reinterpret_cast<int *>(stage_send + P.base_s[t]);
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;
        for (int d = 0; d < 27; d++) {
            const bool mine = P.pidx[d] == t;
            if (first) hdr[d] = mine ? cnt[d] : 0;
            if (mine) { dst[d] = fill; fill += cnt[d]; }
        }
    }
    if (t < 27 && P.pidx[t] < 0) dst[t] = 0;
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (t < 27) dst_dev[t] = dst[t];
        if (t == 0) {
            *lost_out = cnt_dev[28];
            int leave = 0;
            for (int d = 0; d < 27; d++) leave += cnt[d];
            int run = 0;
            for (int d = 0; d < 27; d++) { dir_start[d] = run; run += d == 13 ? n - leave : cnt[d]; }
            dir_start[27] = run;
        }
    }
    const int d = blockIdx.y, p = P.pidx[d];
    if (d == 13 || p < 0) return;
    const int m = min(cnt[d], lc);
    const int *l = lst + (size_t)d * lc;
    // (the direction's list in LDS when it fits: the ranking loop then reads one LDS word per step, the same for every lane)
    __shared__ int ll[2048];
    const bool staged = m <= 2048;
    if (staged) {
        for (int e = t; e < m; e += blockDim.x) ll[e] = l[e];
        __syncthreads();
        l = ll;
    }
    for (int k = blockDim.x * blockIdx.x + t; k < m; k += gridDim.x * blockDim.x) {
        const int i = l[k];
        int rank = 0;
        for (int e = 0; e < m; e++) rank += l[e] < i ? 1 : 0;
        const int q = dst[d] + rank;
        if (q >= P.cap_s[p]) continue;                      // (the whole message follows in the second exchange)
        pack_migrant(a, i, ms, stage_send + P.base_s[p] + MIG_HDR_DOUBLES + (size_t)ms * q);
    }
}

// what each peer announces in its header: report[t] = migrants from peer t (also beyond the capacity)
// (clear: the leavers' counters of k_mig_scan, for the next rebuild)
__global__ void __launch_bounds__(64) k_mig_read_hdr(const double *__restrict__ stage_recv, MigPlan P, int *__restrict__ report, int *__restrict__ clear)
{
    const int t = threadIdx.x;
    if (clear && t < 32) clear[t] = 0;
    if (t < P.np) {
        const int *hdr = reinterpret_cast<const int *>(stage_recv + P.base_r[t]);
        int n = 0;
        for (int d = 0; d < 27; d++) n += hdr[d];
        report[t] = n;
    }
}

struct MigSources {                // where each peer's migrants lie after the exchange(s)
    int np;
    int gbase[27];                 // first arrival of each peer, gbase[np] = total
    long src[26];                  // first double of the peer's records (fixed block or the exact resend)
    int second[26];                // 1: in the buffer of the second exchange
};

__global__ void __launch_bounds__(256) k_unpack_migrate_from(AtomSoA a, const double *__restrict__ first, const double *__restrict__ second,
                                                             MigSources S, const double *__restrict__ mass_type, int base, int ms)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S.gbase[S.np]) return;
    int p = 0;
    for (int t = 1; t < S.np; t++) p += (g >= S.gbase[t]) ? 1 : 0;
    const double *o = (S.second[p] ? second : first) + S.src[p] + (size_t)ms * (g - S.gbase[p]);
    unpack_migrant(a, base + g, o, mass_type);
}

struct DirTab {
    int start[28];   // segment of each direction in the direction-major send list
    int dst[27];     // first slot of the direction's segment in the peer-major staging
    int vofs[27];    // (forward) distance from the coordinate block to the velocity block of that peer
    double shift[27][3];
    double center[27][3];
};

__device__ inline int dir_of(const DirTab &T, int k)
{
    int d = 0;
#pragma unroll
    for (int t = 1; t < 27; t++) d += (k >= T.start[t]) ? 1 : 0;
    return d;
}

// 8 doubles per new ghost: x,y,z (shift applied), vx,vy,vz, (tag,type), (mask,0) [pack_border_vel]: with the velocities on board
// the receiver builds the ghosts' merged float4 pairs itself (k_merge_ghosts) and a rebuild needs no ghost refresh exchange
#define BORDER_DOUBLES 8
__global__ void __launch_bounds__(256) k_pack_border_multi(AtomSoA a, const int *__restrict__ list, int n, DirTab T,
                                                           double *__restrict__ buf)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int d = dir_of(T, k), j = list[k];
    double *o = buf + BORDER_DOUBLES * (size_t)(T.dst[d] + (k - T.start[d]));
    o[0] = a.x[0][j] + T.shift[d][0];
    o[1] = a.x[1][j] + T.shift[d][1];
    o[2] = a.x[2][j] + T.shift[d][2];
    o[3] = a.v[0][j]; o[4] = a.v[1][j]; o[5] = a.v[2][j];
    int2 p = make_int2(a.tag[j], a.type[j]), r = make_int2(a.mask[j], 0);
    o[6] = *reinterpret_cast<double *>(&p);
    o[7] = *reinterpret_cast<double *>(&r);
}

__global__ void __launch_bounds__(256) k_unpack_border(AtomSoA a, const double *__restrict__ buf, int base, int n)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const double *o = buf + BORDER_DOUBLES * (size_t)g;
    int i = base + g;
    a.x[0][i] = o[0]; a.x[1][i] = o[1]; a.x[2][i] = o[2];
    a.v[0][i] = o[3]; a.v[1][i] = o[4]; a.v[2][i] = o[5];
    double t3 = o[6], t4 = o[7];
    int2 p = *reinterpret_cast<int2 *>(&t3), r = *reinterpret_cast<int2 *>(&t4);
    a.tag[i] = p.x; a.type[i] = p.y; a.mask[i] = r.x;
}

// ------------------------------------------------------------------------------------------------
// borders of several ranks WITHOUT a host round trip (option async_counts): fixed-capacity messages with the counts in band
// ------------------------------------------------------------------------------------------------
// A border message to peer p is [header: the 27 per-direction counts as ints][payload: cap_s[p] ghosts of 8 doubles].  The
// capacity is a function of the count the two ranks exchanged at the PREVIOUS rebuild (c + c/4 + 256) - the sender knows what it
// sent, the receiver what it received, so both sides size the message alike without talking, and RCCL gets its host-side sizes
// without the host ever reading this rebuild's counts.  Offsets on both sides are computed on the device (k_border_hdr,
// k_border_unpack_hdr); the counts reach the host through pinned memory behind an event and are read when the next per-step
// ghost refresh needs its tables (Engine::resolve_counts).  A count beyond the capacity is an error reported at the end of
// run(), never a truncated ghost list.
#define MR_HDR_DOUBLES 16
struct MrPlan {
    int np;
    int pidx[27];                  // peer index of each direction (-1: nothing goes that way)
    int cap_s[26], cap_r[26];      // atoms
    long base_s[26], base_r[26];   // first double of each peer's block in stage_send / stage_recv
};
// d_mr (ints): [0..26] slot of each direction's segment inside its peer's payload, [32..57] my count per peer, [64] ghosts
// received, [65..91] first ghost of each peer (arrival order), [65 + np] = [64]
#define MR_NGHOST 64
#define MR_GBASE 65

__global__ void __launch_bounds__(64) k_border_hdr(const int *__restrict__ ds, MrPlan P, int *__restrict__ d_mr, double *__restrict__ stage_send,
                                                   int *__restrict__ flags, int *__restrict__ report)
{
    __shared__ int cnt[27];
    const int t = threadIdx.x;
    if (t < 27) cnt[t] = ds[t + 1] - ds[t];
    __syncthreads();
    if (t < P.np) {
        int fill = 0;
        int *hdr = reinterpret_cast<int *>(stage_send + P.base_s[t]);
        for (int d = 0; d < 27; d++) {
            const bool mine = P.pidx[d] == t;
            hdr[d] = mine ? cnt[d] : 0;
            if (mine) { d_mr[d] = fill; fill += cnt[d]; }
        }
        d_mr[32 + t] = fill;
        report[32 + t] = fill;
        if (fill > P.cap_s[t]) flags[0] = 200002;
    }
}

// (the record's last int: Morton code of the RECEIVER's ghost cell - this rank's cell of the atom moved by the direction, the grids
// of two ranks being translates of each other.  The receiver keeps its ghosts in message order and reads the cells' runs off these
// codes: k_unpack_ghost_runs.)
__global__ void __launch_bounds__(256) k_pack_border_fixed(AtomSoA a, const int *__restrict__ list, const int *__restrict__ dir_start,
                                                           MrPlan P, Shift27 sh, const int *__restrict__ d_mr, BinGeom bg, double *__restrict__ stage_send)
{
    __shared__ int ds[28], dst[27];
    if (threadIdx.x < 28) ds[threadIdx.x] = dir_start[threadIdx.x];
    if (threadIdx.x < 27) dst[threadIdx.x] = d_mr[threadIdx.x];
    __syncthreads();
    for (int k = blockDim.x * blockIdx.x + threadIdx.x; k < ds[27]; k += gridDim.x * blockDim.x) {      // (the grid is an estimate)
        const int d = dir_of_entry(ds, k), p = P.pidx[d], j = list[k];
        if (p < 0) continue;
        const int q = dst[d] + (k - ds[d]);
        if (q >= P.cap_s[p]) continue;                      // (flagged by k_border_hdr)
        double *o = stage_send + P.base_s[p] + MR_HDR_DOUBLES + BORDER_DOUBLES * (size_t)q;
        o[0] = a.x[0][j] + sh.s[d][0];
        o[1] = a.x[1][j] + sh.s[d][1];
        o[2] = a.x[2][j] + sh.s[d][2];
        o[3] = a.v[0][j]; o[4] = a.v[1][j]; o[5] = a.v[2][j];
        const double c[3] = {a.x[0][j], a.x[1][j], a.x[2][j]};
        const int sd[3] = {d % 3 - 1, (d / 3) % 3 - 1, d / 9 - 1};
        u32 gq[3];
        for (int k = 0; k < 3; k++) {
            const int b = clampi((int)((c[k] - bg.lo[k]) * bg.bininv[k] + 1), 0, bg.mbin[k]);       // (the binning of k_fr_count)
            gq[k] = (u32)clampi(b - sd[k] * (bg.mbin[k] - 2), 0, bg.mbin[k]);
        }
        int2 pp = make_int2(a.tag[j], a.type[j]), r = make_int2(a.mask[j], (int)interleave3(gq[0], gq[1], gq[2]));
        o[6] = *reinterpret_cast<double *>(&pp);
        o[7] = *reinterpret_cast<double *>(&r);
    }
}

// border lists, message headers and the packed records in ONE launch behind count + scan (k_border_fill, k_border_hdr and
// k_pack_border_fixed were three): every block works out the direction slots from the direction starts (27 additions per peer),
// block 0 also writes the headers and the report; an atom's record goes into the message at the moment its list entry is written
__global__ void __launch_bounds__(256) k_border_fill_pack(AtomSoA a, int beg, int end, Slabs sl, const int *__restrict__ chunk_offset, int nchunk,
                                                          int *__restrict__ sendlist, const int *__restrict__ dir_start, MrPlan P, Shift27 sh,
                                                          int *__restrict__ d_mr, BinGeom bg, double *__restrict__ stage_send,
                                                          int *__restrict__ flags, int *__restrict__ report, int *__restrict__ img_cnt,
                                                          int *__restrict__ img, int *__restrict__ vofs_out, int *__restrict__ zero, int nzero)
{
    // (zero: the ghost-cell counts the receiving kernel of this rebuild starts from - cleared here, ahead of the exchange)
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nzero; k += gridDim.x * blockDim.x) zero[k] = 0;
    __shared__ int wave_tot[27][4];
    __shared__ int ds[28], dst[27], pfill[27], pbase[27];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t < 28) ds[t] = dir_start[t];
    if (t < 27) dst[t] = 0;
    __syncthreads();
    if (t < P.np) {
        int fill = 0;
        int *hdr = reinterpret_cast<int *>(stage_send + P.base_s[t]);
        for (int d = 0; d < 27; d++) {
            const bool mine = P.pidx[d] == t;
            const int c = ds[d + 1] - ds[d];
            if (blockIdx.x == 0) hdr[d] = mine ? c : 0;
            if (mine) { dst[d] = fill; fill += c; }
        }
        pfill[t] = fill;
        if (blockIdx.x == 0) {
            d_mr[32 + t] = fill;
            report[32 + t] = fill;
            if (fill > P.cap_s[t]) flags[0] = 200002;
        }
    }
    const int i = beg + blockIdx.x * 256 + t;
    int fl = 0;
    double c[3] = {0.0, 0.0, 0.0};
    if (i < end) {
        c[0] = a.x[0][i]; c[1] = a.x[1][i]; c[2] = a.x[2][i];
        fl = near_flags(c[0], c[1], c[2], sl.lo, sl.hi);
    }
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        const u64 m = __ballot(dir != 13 && fl && in_dir(fl, dir));
        if (lane == 0) wave_tot[dir][w] = __popcll(m);
    }
    __syncthreads();                        // direction slots, wave totals
    if (blockIdx.x == 0 && t < 27) d_mr[t] = dst[t];
    // img != null: the atom's slots in the PER-STEP refresh messages (peer-major staging of float4 pairs: [coordinates of peer p]
    // [velocities of peer p], as fill_dirtab lays it out) go into its image table - the force kernel's step boundary then writes
    // the refresh itself (nve_boundary_atom) and k_pack_forward_multi is not launched
    if (!img_cnt) img = nullptr;          // (both or neither)
    if (img) {
        if (t == 0) {
            int run = 0;
            for (int q = 0; q < P.np; q++) { pbase[q] = run; run += pfill[q]; }
        }
        __syncthreads();
        if (blockIdx.x == 0 && t < 27) vofs_out[t] = P.pidx[t] >= 0 ? pfill[P.pidx[t]] : 0;
    }
    if (!fl) return;
    int nimg = 0;
    // the atom's part of its records, once
    const double vx = a.v[0][i], vy = a.v[1][i], vz = a.v[2][i];
    const int2 pp = make_int2(a.tag[i], a.type[i]);
    const int mk = a.mask[i];
    int b[3];
    for (int k = 0; k < 3; k++) b[k] = clampi((int)((c[k] - bg.lo[k]) * bg.bininv[k] + 1), 0, bg.mbin[k]);       // (the binning of k_fr_count)
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        if (dir == 13) continue;
        const bool hit = in_dir(fl, dir);
        const u64 m = __ballot(hit);
        if (!hit) continue;
        int base = chunk_offset[dir * nchunk + blockIdx.x];
        for (int k = 0; k < w; k++) base += wave_tot[dir][k];
        const int slot = base + __popcll(m & ((1ULL << lane) - 1ULL));
        sendlist[slot] = i;
        const int p = P.pidx[dir];
        if (p < 0) continue;
        const int q = dst[dir] + (slot - ds[dir]);
        if (img && nimg < 8) img[(size_t)i * 8 + nimg] = (2 * pbase[p] + q) | (dir << 26);
        nimg++;
        if (q >= P.cap_s[p]) continue;                      // (flagged above)
        double *o = stage_send + P.base_s[p] + MR_HDR_DOUBLES + BORDER_DOUBLES * (size_t)q;
        o[0] = c[0] + sh.s[dir][0]; o[1] = c[1] + sh.s[dir][1]; o[2] = c[2] + sh.s[dir][2];
        o[3] = vx; o[4] = vy; o[5] = vz;
        const int sd[3] = {dir % 3 - 1, (dir / 3) % 3 - 1, dir / 9 - 1};
        u32 gq[3];
        for (int k = 0; k < 3; k++) gq[k] = (u32)clampi(b[k] - sd[k] * (bg.mbin[k] - 2), 0, bg.mbin[k]);
        int2 pq = pp, r = make_int2(mk, (int)interleave3(gq[0], gq[1], gq[2]));
        o[6] = *reinterpret_cast<double *>(&pq);
        o[7] = *reinterpret_cast<double *>(&r);
    }
    if (img) {
        img_cnt[i] = nimg;
        if (nimg > 8) flags[0] = 200004;      // (more than 8 directions: a sub-box narrower than two ghost cutoffs - excluded on the host)
    }
}

__global__ void __launch_bounds__(64) k_border_unpack_hdr(const double *__restrict__ stage_recv, MrPlan P, int room, int *__restrict__ d_mr,
                                                          int *__restrict__ flags, int *__restrict__ report)
{
    __shared__ int c[27];
    const int t = threadIdx.x;
    if (t < P.np) {
        const int *hdr = reinterpret_cast<const int *>(stage_recv + P.base_r[t]);
        int n = 0;
        for (int d = 0; d < 27; d++) n += hdr[d];
        if (n > P.cap_r[t] || n < 0) { flags[0] = 200002; n = 0; }
        c[t] = n;
        report[t] = n;
    }
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int p = 0; p < P.np; p++) { d_mr[MR_GBASE + p] = run; run += c[p]; }
        if (run > room) { flags[0] = 200002; run = 0; }
        d_mr[MR_GBASE + P.np] = run;
        d_mr[MR_NGHOST] = run;
        report[31] = run;
    }
}

__global__ void __launch_bounds__(256) k_unpack_border_fixed(AtomSoA a, const double *__restrict__ stage_recv, MrPlan P,
                                                             const int *__restrict__ d_mr, int nlocal)
{
    __shared__ int gb[28];
    if ((int)threadIdx.x <= P.np) gb[threadIdx.x] = d_mr[MR_GBASE + threadIdx.x];
    __syncthreads();
    const int ng = gb[P.np];
    for (int g = blockDim.x * blockIdx.x + threadIdx.x; g < ng; g += gridDim.x * blockDim.x) {
        int p = 0;
        for (int t = 1; t < P.np; t++) p += (g >= gb[t]) ? 1 : 0;
        const double *o = stage_recv + P.base_r[p] + MR_HDR_DOUBLES + BORDER_DOUBLES * (size_t)(g - gb[p]);
        const int i = nlocal + g;
        a.x[0][i] = o[0]; a.x[1][i] = o[1]; a.x[2][i] = o[2];
        a.v[0][i] = o[3]; a.v[1][i] = o[4]; a.v[2][i] = o[5];
        double t3 = o[6], t4 = o[7];
        int2 pp = *reinterpret_cast<int2 *>(&t3), r = *reinterpret_cast<int2 *>(&t4);
        a.tag[i] = pp.x; a.type[i] = pp.y; a.mask[i] = r.x;
    }
}

// The receiving side of a rebuild in ONE kernel (unpack, ghost binning by count / scan / place / order, merged pairs: 8 launches
// before): the ghosts stay in message order - the atoms of one source cell are consecutive in a message (the sender's border
// section is in cell order and the direction lists keep it), and a ghost cell is fed by one source cell of one direction - so a
// ghost cell is a RUN of consecutive ghosts with one code.  The first atom of a run writes (start, count) of its cell (gcnt
// cleared before: cells without ghosts stay at 0; a cell that two runs claim is reported, flag 200003); every atom also writes
// its merged float4 pair (k_merge_ghosts' expressions).  The per-step refresh then scatters without a slot table.
__global__ void __launch_bounds__(256) k_unpack_ghost_runs(AtomSoA a, const double *__restrict__ stage_recv, MrPlan P, const int *__restrict__ d_mr,
                                                           int nlocal, double cx, double cy, double cz, u32 seed, float4 *__restrict__ coord4,
                                                           float4 *__restrict__ veloc4, int *__restrict__ gstart, int *__restrict__ gcnt,
                                                           int M, int *__restrict__ flags)
{
    __shared__ int gb[28];
    if ((int)threadIdx.x <= P.np) gb[threadIdx.x] = d_mr[MR_GBASE + threadIdx.x];
    __syncthreads();
    const int ng = gb[P.np];
    auto record = [&](int g) {
        int p = 0;
        for (int t = 1; t < P.np; t++) p += (g >= gb[t]) ? 1 : 0;
        return stage_recv + P.base_r[p] + MR_HDR_DOUBLES + BORDER_DOUBLES * (size_t)(g - gb[p]);
    };
    auto code_of = [&](int g) { return reinterpret_cast<const int *>(record(g) + 7)[1]; };
    for (int g = blockDim.x * blockIdx.x + threadIdx.x; g < ng; g += gridDim.x * blockDim.x) {
        const double *o = record(g);
        const int i = nlocal + g;
        const double x = o[0], y = o[1], z = o[2], vx = o[3], vy = o[4], vz = o[5];
        a.x[0][i] = x; a.x[1][i] = y; a.x[2][i] = z;
        a.v[0][i] = vx; a.v[1][i] = vy; a.v[2][i] = vz;
        double t3 = o[6], t4 = o[7];
        const int2 pp = *reinterpret_cast<int2 *>(&t3), r = *reinterpret_cast<int2 *>(&t4);
        a.tag[i] = pp.x; a.type[i] = pp.y; a.mask[i] = r.x;
        float4 c;
        c.x = (float)(x - cx); c.y = (float)(y - cy); c.z = (float)(z - cz);
        c.w = __uint_as_float((u32)(pp.y - 1));
        coord4[g] = c;
        float4 v;
        v.x = (float)vx; v.y = (float)vy; v.z = (float)vz;
        v.w = __uint_as_float(signature(seed, pp.x, v.x, v.y, v.z));
        veloc4[g] = v;
        // run heads and lengths from the wave's own lanes where they can be had there (consecutive lanes hold consecutive ghosts): the
        // predecessor's code by shuffle, the next head inside the wave from the ballot of heads; memory only across wave boundaries
        const int code = r.y;
        const int lane = threadIdx.x & 63;
        const int up = __shfl_up(code, 1, 64);
        const bool head = g == 0 || (lane == 0 ? code_of(g - 1) : up) != code;
        const unsigned long long hm = __ballot(head);
        if (head) {
            const unsigned long long later = lane == 63 ? 0ull : hm >> (lane + 1);
            int len;
            if (later) len = __ffsll((long long)later);               // the next head is `len` lanes on
            else {
                // the run reaches the end of the wave's ghosts: the rest of it, if any, lies with the next wave
                const int last = min(g + (63 - lane), ng - 1);        // last ghost of this wave's lanes
                len = last - g + 1;
                while (g + len < ng && code_of(g + len) == code) len++;
            }
            if ((u32)code >= (u32)M) flags[0] = 200003;
            else {
                gstart[code] = g;
                if (atomicAdd(&gcnt[code], len) != 0) flags[0] = 200003;
            }
        }
    }
}

// merged float4 pairs of the ghosts a rebuild has just created, into their Morton(bin) slots: the values the sender's
// k_pack_forward_multi would deliver (the ghost's fp64 position already carries the periodic shift, and the centre the sender
// uses for this rank is this rank's own: same doubles, same roundings)
__global__ void __launch_bounds__(256) k_merge_ghosts(AtomSoA a, int nlocal, int nghost, const int *__restrict__ nghost_dev,
                                                      const int *__restrict__ gslot, double cx, double cy, double cz, u32 seed,
                                                      float4 *__restrict__ coord4, float4 *__restrict__ veloc4)
{
    if (nghost_dev) nghost = *nghost_dev;          // (nghost only sized the grid while the counts are on their way to the host)
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < nghost; g += gridDim.x * blockDim.x) {
    const int i = nlocal + g, out = gslot ? gslot[g] : g;
    float4 c;
    c.x = (float)(a.x[0][i] - cx);
    c.y = (float)(a.x[1][i] - cy);
    c.z = (float)(a.x[2][i] - cz);
    c.w = __uint_as_float((u32)(a.type[i] - 1));
    coord4[out] = c;
    float4 v;
    v.x = (float)a.v[0][i]; v.y = (float)a.v[1][i]; v.z = (float)a.v[2][i];
    v.w = __uint_as_float(signature(seed, a.tag[i], v.x, v.y, v.z));
    veloc4[out] = v;
    }
}

// per-step payload into the peer-major staging: [coords of peer p][velocities of peer p] ...
__global__ void __launch_bounds__(256) k_pack_forward_multi(AtomSoA a, const int *__restrict__ list, int n, DirTab T,
                                                            u32 seed, float4 *__restrict__ stage)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int d = dir_of(T, k), j = list[k];
    int slot = T.dst[d] + (k - T.start[d]);
    float4 c;
    c.x = (float)((a.x[0][j] + T.shift[d][0]) - T.center[d][0]);
    c.y = (float)((a.x[1][j] + T.shift[d][1]) - T.center[d][1]);
    c.z = (float)((a.x[2][j] + T.shift[d][2]) - T.center[d][2]);
    c.w = __uint_as_float((u32)(a.type[j] - 1));
    stage[slot] = c;
    float4 v;
    v.x = (float)a.v[0][j]; v.y = (float)a.v[1][j]; v.z = (float)a.v[2][j];
    v.w = __uint_as_float(signature(seed, a.tag[j], v.x, v.y, v.z));
    stage[slot + T.vofs[d]] = v;
}

struct PeerTab {
    int np;
    int gbase[28];   // first ghost (arrival order) of each peer, gbase[np] = nghost
};

__global__ void __launch_bounds__(256) k_scatter_ghost(const float4 *__restrict__ stage, PeerTab P, const int *__restrict__ gslot,
                                                       int nghost, float4 *__restrict__ coord4, float4 *__restrict__ veloc4)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nghost) return;
    int p = 0;
    for (int t = 1; t < P.np; t++) p += (g >= P.gbase[t]) ? 1 : 0;
    int n_p = P.gbase[p + 1] - P.gbase[p];
    int q = g - P.gbase[p];
    int out = gslot ? gslot[g] : g;
    coord4[out] = stage[2 * P.gbase[p] + q];
    veloc4[out] = stage[2 * P.gbase[p] + n_p + q];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static Decomp make_decomp(const double *lo, const double *hi, const double *prd, const int *pg, const int *loc)
{
    Decomp D;
    for (int d = 0; d < 3; d++) { D.boxlo[d] = lo[d]; D.boxhi[d] = hi[d]; D.prd[d] = prd[d]; D.pg[d] = pg[d]; D.myloc[d] = loc[d]; }
    return D;
}

bool Engine::owns(const double *x) const
{
    Decomp D = make_decomp(boxlo, boxhi, prd, procgrid, myloc);
    for (int d = 0; d < 3; d++) {
        double c = x[d];
        if (periodic[d]) {
            if (c < boxlo[d]) c += prd[d];
            if (c >= boxhi[d]) { c -= prd[d]; if (c < boxlo[d]) c = boxlo[d]; }
        }
        if (c < boxlo[d] || c >= boxhi[d]) {   // outside a non-periodic box: the edge rank keeps it
            if ((c < boxlo[d] && myloc[d] != 0) || (c >= boxhi[d] && myloc[d] != procgrid[d] - 1)) return false;
            continue;
        }
        if (owner_loc(D, d, c) != myloc[d]) return false;
    }
    return true;
}

// unique peers of the 26 directions (increasing rank), and each direction's peer index
void Engine::build_peer_tables()
{
    peers.clear();
    for (int d = 0; d < 27; d++)
        if (d != 13 && send_active[d]) peers.push_back(peer27[d]);
    std::sort(peers.begin(), peers.end());
    peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
    for (int d = 0; d < 27; d++) {
        peer_index[d] = -1;
        if (d == 13 || !send_active[d]) continue;
        peer_index[d] = (int)(std::lower_bound(peers.begin(), peers.end(), peer27[d]) - peers.begin());
    }
}

// per-peer count messages straight from the device-side direction starts: row p holds the sizes of the directions that go to
// peer p (27 ints each; skip_stay: direction 13 is the atoms that stay)
struct PeerIndex27 { int p[27]; };
__global__ void __launch_bounds__(64) k_counts_to_peers(const int *__restrict__ dir_start, PeerIndex27 pi, int np, int skip_stay,
                                                        int *__restrict__ sendv)
{
    for (int k = threadIdx.x; k < np * 27; k += blockDim.x) {
        const int p = k / 27, d = k - 27 * p;
        sendv[k] = (pi.p[d] == p && !(skip_stay && d == 13)) ? dir_start[d + 1] - dir_start[d] : 0;
    }
}

// Sizes of my direction-major segments (d_dir_start, still on the device) -> per-peer counts, exchanged device to device; ONE
// host round trip then delivers my own direction starts (h_ds[28]), the counts the peers announced and whatever else the caller
// queued for the host before (comm_meso.cu:136-137 is the count swap this replaces)
int Engine::exchange_counts(int skip_stay, int *h_ds, std::vector<int> &send_n, std::vector<int> &recv_n,
                            std::vector<int> &recv_dir /* np*27 */)
{
    int np = (int)peers.size();
    int *dbuf = sendlist_aux;   // device scratch: 2 * np * 27 ints
    PeerIndex27 pi;
    for (int d = 0; d < 27; d++) pi.p[d] = d == 13 ? -1 : peer_index[d];
    if (np > 0) hipLaunchKernelGGL(k_counts_to_peers, dim3(1), dim3(64), 0, stream, d_dir_start, pi, np, skip_stay, dbuf);
    std::vector<void *> sb(np), rb(np);
    std::vector<size_t> nb(np, 27 * sizeof(int));
    for (int p = 0; p < np; p++) { sb[p] = dbuf + (size_t)p * 27; rb[p] = dbuf + (size_t)(np + p) * 27; }
    xchg_what = "count";
    TRY(xchg(np, peers.data(), sb.data(), nb.data(), rb.data(), nb.data()));
    recv_dir.assign((size_t)np * 27, 0);
    if (np > 0) HIPCHK(hipMemcpyAsync(recv_dir.data(), dbuf + (size_t)np * 27, recv_dir.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(h_flags + 16, d_dir_start, 28 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    for (int k = 0; k < 28; k++) h_ds[k] = h_flags[16 + k];
    send_n.assign(np, 0);
    for (int d = 0; d < 27; d++) {
        if (d == 13 && skip_stay) continue;
        int p = d == 13 ? -1 : peer_index[d];
        if (p >= 0) send_n[p] += h_ds[d + 1] - h_ds[d];
    }
    recv_n.assign(np, 0);
    for (int p = 0; p < np; p++)
        for (int d = 0; d < 27; d++) recv_n[p] += recv_dir[(size_t)p * 27 + d];
    return 0;
}

static void fill_dirtab(DirTab &T, const int *dir_start, const std::vector<int> &peers_send_base, const int *peer_index,
                        const std::vector<int> &send_n, const double *shift27, const double *center27, bool forward)
{
    // peer-major staging: for each peer, its directions in increasing order
    std::vector<int> fill(peers_send_base);
    for (int d = 0; d < 27; d++) {
        T.start[d] = dir_start[d];
        T.dst[d] = 0; T.vofs[d] = 0;
        for (int k = 0; k < 3; k++) { T.shift[d][k] = shift27[3 * d + k]; T.center[d][k] = center27[3 * d + k]; }
        int p = peer_index[d];
        if (p < 0) continue;
        int cnt = dir_start[d + 1] - dir_start[d];
        if (forward) {
            // block of peer p starts at 2*base_p: coords then velocities
            T.dst[d] = 2 * peers_send_base[p] + (fill[p] - peers_send_base[p]);
            T.vofs[d] = send_n[p];
        } else {
            T.dst[d] = fill[p];
        }
        fill[p] += cnt;
    }
    T.start[27] = dir_start[27];
}

// MesoComm::exchange (comm_meso.cu:256-550): atoms that left my sub-box move to the neighbour that owns them
int Engine::migrate()
{
    if (nranks == 1) return 0;
    tbegin("migrate");
    mig_holes = false;
    Decomp D = make_decomp(boxlo, boxhi, prd, procgrid, myloc);
    int *code = gslot;   // scratch (rebuilt later in the rebuild)
    int nchunk = (nlocal + 255) / 256;
    mig_lists_built = false;
    if (mig_slim_now()) {
        // (leavers' lists by atomics, ranked when packed: migrate_inband)
        int rc = migrate_inband();
        tend("migrate");
        return rc;
    }
    HIPCHK(hipMemsetAsync(d_flags + 3, 0, sizeof(int), stream));
    if (nlocal > 0) {
        hipLaunchKernelGGL(k_migrate_code, dim3(nchunk), dim3(256), 0, stream, cur.x[0], cur.x[1], cur.x[2], D, nlocal, code,
                           d_flags + 3);
        launch_border_count_code(code, 0, nlocal, chunk_count, nchunk, stream);
        HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
        launch_dir_starts(chunk_offset, nchunk, d_dir_start, stream);
        launch_border_fill_code(code, 0, nlocal, chunk_offset, nchunk, sendlist, stream);
    } else {
        HIPCHK(hipMemsetAsync(d_dir_start, 0, 28 * sizeof(int), stream));
    }
    mig_lists_built = true;
    if (async_counts && mig_caps_ready) { int rc = migrate_inband(); tend("migrate"); return rc; }
    // one host round trip: the "lost atoms" flag, my direction starts and the counts the peers announce arrive together
    HIPCHK(hipMemcpyAsync(h_flags + 3, d_flags + 3, sizeof(int), hipMemcpyDeviceToHost, stream));
    int ds[28], cnt[27];
    std::vector<int> send_n, recv_n, recv_dir;
    TRY(exchange_counts(1, ds, send_n, recv_n, recv_dir));
    if (h_flags[3]) return fail(5, "Atoms moved further than one sub-domain between rebuilds (lost atoms)");
    for (int d = 0; d < 27; d++) cnt[d] = ds[d + 1] - ds[d];
    int nstay = cnt[13];
    for (int d = 0; d < 27; d++)
        if (d != 13 && cnt[d] && peer_index[d] < 0) return fail(5, "Atom left the box through a non-periodic boundary");
    int np = (int)peers.size(), nsend_tot = 0, nrecv_tot = 0;
    std::vector<int> sbase(np, 0), rbase(np, 0);
    for (int p = 0; p < np; p++) { sbase[p] = nsend_tot; nsend_tot += send_n[p]; rbase[p] = nrecv_tot; nrecv_tot += recv_n[p]; }
    const int ms = mig_stride();
    TRY(ensure_stage((size_t)std::max(nsend_tot, 1) * ms * sizeof(double), (size_t)std::max(nrecv_tot, 1) * ms * sizeof(double)));
    if (nsend_tot > 0) {
        // destination slot of each direction's segment in the peer-major buffer
        std::vector<int> fill(sbase);
        int h_dst[27];
        for (int d = 0; d < 27; d++) {
            h_dst[d] = 0;
            int p = peer_index[d];
            if (d == 13 || p < 0) continue;
            h_dst[d] = fill[p];
            fill[p] += cnt[d];
        }
        HIPCHK(hipMemcpyAsync(sendlist_aux, h_dst, 27 * sizeof(int), hipMemcpyHostToDevice, stream));
        // segments before and after the stay segment
        if (ds[13] > 0)
            hipLaunchKernelGGL(k_pack_migrate, dim3((ds[13] + 255) / 256), dim3(256), 0, stream, cur, sendlist, d_dir_start,
                               sendlist_aux, 0, ds[13], ms, (double *)stage_send);
        int tail = ds[27] - ds[14];
        if (tail > 0)
            hipLaunchKernelGGL(k_pack_migrate, dim3((tail + 255) / 256), dim3(256), 0, stream, cur, sendlist, d_dir_start,
                               sendlist_aux, ds[14], tail, ms, (double *)stage_send);
    }
    // compact the stayers (their order is preserved; the reorder sort follows anyway)
    if (nstay != nlocal) {
        launch_permute_atoms(cur, alt, sendlist + ds[13], nstay, 1, stream);
        std::swap(cur, alt);
    }
    TRY(ensure_capacity(nstay + nrecv_tot + 1));
    std::vector<void *> sb(np), rb(np);
    std::vector<size_t> sn(np), rn(np);
    for (int p = 0; p < np; p++) {
        sb[p] = (double *)stage_send + (size_t)ms * sbase[p]; sn[p] = (size_t)send_n[p] * ms * sizeof(double);
        rb[p] = (double *)stage_recv + (size_t)ms * rbase[p]; rn[p] = (size_t)recv_n[p] * ms * sizeof(double);
    }
    xchg_what = "migration";
    TRY(xchg(np, peers.data(), sb.data(), sn.data(), rb.data(), rn.data()));
    if (nrecv_tot > 0)
        hipLaunchKernelGGL(k_unpack_migrate, dim3((nrecv_tot + 255) / 256), dim3(256), 0, stream, cur, (const double *)stage_recv,
                           d_mass_type, nstay, nrecv_tot, ms);
    nlocal = nstay + nrecv_tot;
    mig_update_caps(send_n, recv_n);
    tend("migrate");
    return 0;
}

void Engine::mig_update_caps(const std::vector<int> &send_n, const std::vector<int> &recv_n)
{
    // (the same number on both sides of a message: what A sent to B is what B received from A)
    const int np = (int)peers.size();
    mig_cap_s.assign(np, 0); mig_cap_r.assign(np, 0);
    for (int p = 0; p < np; p++) { mig_cap_s[p] = 2 * send_n[p] + mig_cap_floor; mig_cap_r[p] = 2 * recv_n[p] + mig_cap_floor; }
    mig_caps_ready = np > 0 && np <= 26;
}

// The exchange of migrate() with the counts in the messages (kernels above): fixed-capacity blocks, one host round trip for the
// counts, and an exact resend in a second exchange for the (rare) message that did not fit - zero-size messages are not posted, so
// that exchange costs nothing when nobody needs it, and the two ranks of a message decide alike (the sender knows its count, the
// receiver reads it in the header).  Called with the migration lists built (sendlist, d_dir_start).
// the slim front of the migration runs this rebuild (it also wraps the atoms: reneighbor then skips k_pbc)
bool Engine::mig_slim_now() const { return nranks > 1 && async_counts && mig_caps_ready && mig_slim; }

// the direction-major list of every local atom (sendlist, d_dir_start) from the migration codes: the counting chain, on demand
// when the slim front of the in-band migration ran (a message to send again exactly; stayers to compact for a sorting reorder)
int Engine::build_mig_lists()
{
    if (mig_lists_built) return 0;
    mig_lists_built = true;
    const int nchunk = (nlocal + 255) / 256;
    if (nlocal <= 0) return 0;
    launch_border_count_code(gslot, 0, nlocal, chunk_count, nchunk, stream);
    HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
    launch_dir_starts(chunk_offset, nchunk, d_dir_start, stream);
    launch_border_fill_code(gslot, 0, nlocal, chunk_offset, nchunk, sendlist, stream);
    return 0;
}

int Engine::migrate_inband()
{
    const int np = (int)peers.size();
    const int ms = mig_stride();
    MigPlan P;
    P.np = np;
    for (int d = 0; d < 27; d++) P.pidx[d] = d == 13 ? -1 : peer_index[d];
    long bs = 0, br = 0;
    int bound_s = 0;
    for (int p = 0; p < 26; p++) { P.cap_s[p] = P.cap_r[p] = 0; P.base_s[p] = P.base_r[p] = 0; }
    for (int p = 0; p < np; p++) {
        P.cap_s[p] = mig_cap_s[p]; P.cap_r[p] = mig_cap_r[p];
        P.base_s[p] = bs; bs += MIG_HDR_DOUBLES + (long)ms * mig_cap_s[p];
        P.base_r[p] = br; br += MIG_HDR_DOUBLES + (long)ms * mig_cap_r[p];
        bound_s += mig_cap_s[p];
    }
    TRY(ensure_stage((size_t)bs * sizeof(double), (size_t)br * sizeof(double)));
    if (!d_mr) { HIPCHK(hipMalloc((void **)&d_mr, 128 * sizeof(int))); HIPCHK(hipMemsetAsync(d_mr, 0, 128 * sizeof(int), stream)); }
    if (!h_flags_dev) HIPCHK(hipHostGetDevicePointer((void **)&h_flags_dev, h_flags, 0));
    int *dst_dev = d_mr + 96;           // [96..122]: slot of each direction's segment inside its peer's payload
    if (!mig_lists_built) {
        // the slim front: k_mig_scan + k_mig_pack
        int lc = 64;
        for (int p = 0; p < np; p++) lc = std::max(lc, mig_cap_s[p]);
        if (mig_lst_n < 27 * lc) {
            if (mig_lst) (void)hipFree(mig_lst);
            mig_lst = nullptr;
            mig_lst_n = 27 * lc * 2;
            HIPCHK(hipMalloc((void **)&mig_lst, (size_t)mig_lst_n * sizeof(int)));
        }
        if (!mig_cnt) { HIPCHK(hipMalloc((void **)&mig_cnt, 32 * sizeof(int))); HIPCHK(hipMemsetAsync(mig_cnt, 0, 32 * sizeof(int), stream)); }
        if (nlocal > 0) {
            Decomp D = make_decomp(boxlo, boxhi, prd, procgrid, myloc);
            PbcArgs pb;
            pb.on = 1;          // (the rebuild's wrap: reneighbor skipped k_pbc for this path)
            for (int d = 0; d < 3; d++) { pb.lo[d] = boxlo[d]; pb.hi[d] = boxhi[d]; pb.per[d] = periodic[d]; }
            pb.image = cur.image;
            hipLaunchKernelGGL(k_mig_scan, dim3((nlocal + 255) / 256), dim3(256), 0, stream, cur.x[0], cur.x[1], cur.x[2], D, nlocal, gslot,
                               mig_cnt, mig_lst, lc, pb);
        }
        hipLaunchKernelGGL(k_mig_pack, dim3(std::max(1, std::min(8, (lc + 255) / 256)), 27), dim3(256), 0, stream, cur, mig_cnt, mig_lst, lc, nlocal, P,
                           dst_dev, d_dir_start, ms, (double *)stage_send, d_flags + 3);
    } else {
    hipLaunchKernelGGL(k_mig_hdr, dim3(1), dim3(64), 0, stream, d_dir_start, P, dst_dev, (double *)stage_send);
    if (nlocal > 0)
        hipLaunchKernelGGL(k_pack_migrate_fixed, dim3(std::max(1, std::min((nlocal + 255) / 256, (bound_s + 255) / 256))), dim3(256), 0, stream,
                           cur, sendlist, d_dir_start, P, dst_dev, ms, (double *)stage_send);
    }
    std::vector<void *> sb(np), rb(np);
    std::vector<size_t> sn(np), rn(np);
    for (int p = 0; p < np; p++) {
        sb[p] = (double *)stage_send + P.base_s[p]; sn[p] = (size_t)(MIG_HDR_DOUBLES + (long)ms * mig_cap_s[p]) * sizeof(double);
        rb[p] = (double *)stage_recv + P.base_r[p]; rn[p] = (size_t)(MIG_HDR_DOUBLES + (long)ms * mig_cap_r[p]) * sizeof(double);
    }
    xchg_what = "migration (fixed capacity)";
    TRY(xchg(np, peers.data(), sb.data(), sn.data(), rb.data(), rn.data()));
    // the one host round trip: lost-atom flag, my direction starts, the counts in the peers' headers
    hipLaunchKernelGGL(k_mig_read_hdr, dim3(1), dim3(64), 0, stream, (const double *)stage_recv, P, h_flags_dev + 128,      // (slots of its own)
                       mig_lists_built ? nullptr : mig_cnt);
    HIPCHK(hipMemcpyAsync(h_flags + 3, d_flags + 3, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(h_flags + 16, d_dir_start, 28 * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    // an error of this rank is reported AFTER the second exchange: the peers of this rebuild get what they wait for, and the host
    // (LAMMPS error->one -> MPI_Abort, torch.distributed's launcher) ends the job of every rank on the status this one returns
    const char *deferred = nullptr;
    if (h_flags[3]) deferred = "Atoms moved further than one sub-domain between rebuilds (lost atoms)";
    int ds[28], cnt[27];
    for (int k = 0; k < 28; k++) ds[k] = h_flags[16 + k];
    for (int d = 0; d < 27; d++) cnt[d] = ds[d + 1] - ds[d];
    const int nstay = cnt[13];
    std::vector<int> send_n(np, 0), recv_n(np, 0);
    for (int d = 0; d < 27; d++) {
        if (d == 13 || !cnt[d]) continue;
        if (peer_index[d] < 0) { deferred = "Atom left the box through a non-periodic boundary"; continue; }
        send_n[peer_index[d]] += cnt[d];
    }
    int nrecv_tot = 0;
    bool again_s = false, again_r = false;
    for (int p = 0; p < np; p++) {
        recv_n[p] = h_flags[128 + p];
        if (recv_n[p] < 0) { deferred = "migration: corrupt message header"; recv_n[p] = 0; }
        nrecv_tot += recv_n[p];
        again_s |= send_n[p] > mig_cap_s[p];
        if (send_n[p] > mig_cap_s[p]) mig_resends++;
        again_r |= recv_n[p] > mig_cap_r[p];
    }
    // second exchange: the messages that did not fit, whole and exact (sizes known on both sides now)
    {
        std::vector<int> sbase(np, 0), rbase(np, 0);
        int stot = 0, rtot = 0;
        for (int p = 0; p < np; p++) {
            sbase[p] = stot; rbase[p] = rtot;
            if (send_n[p] > mig_cap_s[p]) stot += send_n[p];
            if (recv_n[p] > mig_cap_r[p]) rtot += recv_n[p];
        }
        if (again_s || again_r) {
            if ((size_t)std::max(stot, 1) * ms * sizeof(double) > stage2_send_bytes) {
                if (stage2_send) (void)hipFree(stage2_send);
                stage2_send = nullptr;
                stage2_send_bytes = (size_t)std::max(stot, 1) * ms * sizeof(double) * 2;
                HIPCHK(hipMalloc(&stage2_send, stage2_send_bytes));
            }
            if ((size_t)std::max(rtot, 1) * ms * sizeof(double) > stage2_recv_bytes) {
                if (stage2_recv) (void)hipFree(stage2_recv);
                stage2_recv = nullptr;
                stage2_recv_bytes = (size_t)std::max(rtot, 1) * ms * sizeof(double) * 2;
                HIPCHK(hipMalloc(&stage2_recv, stage2_recv_bytes));
            }
        }
        if (again_s) {
            TRY(build_mig_lists());
            // exact pack of the directions whose peer needs the resend (k_pack_migrate over the whole list, others go nowhere)
            std::vector<int> fill(sbase);
            int h_dst[27];
            for (int d = 0; d < 27; d++) {
                h_dst[d] = -1;
                const int q = peer_index[d];
                if (d == 13 || q < 0 || send_n[q] <= mig_cap_s[q]) continue;
                h_dst[d] = fill[q];
                fill[q] += cnt[d];
            }
            HIPCHK(hipMemcpyAsync(sendlist_aux, h_dst, 27 * sizeof(int), hipMemcpyHostToDevice, stream));
            if (ds[13] > 0)
                hipLaunchKernelGGL(k_pack_migrate, dim3((ds[13] + 255) / 256), dim3(256), 0, stream, cur, sendlist, d_dir_start, sendlist_aux, 0,
                                   ds[13], ms, (double *)stage2_send);
            const int tail = ds[27] - ds[14];
            if (tail > 0)
                hipLaunchKernelGGL(k_pack_migrate, dim3((tail + 255) / 256), dim3(256), 0, stream, cur, sendlist, d_dir_start, sendlist_aux, ds[14],
                                   tail, ms, (double *)stage2_send);
        }
        std::vector<void *> sb2(np, nullptr), rb2(np, nullptr);
        std::vector<size_t> sn2(np, 0), rn2(np, 0);
        for (int p = 0; p < np; p++) {
            if (send_n[p] > mig_cap_s[p]) { sb2[p] = (double *)stage2_send + (size_t)ms * sbase[p]; sn2[p] = (size_t)send_n[p] * ms * sizeof(double); }
            if (recv_n[p] > mig_cap_r[p]) { rb2[p] = (double *)stage2_recv + (size_t)ms * rbase[p]; rn2[p] = (size_t)recv_n[p] * ms * sizeof(double); }
        }
        xchg_what = "migration (resend)";
        TRY(xchg(np, peers.data(), sb2.data(), sn2.data(), rb2.data(), rn2.data()));
        if (deferred) return fail(5, deferred);
        // where every peer's migrants lie
        MigSources S;
        S.np = np;
        int run = 0;
        for (int p = 0; p < 26; p++) { S.src[p] = 0; S.second[p] = 0; }
        for (int p = 0; p < np; p++) {
            S.gbase[p] = run; run += recv_n[p];
            if (recv_n[p] > mig_cap_r[p]) { S.second[p] = 1; S.src[p] = (long)ms * rbase[p]; }
            else S.src[p] = P.base_r[p] + MIG_HDR_DOUBLES;
        }
        for (int p = np; p < 27; p++) S.gbase[p] = run;
        // The stayers are NOT compacted when the reorder that follows is the counting one (rebuild.hip): its count skips the
        // leavers by their migration code and walks the arrivals behind the old atoms - no copy of every atom for the few that left
        // (k_permute_atoms: 131 k atoms x 200 B per rebuild).  Otherwise (sorting reorder; arrays that have to grow first, which
        // would drop the codes): compact, then append.
        const bool holes = reorder_fuses((long)nlocal + nrecv_tot) && nlocal + nrecv_tot + 1 <= nmax && nlocal > 0;
        int base = nstay;
        if (holes) {
            mig_holes = true;
            mig_nold = nlocal;
            mig_span = nlocal + nrecv_tot;
            base = nlocal;
        } else if (nstay != nlocal) {
            TRY(build_mig_lists());
            launch_permute_atoms(cur, alt, sendlist + ds[13], nstay, 1, stream);
            std::swap(cur, alt);
        }
        if (!holes) TRY(ensure_capacity(nstay + nrecv_tot + 1));
        if (nrecv_tot > 0)
            hipLaunchKernelGGL(k_unpack_migrate_from, dim3((nrecv_tot + 255) / 256), dim3(256), 0, stream, cur, (const double *)stage_recv,
                               (const double *)stage2_recv, S, d_mass_type, base, ms);
        nlocal = nstay + nrecv_tot;
    }
    if (getenv("MESO_DEBUG_BUILD") && rank == 0 && (again_s || again_r)) fprintf(stderr, "migration: %ld message(s) of rank 0 sent again so far\n", mig_resends);
    mig_update_caps(send_n, recv_n);
    return 0;
}

// MesoComm::borders (comm_meso.cu:41-186) for nranks > 1
int Engine::halo_borders_multi()
{
    if (mr_async_ok()) return halo_borders_multi_async();
    tbegin("halo");
    // (bulk_pending: the reorder left the bulk count on its way to the host - the scan then covers every local atom, bulk atoms
    // carry no border flags, and the count arrives with the round trip below)
    int beg = bulk_pending ? 0 : n_bulk, end = nlocal;
    int nchunk = (end - beg + 255) / 256;
    if (nchunk > 0) {
        launch_border_count(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, stream);
        HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
        launch_dir_starts(chunk_offset, nchunk, d_dir_start, stream);
    } else {
        HIPCHK(hipMemsetAsync(d_dir_start, 0, 28 * sizeof(int), stream));
    }
    // one host round trip for my direction starts and the peers' counts (the counts go device to device)
    std::vector<int> recv_dir;
    TRY(exchange_counts(0, h_dir_start, peer_send_n, peer_recv_n, recv_dir));
    if (bulk_pending) {
        bulk_pending = false;
        if (h_flags[0]) return check_overflow();
        n_bulk = h_flags[1];
    }
    nsend = h_dir_start[27];
    for (int d = 0; d < 27; d++)
        if (h_dir_start[d + 1] - h_dir_start[d] && peer_index[d] < 0) return fail(5, "border list for an inactive direction");
    int np = (int)peers.size();
    peer_send_base.assign(np, 0); peer_recv_base.assign(np + 1, 0);
    int stot = 0, rtot = 0;
    for (int p = 0; p < np; p++) { peer_send_base[p] = stot; stot += peer_send_n[p]; peer_recv_base[p] = rtot; rtot += peer_recv_n[p]; }
    peer_recv_base[np] = rtot;
    nghost = rtot;
    {
        // a growth reallocates chunk_count / chunk_offset (alloc_atoms keeps their contents as little as the single-rank
        // path's, engine.hip halo_borders): count and scan again before the fill pass reads the offsets
        const int nmax_before = nmax;
        TRY(ensure_capacity(nlocal + nghost + 1));
        if (nmax != nmax_before && nchunk > 0) {
            launch_border_count(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, stream);
            HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
            launch_dir_starts(chunk_offset, nchunk, d_dir_start, stream);
        }
    }
    if (nsend > send_cap) return fail(4, "send list capacity exceeded");
    TRY(ensure_stage((size_t)std::max(std::max(stot, 1) * BORDER_DOUBLES * sizeof(double), (size_t)std::max(stot, 1) * 2 * sizeof(float4)),
                     (size_t)std::max(std::max(rtot, 1) * BORDER_DOUBLES * sizeof(double), (size_t)std::max(rtot, 1) * 2 * sizeof(float4))));
    if (nsend > 0) launch_border_fill(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_offset, nchunk, sendlist, stream);
    DirTab T;
    fill_dirtab(T, h_dir_start, peer_send_base, peer_index, peer_send_n, shift27, center27, false);
    if (nsend > 0)
        hipLaunchKernelGGL(k_pack_border_multi, dim3((nsend + 255) / 256), dim3(256), 0, stream, cur, sendlist, nsend, T,
                           (double *)stage_send);
    std::vector<void *> sb(np), rb(np);
    std::vector<size_t> sn(np), rn(np);
    for (int p = 0; p < np; p++) {
        sb[p] = (double *)stage_send + BORDER_DOUBLES * (size_t)peer_send_base[p]; sn[p] = (size_t)peer_send_n[p] * BORDER_DOUBLES * sizeof(double);
        rb[p] = (double *)stage_recv + BORDER_DOUBLES * (size_t)peer_recv_base[p]; rn[p] = (size_t)peer_recv_n[p] * BORDER_DOUBLES * sizeof(double);
    }
    xchg_what = "border";
    TRY(xchg(np, peers.data(), sb.data(), sn.data(), rb.data(), rn.data()));
    if (nghost > 0)
        hipLaunchKernelGGL(k_unpack_border, dim3((nghost + 255) / 256), dim3(256), 0, stream, cur, (const double *)stage_recv,
                           nlocal, nghost);
    // tables of the per-step refresh
    fill_dirtab(fwd_tab_host(), h_dir_start, peer_send_base, peer_index, peer_send_n, shift27, center27, true);
    mr_update_caps();
    tend("halo");
    return 0;
}

// ---- the same stage with its counts left on the device (see the kernels above)
bool Engine::mr_async_ok() const
{
    // (every term is the same on all ranks: the two sides of a message must choose the same format)
    return async_counts && nranks > 1 && mr_caps_ready && !ghost_sort && !reorder_sort;
}

void Engine::mr_update_caps()
{
    const int np = (int)peers.size();
    mr_cap_s.assign(np, 0); mr_cap_r.assign(np, 0);
    for (int p = 0; p < np; p++) {
        // (integer arithmetic on the same inputs: both sides of a message get the same number)
        const long m1024 = (long)(mr_cap_margin * 1024.0);
        mr_cap_s[p] = (int)std::max(1L, peer_send_n[p] + peer_send_n[p] * m1024 / 1024 + (mr_cap_margin < 0 ? 0 : 256));
        mr_cap_r[p] = (int)std::max(1L, peer_recv_n[p] + peer_recv_n[p] * m1024 / 1024 + (mr_cap_margin < 0 ? 0 : 256));
    }
    mr_caps_ready = np > 0 && np <= 26;
}

int Engine::halo_borders_multi_async()
{
    if (getenv("MESO_DEBUG_BUILD") && rank == 0) fprintf(stderr, "borders: fixed-capacity exchange, no host round trip (rank 0 caps %d...)\n", mr_cap_s.empty() ? 0 : mr_cap_s[0]);
    tbegin("halo");
    const int np = (int)peers.size();
    MrPlan P;
    P.np = np;
    for (int d = 0; d < 27; d++) P.pidx[d] = d == 13 ? -1 : peer_index[d];
    long bs = 0, br = 0;
    int bound_s = 0, bound_r = 0;
    for (int p = 0; p < 26; p++) { P.cap_s[p] = P.cap_r[p] = 0; P.base_s[p] = P.base_r[p] = 0; }
    for (int p = 0; p < np; p++) {
        P.cap_s[p] = mr_cap_s[p]; P.cap_r[p] = mr_cap_r[p];
        P.base_s[p] = bs; bs += MR_HDR_DOUBLES + (long)BORDER_DOUBLES * mr_cap_s[p];
        P.base_r[p] = br; br += MR_HDR_DOUBLES + (long)BORDER_DOUBLES * mr_cap_r[p];
        bound_s += mr_cap_s[p]; bound_r += mr_cap_r[p];
    }
    // capacities first: nothing below may reallocate (the chunk arrays are regrown with the atoms)
    TRY(ensure_capacity(nlocal + bound_r + 1));
    TRY(ensure_stage((size_t)std::max((size_t)bs * sizeof(double), (size_t)std::max(bound_s, 1) * 2 * sizeof(float4)),
                     (size_t)std::max((size_t)br * sizeof(double), (size_t)std::max(bound_r, 1) * 2 * sizeof(float4))));
    if (bound_s > send_cap) return fail(4, "send list capacity exceeded");
    if (!d_mr) { HIPCHK(hipMalloc((void **)&d_mr, 128 * sizeof(int))); HIPCHK(hipMemsetAsync(d_mr, 0, 128 * sizeof(int), stream)); }
    if (!h_flags_dev) HIPCHK(hipHostGetDevicePointer((void **)&h_flags_dev, h_flags, 0));
    // border lists: every local atom while the bulk count of the reorder is still on the device, the border section otherwise
    const int beg = bulk_pending ? 0 : n_bulk, end = nlocal;
    const int nchunk = (end - beg + 255) / 256;
    const int *nb_dev = estart + bargs.M;
    if (nchunk > 0) {
        launch_border_count(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_count, nchunk, stream, novf_pending ? fr_novf : nullptr);
        if (!launch_border_scan(chunk_count, chunk_offset, nchunk, d_dir_start, nb_dev, beg, std::min(bound_s, send_cap), d_flags, h_flags_dev, stream)) {
            HIPCHK(exclusive_scan_i32(sort_temp, sort_temp_bytes, chunk_count, chunk_offset, 27 * nchunk + 1, stream));
            launch_dir_starts_check(chunk_offset, nchunk, d_dir_start, nb_dev, beg, std::min(bound_s, send_cap), d_flags, stream);
            HIPCHK(hipMemcpyAsync(h_flags + 16, d_dir_start, 28 * sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_flags + 8, d_flags, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_flags + 9, nb_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
        }
    } else {
        HIPCHK(hipMemsetAsync(d_dir_start, 0, 28 * sizeof(int), stream));
        if (novf_pending) HIPCHK(hipMemsetAsync(fr_novf, 0, sizeof(int), stream));
        for (int k = 0; k < 28; k++) h_flags[16 + k] = 0;
        h_flags[8] = 0; h_flags[9] = n_bulk;
    }
    novf_pending = false;
    // ghosts in message order, their cells as runs (one kernel instead of unpack + count + scan + place + order + merge) when the
    // list builder reads (start, count) per ghost cell: the tile builder with the plan in its prologue, bins no narrower than the
    // ghost cutoff (a ghost cell is then fed by one source cell)
    bool runs = border_runs && neigh_kernel == 1 && tile_fits && n_col <= tile_build_rowcap() && tile_plan == 0 && n_col >= 64;
    for (int d = 0; d < 3; d++) runs = runs && geom.binsize[d] >= cutghost;
    bool gcnt_cleared = false;
    if (runs && mr_gcnt_n < bargs.M + 1) {
        if (mr_gcnt) (void)hipFree(mr_gcnt);
        mr_gcnt = nullptr;
        mr_gcnt_n = bargs.M + 1;
        HIPCHK(hipMalloc((void **)&mr_gcnt, (size_t)mr_gcnt_n * sizeof(int)));
    }
    Shift27 sh;
    for (int d = 0; d < 27; d++) for (int k = 0; k < 3; k++) sh.s[d][k] = shift27[3 * d + k];
    if (nchunk > 0 && border_fused) {
        Slabs sl;
        for (int d = 0; d < 3; d++) { sl.lo[d] = slab_lo[d]; sl.hi[d] = slab_hi[d]; }
        // (the reorder cleared the image counters of every atom when mr_img: fused_locals_args)
        const bool rec = mr_img_wanted() && img_zero_gen == img_alloc_gen;      // (... and the arrays were not regrown since)
        if (rec && !d_vofs) { HIPCHK(hipMalloc((void **)&d_vofs, 32 * sizeof(int))); }
        hipLaunchKernelGGL(k_border_fill_pack, dim3(nchunk), dim3(256), 0, stream, cur, beg, end, sl, chunk_offset, nchunk, sendlist, d_dir_start, P, sh,
                           d_mr, geom, (double *)stage_send, d_flags, h_flags_dev + 64, rec ? img_cnt : nullptr, rec ? img : nullptr, d_vofs,
                           runs ? mr_gcnt : nullptr, runs ? bargs.M + 1 : 0);
        gcnt_cleared = runs;
        mr_images_ready = rec;
        mr_img_stage = stage_send;
    } else {
    hipLaunchKernelGGL(k_border_hdr, dim3(1), dim3(64), 0, stream, d_dir_start, P, d_mr, (double *)stage_send, d_flags, h_flags_dev + 64);
    if (nchunk > 0) launch_border_fill(cur, beg, end, slab_lo, slab_hi, nullptr, chunk_offset, nchunk, sendlist, stream);
    if (bound_s > 0)
        hipLaunchKernelGGL(k_pack_border_fixed, dim3((bound_s + 255) / 256), dim3(256), 0, stream, cur, sendlist, d_dir_start, P, sh, d_mr, geom,
                           (double *)stage_send);
    }
    std::vector<void *> sb(np), rb(np);
    std::vector<size_t> sn(np), rn(np);
    for (int p = 0; p < np; p++) {
        sb[p] = (double *)stage_send + P.base_s[p]; sn[p] = (size_t)(MR_HDR_DOUBLES + BORDER_DOUBLES * mr_cap_s[p]) * sizeof(double);
        rb[p] = (double *)stage_recv + P.base_r[p]; rn[p] = (size_t)(MR_HDR_DOUBLES + BORDER_DOUBLES * mr_cap_r[p]) * sizeof(double);
    }
    xchg_what = "border (fixed capacity)";
    TRY(xchg(np, peers.data(), sb.data(), sn.data(), rb.data(), rn.data()));
    hipLaunchKernelGGL(k_border_unpack_hdr, dim3(1), dim3(64), 0, stream, (const double *)stage_recv, P, nmax - nlocal - 1, d_mr, d_flags,
                       h_flags_dev + 64);
    if (runs) {
        // (the ghost-cell counts were cleared by the border kernel ahead of the exchange, or are now)
        if (!gcnt_cleared) HIPCHK(hipMemsetAsync(mr_gcnt, 0, (size_t)(bargs.M + 1) * sizeof(int), stream));
        if (bound_r > 0)
            hipLaunchKernelGGL(k_unpack_ghost_runs, dim3((bound_r + 255) / 256), dim3(256), 0, stream, cur, (const double *)stage_recv, P, d_mr, nlocal,
                               0.5 * (subhi[0] + sublo[0]), 0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]),
                               premix_tea<64>((u32)seed, (u32)ntimestep), coord4 + nlocal, veloc4 + nlocal, gstart, mr_gcnt, bargs.M, d_flags);
        mr_runs = true;
        ghosts_binned = true;
    } else if (bound_r > 0)
        hipLaunchKernelGGL(k_unpack_border_fixed, dim3((bound_r + 255) / 256), dim3(256), 0, stream, cur, (const double *)stage_recv, P, d_mr,
                           nlocal);
    // the report (flag, bulk count, direction starts: k_border_scan; per-peer counts: the two header kernels) is in pinned memory
    // once this event has passed
    HIPCHK(hipMemcpyAsync(h_flags + 8, d_flags, sizeof(int), hipMemcpyDeviceToHost, stream));
    if (!ev_counts) HIPCHK(hipEventCreateWithFlags(&ev_counts, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_counts, stream));
    counts_by_seq = false;
    counts_pending = true;
    mr_pending = true;
    nsend = bound_s; nghost = bound_r;            // launch bounds until resolve_counts() has the numbers
    tend("halo");
    return 0;
}

// host tables of the exchange whose counts have just arrived (Engine::resolve_counts): what halo_borders_multi computes on the spot
int Engine::mr_resolve()
{
    mr_pending = false;
    const int np = (int)peers.size();
    const int *rep = h_flags + 64;
    nsend = h_dir_start[27];
    peer_send_n.assign(np, 0); peer_recv_n.assign(np, 0);
    for (int d = 0; d < 27; d++) {
        if (h_dir_start[d + 1] - h_dir_start[d] && (d == 13 || peer_index[d] < 0)) return fail(5, "border list for an inactive direction");
        if (d != 13 && peer_index[d] >= 0) peer_send_n[peer_index[d]] += h_dir_start[d + 1] - h_dir_start[d];
    }
    peer_send_base.assign(np, 0); peer_recv_base.assign(np + 1, 0);
    int stot = 0, rtot = 0;
    for (int p = 0; p < np; p++) {
        peer_recv_n[p] = rep[p];
        if (rep[32 + p] != peer_send_n[p]) return fail(5, "border exchange: the device-side and host-side send counts differ");
        peer_send_base[p] = stot; stot += peer_send_n[p]; peer_recv_base[p] = rtot; rtot += peer_recv_n[p];
    }
    peer_recv_base[np] = rtot;
    if (rtot != rep[31]) return fail(5, "border exchange: inconsistent ghost count report");
    nghost = rtot;
    TRY(ensure_stage((size_t)std::max(stot, 1) * 2 * sizeof(float4), (size_t)std::max(rtot, 1) * 2 * sizeof(float4)));
    fill_dirtab(fwd_tab_host(), h_dir_start, peer_send_base, peer_index, peer_send_n, shift27, center27, true);
    mr_update_caps();
    return 0;
}

DirTab &Engine::fwd_tab_host()
{
    if (!fwd_tab) fwd_tab = new DirTab;
    return *fwd_tab;
}

void Engine::free_fwd_tab()
{
    delete fwd_tab;
    fwd_tab = nullptr;
}

// Comm::forward_comm for nranks > 1.  Split in two so the engine can run the bulk force kernel between them.
int Engine::halo_forward_multi_begin(uint32_t sd, bool async)
{
    TRY(resolve_counts());      // (the tables of the refresh come from the counts of the last rebuild)
    tbegin("halo");
    int np = (int)peers.size();
    // (fwd_packed: the previous step's force kernel wrote this refresh into the staging - run loop)
    const bool packed = fwd_packed && stage_send == mr_img_stage;
    fwd_packed = false;
    if (nsend > 0 && !packed)
        hipLaunchKernelGGL(k_pack_forward_multi, dim3((nsend + 255) / 256), dim3(256), 0, stream, cur, sendlist, nsend,
                           fwd_tab_host(), sd, (float4 *)stage_send);
    std::vector<void *> sb(np), rb(np);
    std::vector<size_t> sn(np), rn(np);
    for (int p = 0; p < np; p++) {
        sb[p] = (float4 *)stage_send + 2 * (size_t)peer_send_base[p]; sn[p] = (size_t)peer_send_n[p] * 2 * sizeof(float4);
        rb[p] = (float4 *)stage_recv + 2 * (size_t)peer_recv_base[p]; rn[p] = (size_t)peer_recv_n[p] * 2 * sizeof(float4);
    }
    // exchange + scatter run on the side stream so the bulk force kernel (main stream) overlaps them, the role the
    // reference gives its bulk/border split (mvv_meso.cu:338-362) to hide the PCIe + MPI round trip
    if (!ev_pack) { HIPCHK(hipEventCreateWithFlags(&ev_pack, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&ev_halo, hipEventDisableTiming)); }
    HIPCHK(hipEventRecord(ev_pack, stream));
    HIPCHK(hipStreamWaitEvent(side, ev_pack, 0));
    xs = side;
    xchg_what = "ghost refresh";
    // ghosts in message order (this rebuild's ghost stage left them so: mr_runs): the halves of every message go straight into the
    // merged arrays - no scatter kernel
    const bool direct = mr_runs && refresh_direct;
    std::vector<void *> rb2;
    if (direct) {
        rb2.resize(np);
        for (int p = 0; p < np; p++) { rb[p] = coord4 + nlocal + peer_recv_base[p]; rb2[p] = veloc4 + nlocal + peer_recv_base[p]; }
    }
    int rc = xchg(np, peers.data(), sb.data(), sn.data(), rb.data(), rn.data(), direct ? rb2.data() : nullptr);
    xs = nullptr;
    if (rc) return rc;
    if (debug_early_reuse && nsend > 0) {
        // tests only (tests/test_gpu_rccl_branch.py): the send staging is overwritten from a stream that waits for nothing - the hazard
        // an asynchronous transport adds and a host-synchronous stand-in cannot see
        if (!debug_stream) HIPCHK(hipStreamCreateWithFlags(&debug_stream, hipStreamNonBlocking));
        HIPCHK(hipMemsetAsync(stage_send, 0xFF, std::min<size_t>((size_t)nsend * 2 * sizeof(float4), 1 << 16), debug_stream));
        // (it races with this exchange only: whatever the engine does next on its own streams waits for the scribble)
        if (!debug_event) HIPCHK(hipEventCreateWithFlags(&debug_event, hipEventDisableTiming));
        HIPCHK(hipEventRecord(debug_event, debug_stream));
        HIPCHK(hipStreamWaitEvent(stream, debug_event, 0));
        HIPCHK(hipStreamWaitEvent(side, debug_event, 0));
    }
    if (nghost > 0 && !direct) {
        PeerTab P;
        P.np = np;
        for (int p = 0; p <= np; p++) P.gbase[p] = peer_recv_base[p];
        hipLaunchKernelGGL(k_scatter_ghost, dim3((nghost + 255) / 256), dim3(256), 0, side, (const float4 *)stage_recv, P,
                           (layout >= 1 && !mr_runs) ? gslot : nullptr, nghost, coord4 + nlocal, veloc4 + nlocal);
    }
    HIPCHK(hipEventRecord(ev_halo, side));
    if (!async) HIPCHK(hipStreamWaitEvent(stream, ev_halo, 0));
    tend("halo");
    return 0;
}

// the ghosts of a rebuild as merged float4 pairs (several ranks): local, no exchange - their velocities came with the border message
int Engine::merge_new_ghosts(uint32_t sd)
{
    if (nghost <= 0) return 0;
    tbegin("halo");
    hipLaunchKernelGGL(k_merge_ghosts, dim3((nghost + 255) / 256), dim3(256), 0, stream, cur, nlocal, nghost, pending_nghost_dev(), (layout >= 1 && !mr_runs) ? gslot : nullptr,
                       0.5 * (subhi[0] + sublo[0]), 0.5 * (subhi[1] + sublo[1]), 0.5 * (subhi[2] + sublo[2]), sd, coord4 + nlocal, veloc4 + nlocal);
    tend("halo");
    return 0;
}

int Engine::halo_wait()
{
    if (nranks > 1 && ev_halo) HIPCHK(hipStreamWaitEvent(stream, ev_halo, 0));
    return 0;
}

int Engine::ensure_stage(size_t sbytes, size_t rbytes)
{
    if (sbytes > stage_send_bytes) {
        if (stage_send) (void)hipFree(stage_send);
        stage_send = nullptr;
        stage_send_bytes = sbytes + sbytes / 4 + 4096;
        HIPCHK(hipMalloc(&stage_send, stage_send_bytes));
    }
    if (rbytes > stage_recv_bytes) {
        if (stage_recv) (void)hipFree(stage_recv);
        stage_recv = nullptr;
        stage_recv_bytes = rbytes + rbytes / 4 + 4096;
        HIPCHK(hipMalloc(&stage_recv, stage_recv_bytes));
    }
    if (!sendlist_aux) HIPCHK(hipMalloc((void **)&sendlist_aux, 2 * 27 * 27 * sizeof(int) + 64));
    return 0;
}

} // namespace meso
