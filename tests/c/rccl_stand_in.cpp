// TEST INFRASTRUCTURE ONLY - an in-process stand-in for the nine librccl entry points the engine calls (ncclGetUniqueId,
// ncclCommInitRank, ncclCommCount, ncclGetErrorString, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv).
//
// RCCL refuses two ranks on one device ("Duplicate GPU detected", tests/test_gpu_multiproc.py), so on the one-GPU test box the
// engine's RCCL branch - Engine::comm_init("rccl") and the grouped send / receive schedule of Engine::xchg (comm.hip) - never ran.
// Preloaded into a test process (LD_PRELOAD), this library lets several in-process ranks (one host thread and one engine context
// each, as the LOCAL transport's tests have them) run that very branch: it keeps RCCL's matching rules - point-to-point messages
// between two ranks match in the order they were posted, a send and its receive must agree on the byte count, everything inside
// one ncclGroupStart / ncclGroupEnd is posted together - and moves the bytes with device-to-device copies.  It is not shipped, not
// linked by the product, and says nothing about RCCL's performance.  Two modes: host-synchronous (default: validates sizes, offsets,
// order and matching) and, with RCCL_STAND_IN_ASYNC=1, stream-ordered like RCCL itself (run_ops_async: also validates the engine's
// stream and event order around its exchanges - buffer reuse, reads of ghosts in flight).
//
// Build: hipcc --offload-arch=gfx950 -shared -fPIC tests/c/rccl_stand_in.cpp -o tests/c/librccl_stand_in.so  (__graft_entry__.build())
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Msg {
    const void *buf;
    size_t bytes;
    bool taken = false;
    // asynchronous mode: `ready` is recorded on the sender's stream when the message is posted (what the buffer holds is complete
    // once it has passed); `done` is recorded on the receiver's stream behind its copy (the sender's stream waits for it)
    hipEvent_t ready = nullptr, done = nullptr;
    bool done_recorded = false;
};
struct Group {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<std::shared_ptr<Msg>>> box;      // [src * nranks + dst]: messages posted and not yet received, in order
};
struct Comm {
    Group *g;
    int rank;
};
struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *c;
    hipStream_t stream;
};

std::mutex g_reg_mu;
std::map<std::string, Group *> g_registry;
std::atomic<long> g_next_id{1};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

// RCCL_STAND_IN_ASYNC=1: RCCL's stream semantics instead of host synchronisation.  A group becomes work ON the callers' streams: the
// receiver's stream waits for the sender's `ready` event, copies, records `done`; the sender's stream waits for `done` before anything
// posted behind the send may run.  No hipStreamSynchronize anywhere - the host threads only rendezvous with each other (a receive needs
// the peer's message descriptor, a send needs the event its receiver records), so whatever the engine enqueues behind a group really
// runs concurrently with the peers' copies unless the engine's own stream / event order forbids it.  A staging buffer that is
// rewritten from another stream without waiting for the exchange, or a ghost array read before its receive has landed, now corrupts
// the trajectory and fails the bit-identity test; the synchronous mode (default) cannot see either.
// RCCL_STAND_IN_DELAY_US=n (stream-ordered mode): every receive is held back by a spin kernel of n microseconds on the receiver's stream, so
// that whatever the engine enqueues behind a group WITHOUT ordering it behind the group has certainly run before the bytes move - a
// hazard then shows on every run, not on the runs where the race happens to go that way
__global__ void k_stand_in_delay(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static long long delay_ticks()
{
    static const long long t = [] {
        const char *e = getenv("RCCL_STAND_IN_DELAY_US");
        const long long us = (e && *e) ? atoll(e) : 0;
        return us > 0 ? us * 100 : 0;          // wall_clock64 counts at 100 MHz on gfx9
    }();
    return t;
}

static bool async_mode()
{
    static const bool on = [] { const char *e = getenv("RCCL_STAND_IN_ASYNC"); return e && *e && *e != '0'; }();
    return on;
}

ncclResult_t run_ops_async(std::vector<Op> &ops)
{
    std::vector<std::shared_ptr<Msg>> mine;
    // 1. post every send: its `ready` event is on my stream before the peer can see the message
    for (auto &o : ops) {
        if (!o.send) continue;
        Group *g = o.c->g;
        auto m = std::make_shared<Msg>();
        m->buf = o.buf; m->bytes = o.bytes;
        if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(m->ready, o.stream) != hipSuccess)
            return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->box[(size_t)o.c->rank * g->nranks + o.peer].push_back(m);
        }
        g->cv.notify_all();
        mine.push_back(m);
    }
    ncclResult_t rc = ncclSuccess;
    // 2. every receive: wait (host) for the peer's descriptor, then the copy rides on MY stream behind the peer's `ready`
    for (auto &o : ops) {
        if (o.send) continue;
        Group *g = o.c->g;
        std::shared_ptr<Msg> m;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            auto &q = g->box[(size_t)o.peer * g->nranks + o.c->rank];
            g->cv.wait(lk, [&] { return !q.empty(); });
            m = q.front();
            q.pop_front();
        }
        if (m->bytes != o.bytes) {
            fprintf(stderr, "rccl stand-in: rank %d receives %zu bytes from rank %d, which sent %zu\n", o.c->rank, o.bytes, o.peer, m->bytes);
            rc = ncclInvalidUsage;
        } else {
            if (hipStreamWaitEvent(o.stream, m->ready, 0) != hipSuccess) rc = ncclUnhandledCudaError;
            if (delay_ticks()) hipLaunchKernelGGL(k_stand_in_delay, dim3(1), dim3(1), 0, o.stream, delay_ticks());
            if (o.bytes && hipMemcpyAsync(o.buf, m->buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
        }
        if (hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess || hipEventRecord(m->done, o.stream) != hipSuccess)
            rc = ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            m->done_recorded = true;
        }
        g->cv.notify_all();
    }
    // 3. every send: my stream goes on only when the receiver's copy has run (the buffer may be rewritten behind this point)
    for (size_t k = 0, s = 0; k < ops.size(); k++) {
        if (!ops[k].send) continue;
        Group *g = ops[k].c->g;
        auto &m = mine[s++];
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return m->done_recorded; });
        }
        if (hipStreamWaitEvent(ops[k].stream, m->done, 0) != hipSuccess) rc = ncclUnhandledCudaError;
        // (both events have their waiters enqueued: the runtime keeps what it still needs)
        (void)hipEventDestroy(m->ready); (void)hipEventDestroy(m->done);
    }
    return rc;
}

ncclResult_t run_ops(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    if (async_mode()) return run_ops_async(ops);
    // what the messages hold must be complete before a peer copies it
    for (auto &o : ops) if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    std::vector<std::shared_ptr<Msg>> mine;
    for (auto &o : ops) {
        if (!o.send) continue;
        Group *g = o.c->g;
        auto m = std::make_shared<Msg>();
        m->buf = o.buf; m->bytes = o.bytes;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->box[(size_t)o.c->rank * g->nranks + o.peer].push_back(m);
        }
        g->cv.notify_all();
        mine.push_back(m);
    }
    ncclResult_t rc = ncclSuccess;
    for (auto &o : ops) {
        if (o.send) continue;
        Group *g = o.c->g;
        std::shared_ptr<Msg> m;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            auto &q = g->box[(size_t)o.peer * g->nranks + o.c->rank];
            g->cv.wait(lk, [&] { return !q.empty(); });
            m = q.front();
            q.pop_front();
        }
        if (m->bytes != o.bytes) {
            fprintf(stderr, "rccl stand-in: rank %d receives %zu bytes from rank %d, which sent %zu\n", o.c->rank, o.bytes, o.peer, m->bytes);
            rc = ncclInvalidUsage;
        } else if (o.bytes && (hipMemcpyAsync(o.buf, m->buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
                               hipStreamSynchronize(o.stream) != hipSuccess))
            rc = ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            m->taken = true;
        }
        g->cv.notify_all();
    }
    // a send buffer may be reused once its message was received
    for (size_t k = 0, s = 0; k < ops.size(); k++) {
        if (!ops[k].send) continue;
        Group *g = ops[k].c->g;
        std::unique_lock<std::mutex> lk(g->mu);
        auto &m = mine[s++];
        g->cv.wait(lk, [&] { return m->taken; });
    }
    return rc;
}

}      // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "stand-in-%ld", g_next_id.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    std::lock_guard<std::mutex> lk(g_reg_mu);
    Group *&g = g_registry[std::string(id.internal, sizeof id.internal)];
    if (!g) {
        g = new Group;
        g->nranks = nranks;
        g->box.resize((size_t)nranks * nranks);
    }
    if (g->nranks != nranks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm{g, rank};
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "stand-in error"; }

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    *count = reinterpret_cast<Comm *>(comm)->g->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<Comm *>(comm);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (--t_depth > 0) return ncclSuccess;
    t_depth = 0;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    t_ops.push_back(Op{true, const_cast<void *>(sendbuff), count * type_bytes(datatype), peer, reinterpret_cast<Comm *>(comm), stream});
    if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run_ops(ops); }
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    t_ops.push_back(Op{false, recvbuff, count * type_bytes(datatype), peer, reinterpret_cast<Comm *>(comm), stream});
    if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run_ops(ops); }
    return ncclSuccess;
}

}      // extern "C"
