// Measured floors of the fp32 force kernel's MANDATORY work (diagnostics, not on the product path:
// meso_pair_floor, tools/pair_floor.py, the `floor_us` block of the bench line).
//
// gpu_dpd_fast<0> (pair_dpd_fast_meso.cu:124-162) prescribes per stored row entry one 16-byte coordinate gather and one cutoff
// test, and per pair inside the cutoff one 16-byte velocity gather, four TEA rounds, the Gaussian, the weights and the force.
// The ring kernel (pair_ring.hip) waits for its own gathers at 1.13 x the algorithmic HBM traffic; whether a better arrangement of
// the SAME work could reach the north star's 0.50 of the HBM roofline (53 us at 64^3) is answered here by timing that work alone,
// on the real table of the running system, in three idealised kernels that drop everything a formulation adds (ballots, ring
// records, pairing decisions, drain checks):
//
//   ARITH   the arithmetic only: per wave ceil(row entries / 64) full-lane distance tests and ceil(evaluated pairs / 64) full-lane
//           pair evaluations (the ring kernel's own FAST / one-type / s = 1 code: TEA, v_sin / v_log / v_sqrt / v_rsq, fixed-point
//           LDS sums), operands from LDS; own records read and forces written as the real kernel must (56 B per atom);
//   GATHER  the address stream only: the front sections' row words, one coordinate gather per entry, one velocity gather per
//           evaluated pair whose partner is outside the workgroup (the hits of each wave come as a ready-made list: no test, no
//           compaction), every loaded word folded into one xor so that nothing is dead;
//   BOTH    the two in one kernel, independent of each other: what perfect overlap of this arithmetic with these loads takes.
//
// The per-wave hit lists are prepared once by k_floor_prepare with the ring kernel's own rules (front section, cutoff test,
// in-group pairs evaluated once).  All three kernels use the ring kernel's workgroup (4 waves, 256 atoms), its XCD-contiguous
// order and its LDS footprint, so occupancy is comparable.
#include <cstdio>
#include <algorithm>
#include "engine.h"
#include "kernels.h"
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

#define FL_WAVES 4
#define FL_PITCH 2560           // hit records a wave may hold (64 atoms x 40: rho = 4 has ~11 per atom)

typedef u32 u32x4_ __attribute__((ext_vector_type(4)));
__device__ inline float4 fl_buf_load4(__amdgpu_buffer_rsrc_t r, u32 byte_off)
{
    u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

struct FloorArgs {
    const float4 *coord4, *veloc4;
    const int *count, *table;
    int n_col, n, nall;
    float cutsq, cutinv, a0, gamma, sigma, dtis;
    u32 *hits;          // [waves][FL_PITCH] partner index | shared << 31
    int *nhits;         // [waves]
    double *f[3];
    int *overflow;
    int drop16, keep16;
    int cu_map;         // experiments: 1 = workgroups that the dispatcher places on one CU in its first round take adjacent groups of atoms
    u32 local_mask;     // experiments: 0 = the real partner indices; else partner -> (own group base) + (index & mask): every gather inside a window
};

__device__ inline int fl_block(int nbk, int cu_map = 0)
{
    if (nbk & 7) return (int)blockIdx.x;
    const int x = (int)(blockIdx.x & 7), r = (int)(blockIdx.x >> 3), per = nbk >> 3;
    if (cu_map && (per & 31) == 0) return x * per + (r & 31) * (per >> 5) + (r >> 5);      // (32 CUs per XCD, dealt round robin)
    return x * per + r;
}

// the hits of every wave, by the ring kernel's rules for partitioned rows (LP = 1): a front entry inside the cutoff is evaluated;
// it is shared (evaluated once for both atoms) when the partner lies in the same aligned 256-group and the group is complete
__global__ void __launch_bounds__(64 * FL_WAVES) k_floor_prepare(FloorArgs a)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = fl_block(gridDim.x), base = blk * 256, i = base + w * 64 + lane;
    const bool mine = i < a.n;
    const int gw = blk * FL_WAVES + w;
    float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int n = 0;
    if (mine) { c1 = a.coord4[i]; n = a.count[i]; }
    const bool full = base + 256 <= a.n;
    int nmax = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    int tot = 0;
    const int4 *rows = (const int4 *)a.table + 2 * row_word8(mine ? i : 0, 0, a.n_col);
    for (int ch = 0; ch * 8 < nmax; ch++) {
        int4 w0 = make_int4(0, 0, 0, 0), w1 = w0;
        if (ch * 8 < n) { w0 = rows[(size_t)ch * 128]; w1 = rows[(size_t)ch * 128 + 1]; }
        const int j[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            bool hit = false;
            if (ch * 8 + q < n) {
                const float4 c2 = a.coord4[j[q]];
                const float dx = c1.x - c2.x, dy = c1.y - c2.y, dz = c1.z - c2.z;
                const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                hit = (rsq < a.cutsq) & (rsq >= (float)MESO_EPSILON_SQ);
            }
            const u64 m = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const int pos = tot + (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                const bool sh = full && (((u32)j[q] ^ (u32)i) < 256u);
                if (pos < FL_PITCH) a.hits[(size_t)gw * FL_PITCH + pos] = (u32)j[q] | (sh ? 0x80000000u : 0u);
            }
            tot += __popcll(m);
        }
    }
    if (lane == 0) {
        a.nhits[gw] = min(tot, FL_PITCH);
        if (tot > FL_PITCH) atomicMax(a.overflow, tot);
    }
}

// MODE bit 0: arithmetic, bit 1: the loads, bit 2: the loads with two row chunks of gathers in flight
// EXP: the experiment knobs (local_mask, drop16, cu_map) are compiled in; the plain modes carry none of their instructions
template <int MODE, bool EXP = false>
__global__ void __launch_bounds__(64 * FL_WAVES, 5) k_pair_floor(FloorArgs a)
{
    constexpr bool ARITH = (MODE & 1) != 0, LOADS = (MODE & 2) != 0, DEEP = (MODE & 4) != 0;
    __shared__ float4 own_c[256], own_v[256];
    __shared__ u32 facc[3 * 256];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int blk = fl_block(gridDim.x, EXP ? a.cu_map : 0), base = blk * 256, i = base + w * 64 + lane;
    const bool mine = i < a.n;
    const int gw = blk * FL_WAVES + w;
    float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = c1;
    int n = 0;
    const int4 *rows = (const int4 *)a.table + 2 * row_word8(mine ? i : 0, 0, a.n_col);
    int4 w0 = make_int4(0, 0, 0, 0), w1 = w0;
    if (mine) {
        c1 = a.coord4[i]; v1 = a.veloc4[i]; n = a.count[i];
        if (LOADS && n > 0) { w0 = rows[0]; w1 = rows[1]; }
    }
    // experiments (keep16 in 1..15): only keep16 sixteenths of every row are walked with gathers - whole gather INSTRUCTIONS fewer - and
    // the rest of the row's entries are read from the workgroup's LDS copy instead (what a row section of in-group partners would cost)
    int n_lds = 0;
    if (EXP && a.keep16 > 0 && a.keep16 < 16) { const int nk = (n * a.keep16 + 15) >> 4; n_lds = n - nk; n = nk; }
    const int nh = a.nhits[gw];
    const int ob = w * 64 + lane;
    own_c[ob] = c1; own_v[ob] = v1;
    facc[ob] = 0; facc[256 + ob] = 0; facc[512 + ob] = 0;
    __syncthreads();
    int nmax = n, nsum = n;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nmax = max(nmax, __shfl_xor(nmax, o, 64)); nsum += __shfl_xor(nsum, o, 64); }
    nmax = __builtin_amdgcn_readfirstlane(nmax); nsum = __builtin_amdgcn_readfirstlane(nsum);
    const u32 nrec = (u32)min((unsigned long long)(u32)a.nall * 16ull, 0xFFFFFFFFull);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)a.coord4, 0, (int)nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)a.veloc4, 0, (int)nrec, 0x00020000);
    u32 fold = 0;
    u32 inside = 0;
    if (EXP && LOADS) {
        int lmax = n_lds;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o, 64));
        lmax = __builtin_amdgcn_readfirstlane(lmax);
        for (int k = 0; k < lmax; k++) {
            const float4 p = own_c[(lane * 37 + k * 101 + w * 64) & 255];
            if (k < n_lds) fold ^= __float_as_uint(p.x) ^ __float_as_uint(p.y) ^ __float_as_uint(p.z) ^ __float_as_uint(p.w);
        }
    }
    const u32 lmask = EXP ? a.local_mask : 0u, lbase = (u32)base;
    const u32 drop = EXP ? (u32)a.drop16 : 0u;
    // (drop16: experiments - that many sixteenths of the entries, picked by a hash of the partner index, are not fetched: out-of-range offset)
    auto jx = [&](u32 j) -> u32 { if (!EXP) return j; if (drop && ((j * 2654435761u) >> 28) < drop) return 0x0FFFFFFFu; return lmask ? lbase + (j & lmask) : j; };

    // ---- per row entry: gather + distance test
    auto tests = [&](int k) {       // 64 full-lane tests on LDS operands: 3 sub, 3 mul/fma, 2 compares - the mandatory part of a cutoff test
        const float4 p = own_c[(w * 64 + ((lane * 5 + k * 17) & 63))];
        const float dx = c1.x - p.x, dy = c1.y - p.y, dz = c1.z - p.z;
        const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        inside += ((rsq < a.cutsq) & (rsq >= (float)MESO_EPSILON_SQ)) ? 1u : 0u;
    };
    if (LOADS && DEEP) {
        // two chunks of gathers in flight: the 8 gathers of chunk c + 1 are out before chunk c is folded (16 outstanding per lane)
        const int nch = (n + 7) >> 3, nchmax = (nmax + 7) >> 3;
        int ntest = 0;
        int4 x0 = make_int4(0, 0, 0, 0), x1 = x0;
        if (1 < nch) { x0 = rows[(size_t)128]; x1 = rows[(size_t)128 + 1]; }
        float4 cA[8];
        {
            const int j[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int q = 0; q < 8; q++) cA[q] = fl_buf_load4(rc, 0 < nch ? (jx((u32)j[q]) << 4) : 0xFFFFFFF0u);
        }
        for (int ch = 0; ch < nchmax; ch++) {
            int4 y0 = make_int4(0, 0, 0, 0), y1 = y0;
            if (ch + 2 < nch) { y0 = rows[(size_t)(ch + 2) * 128]; y1 = rows[(size_t)(ch + 2) * 128 + 1]; }
            const int j[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            float4 cB[8];
#pragma unroll
            for (int q = 0; q < 8; q++) cB[q] = fl_buf_load4(rc, ch + 1 < nch ? (jx((u32)j[q]) << 4) : 0xFFFFFFF0u);
            if (ARITH) {
                const int upto = (int)(((long)(nsum + 63) / 64) * (ch + 1) / nchmax);
                for (; ntest < upto; ntest++) tests(ntest);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                fold ^= __float_as_uint(cA[q].x) ^ __float_as_uint(cA[q].y) ^ __float_as_uint(cA[q].z) ^ __float_as_uint(cA[q].w);
                cA[q] = cB[q];
            }
            x0 = y0; x1 = y1;
        }
    } else if (LOADS) {
        const int nch = (n + 7) >> 3, nchmax = (nmax + 7) >> 3;
        int ntest = 0;
        for (int ch = 0; ch < nchmax; ch++) {
            const int4 v0 = w0, v1_ = w1;
            w0 = make_int4(0, 0, 0, 0); w1 = w0;
            if (ch + 1 < nch) { w0 = rows[(size_t)(ch + 1) * 128]; w1 = rows[(size_t)(ch + 1) * 128 + 1]; }
            const int j[8] = {v0.x, v0.y, v0.z, v0.w, v1_.x, v1_.y, v1_.z, v1_.w};
            float4 c2[8];
#pragma unroll
            for (int q = 0; q < 8; q++) c2[q] = fl_buf_load4(rc, ch < nch ? (jx((u32)j[q]) << 4) : 0xFFFFFFF0u);
            if (ARITH) {
                // this wave's share of the full-lane tests, spread over its chunks (independent of the loads in flight)
                const int upto = (int)(((long)(nsum + 63) / 64) * (ch + 1) / nchmax);
                for (; ntest < upto; ntest++) tests(ntest);
            }
#pragma unroll
            for (int q = 0; q < 8; q++) fold ^= __float_as_uint(c2[q].x) ^ __float_as_uint(c2[q].y) ^ __float_as_uint(c2[q].z) ^ __float_as_uint(c2[q].w);
        }
    } else if (ARITH) {
        const int nt = (nsum + 63) / 64;
        for (int k = 0; k < nt; k++) tests(k);
    }

    // ---- per evaluated pair: velocity gather + evaluation, 64 at a time (lane = pair)
    const int nb = (nh + 63) >> 6;
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    u32 rec = 0;
    auto request = [&](int b) {
        const int k = b * 64 + lane;
        rec = 0;
        pv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < nh) {
            rec = a.hits[(size_t)gw * FL_PITCH + k];
            // (in-group partners: velocity record from the workgroup's LDS copy, as the ring kernel does at this size)
            if (rec & 0x80000000u) pv = own_v[(rec & 0x7FFFFFFFu) - (u32)base];
            else pv = fl_buf_load4(rv, (rec & 0x7FFFFFFFu) << 4);
        }
    };
    auto evaluate = [&](int b) {
        // the ring kernel's evaluation of one pair (FAST, one type, s = 1, TEA Gaussian), operands from LDS
        const int o = (lane * 3 + b) & 63, p = (lane * 7 + b * 5 + 1) & 63;
        const float4 ci = own_c[w * 64 + o], vi = own_v[w * 64 + o], cj = own_c[w * 64 + p], vj = own_v[w * 64 + p];
        const float dx = ci.x - cj.x, dy = ci.y - cj.y, dz = ci.z - cj.z;
        const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) + 1.0e-3f;
        const float rn = gaussian_tea_fast(__float_as_uint(vi.w), __float_as_uint(vj.w));
        const float rinv = __builtin_amdgcn_rsqf(rsq);
        const float r = rsq * rinv;
        const float dvx = vi.x - vj.x, dvy = vi.y - vj.y, dvz = vi.z - vj.z;
        const float dot = __builtin_fmaf(dz, dvz, __builtin_fmaf(dy, dvy, dx * dvx));
        const float wc = __builtin_fmaf(-r, a.cutinv, 1.0f);
        const float fcons = a.a0 * wc;
        float fpair = __builtin_fmaf(a.sigma * wc * rn, a.dtis, fcons - (a.gamma * wc * wc * dot * rinv));
        fpair *= rinv;
        const u32 qx = to_fixed16(dx * fpair), qy = to_fixed16(dy * fpair), qz = to_fixed16(dz * fpair);
        __hip_atomic_fetch_add(&facc[w * 64 + o], qx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&facc[256 + w * 64 + o], qy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&facc[512 + w * 64 + o], qz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane & 1) {        // (about half of the evaluated pairs are shared: the partner receives the opposite force)
            __hip_atomic_fetch_sub(&facc[w * 64 + p], qx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_sub(&facc[256 + w * 64 + p], qy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_sub(&facc[512 + w * 64 + p], qz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    if (LOADS) {
        // (the batch requested one iteration earlier is folded while the next one is in flight - the ring kernel's pipelining)
        if (nb > 0) request(0);
        for (int b = 0; b < nb; b++) {
            const float4 got = pv;
            const u32 grec = rec;
            if (b + 1 < nb) request(b + 1);
            if (ARITH) evaluate(b);
            fold ^= __float_as_uint(got.x) ^ __float_as_uint(got.y) ^ __float_as_uint(got.z) ^ __float_as_uint(got.w) ^ grec;
        }
    } else if (ARITH) {
        for (int b = 0; b < nb; b++) evaluate(b);
    }
    __syncthreads();
    if (mine) {
        // (the fold and the test count keep every load and every test alive; they change the stored number by at most one unit of the
        // fixed-point scale and only when all their bits happen to be set)
        const double eps = ((!LOADS || fold == 0xFFFFFFFFu) && (!ARITH || inside == 0xFFFFFFFFu)) ? 1.0 : 0.0;
        a.f[0][i] = from_fixed16(facc[ob]) + eps; a.f[1][i] = from_fixed16(facc[256 + ob]); a.f[2][i] = from_fixed16(facc[512 + ob]);
    }
}

// What an INCREMENTAL list build would cost at the least (profiles/r06_notes.md section 2): every atom walks a stored row
// - here its front and back sections, 36 entries: the stand-in for the 46 of a wider previous row - gathers each candidate's 16-byte
// record, tests it against the list cutoff and writes the row back in the same two sections, compaction and classification for free
// (a rejected entry becomes the atom itself in place).  No index translation, no ghosts' identity, no count-in-front pass.
__global__ void __launch_bounds__(64 * FL_WAVES, 5) k_refilter_floor(FloorArgs a, const int *__restrict__ nback, const int *__restrict__ back, int nb_col,
                                                                     int *__restrict__ out_f, int *__restrict__ out_b, int *__restrict__ out_n, float rlist2)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = fl_block(gridDim.x), i = blk * 256 + w * 64 + lane;
    const bool mine = i < a.n;
    float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int nsec[2] = {0, 0};
    if (mine) { c1 = a.coord4[i]; nsec[0] = a.count[i]; nsec[1] = nback[i]; }
    const u32 nrec = (u32)min((unsigned long long)(u32)a.nall * 16ull, 0xFFFFFFFFull);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)a.coord4, 0, (int)nrec, 0x00020000);
    int kept = 0;
#pragma unroll 1
    for (int sec = 0; sec < 2; sec++) {
        const int pitch = sec ? nb_col : a.n_col, n = nsec[sec];
        const int4 *rows = (const int4 *)(sec ? back : a.table) + 2 * row_word8(mine ? i : 0, 0, pitch);
        int4 *outr = (int4 *)(sec ? out_b : out_f) + 2 * row_word8(mine ? i : 0, 0, pitch);
        int nmax = n;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        const int nch = (n + 7) >> 3, nchmax = (__builtin_amdgcn_readfirstlane(nmax) + 7) >> 3;
        int4 w0 = make_int4(0, 0, 0, 0), w1 = w0;
        if (nch > 0) { w0 = rows[0]; w1 = rows[1]; }
#pragma unroll 1
        for (int ch = 0; ch < nchmax; ch++) {
            const int4 v0 = w0, v1 = w1;
            w0 = make_int4(0, 0, 0, 0); w1 = w0;
            if (ch + 1 < nch) { w0 = rows[(size_t)(ch + 1) * 128]; w1 = rows[(size_t)(ch + 1) * 128 + 1]; }
            int j[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            float4 c2[8];
#pragma unroll
            for (int q = 0; q < 8; q++) c2[q] = fl_buf_load4(rc, ch < nch ? ((u32)j[q] << 4) : 0xFFFFFFF0u);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float dx = c1.x - c2[q].x, dy = c1.y - c2[q].y, dz = c1.z - c2[q].z;
                const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                const bool keep = (rsq <= rlist2) & (ch * 8 + q < n) & (j[q] != i);
                kept += keep ? 1 : 0;
                j[q] = keep ? j[q] : i;
            }
            if (ch < nch) {
                outr[(size_t)ch * 128] = make_int4(j[0], j[1], j[2], j[3]);
                outr[(size_t)ch * 128 + 1] = make_int4(j[4], j[5], j[6], j[7]);
            }
        }
    }
    if (mine) out_n[i] = kept;
}

// mode 1: arithmetic, 2: loads, 3: both; us = mean over reps launches (HIP events on the engine's stream); counts[0..1]: row entries
// walked, pairs evaluated.  Needs a built neighbour table with rows in two sections and one
// atom type; writes the force arrays (call it outside a run: the next setup / compute overwrites them).
int Engine::pair_floor(int mode, int reps, double *us, long *counts)
{
    u32 local_mask = 0;
    int drop16 = 0, keep16 = 0;
    if (mode >= 1000000) { keep16 = mode / 1000000; mode %= 1000000; }      // 1000000 k + m: mode m walking k sixteenths of every row with gathers, the rest from LDS
    if (mode >= 10000) { drop16 = mode / 10000; mode %= 10000; }      // 10000 d + m: mode m with d sixteenths of the coordinate gathers not fetched
    if (mode >= 100) { local_mask = (1u << (mode / 100)) - 1u; mode %= 100; }
    int cu_map = 0;
    if (mode >= 50) { cu_map = 1; mode -= 50; }      // 50 + m: mode m with CU-local groups      // experiments: 800 + m = mode m with every gather inside a 256-atom window
    if (mode == 8) {
        // the least an incremental list build costs (k_refilter_floor): scratch tables of the stored tables' sizes
        if (!is_setup || !pair_table || !pair_back || !rows_part || nlocal <= 0) return fail(3, "pair_floor 8: needs a built table with rows in two sections");
        const int nblk = ((nlocal + 255) / 256 + 7) / 8 * 8;
        int *of = nullptr, *ob = nullptr, *on = nullptr;
        HIPCHK(hipMalloc(&of, table_tiles * 64 * (size_t)n_col * sizeof(int)));
        HIPCHK(hipMalloc(&ob, table_tiles * 64 * (size_t)nb_col * sizeof(int)));
        HIPCHK(hipMalloc(&on, (size_t)nmax * sizeof(int)));
        FloorArgs a = {};
        a.coord4 = coord4; a.veloc4 = veloc4; a.count = pair_count; a.table = pair_table; a.n_col = n_col; a.n = nlocal;
        a.nall = (int)std::min<long>((long)nlocal + nghost, (1L << 28) - 1);
        const float rl2 = (float)((cutmax + skin) * (cutmax + skin));
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_refilter_floor, dim3(nblk), dim3(64 * FL_WAVES), 0, stream, a, pair_nback, pair_back, nb_col, of, ob, on, rl2);
        (void)hipEventRecord(e0, stream);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_refilter_floor, dim3(nblk), dim3(64 * FL_WAVES), 0, stream, a, pair_nback, pair_back, nb_col, of, ob, on, rl2);
        (void)hipEventRecord(e1, stream);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *us = 1.0e3 * (double)ms / reps;
        if (counts) {
            std::vector<int> hn((size_t)nlocal);
            HIPCHK(hipMemcpy(hn.data(), on, hn.size() * sizeof(int), hipMemcpyDeviceToHost));
            long k = 0;
            for (int q = 0; q < nlocal; q++) k += hn[q];
            counts[0] = k; counts[1] = 0;
        }
        (void)hipFree(of); (void)hipFree(ob); (void)hipFree(on);
        return hipGetLastError() == hipSuccess ? 0 : fail(2, "pair_floor 8: launch failed");
    }
    if (!(mode >= 1 && mode <= 3) && mode != 6 && mode != 7) return fail(1, "pair_floor: mode 1 (arithmetic), 2 (loads), 3 (both); 6 / 7: 2 / 3 with two chunks of gathers in flight");
    if (reps < 1 || !us) return fail(1, "pair_floor: invalid arguments");
    if ((local_mask || drop16 || cu_map || keep16) && mode != 2 && mode != 6) return fail(1, "pair_floor: the experiment knobs go with the loads-only modes 2 and 6");
    if (!is_setup || !pair_table || nlocal <= 0) return fail(3, "pair_floor: no neighbour table (run setup first)");
    if (!rows_part || part_group != 256 || ntypes != 1) return fail(3, "pair_floor: needs rows in two sections for 256-atom groups and one atom type");
    TRY(resolve_counts());
    const int nblk = ((nlocal + 255) / 256 + 7) / 8 * 8, nw = nblk * FL_WAVES;
    u32 *hits = nullptr;
    int *nh = nullptr, *ovf = nullptr;
    HIPCHK(hipMalloc(&hits, (size_t)nw * FL_PITCH * sizeof(u32)));
    HIPCHK(hipMalloc(&nh, (size_t)(nw + 1) * sizeof(int)));
    ovf = nh + nw;
    HIPCHK(hipMemsetAsync(nh, 0, (size_t)(nw + 1) * sizeof(int), stream));
    FloorArgs a;
    a.coord4 = coord4; a.veloc4 = veloc4; a.count = pair_count; a.table = pair_table; a.n_col = n_col; a.n = nlocal;
    a.nall = (int)std::min<long>((long)nlocal + nghost, (1L << 28) - 1);
    a.cutsq = (float)coeff[P_CUTSQ]; a.cutinv = (float)coeff[P_CUTINV]; a.a0 = (float)coeff[P_A0]; a.gamma = (float)coeff[P_GAMMA];
    a.sigma = (float)coeff[P_SIGMA]; a.dtis = (float)(1.0 / std::sqrt(dt));
    a.hits = hits; a.nhits = nh; a.overflow = ovf; a.local_mask = local_mask; a.cu_map = cu_map; a.drop16 = drop16; a.keep16 = keep16;
    for (int d = 0; d < 3; d++) a.f[d] = cur.f[d];
    hipLaunchKernelGGL(k_floor_prepare, dim3(nblk), dim3(64 * FL_WAVES), 0, stream, a);
    std::vector<int> hn((size_t)nw + 1);
    HIPCHK(hipMemcpyAsync(hn.data(), nh, hn.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    int rc = 0;
    if (hn[nw]) rc = fail(4, "pair_floor: a wave holds more evaluated pairs than its list (density far above rho = 4)");
    if (!rc && counts) {
        long ne = 0;
        for (int k = 0; k < nw; k++) ne += hn[k];
        std::vector<int> hc((size_t)nlocal);
        HIPCHK(hipMemcpy(hc.data(), pair_count, hc.size() * sizeof(int), hipMemcpyDeviceToHost));
        long nt = 0;
        for (int k = 0; k < nlocal; k++) nt += hc[k];
        counts[0] = nt; counts[1] = ne;
    }
    if (!rc) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const bool ex = local_mask || drop16 || cu_map || keep16;
        auto launch = [&]() {
            const dim3 g(nblk), b(64 * FL_WAVES);
            if (ex) {
                if (mode == 2) hipLaunchKernelGGL((k_pair_floor<2, true>), g, b, 0, stream, a);
                else hipLaunchKernelGGL((k_pair_floor<6, true>), g, b, 0, stream, a);
            } else if (mode == 1) hipLaunchKernelGGL((k_pair_floor<1>), g, b, 0, stream, a);
            else if (mode == 2) hipLaunchKernelGGL((k_pair_floor<2>), g, b, 0, stream, a);
            else if (mode == 3) hipLaunchKernelGGL((k_pair_floor<3>), g, b, 0, stream, a);
            else if (mode == 6) hipLaunchKernelGGL((k_pair_floor<6>), g, b, 0, stream, a);
            else hipLaunchKernelGGL((k_pair_floor<7>), g, b, 0, stream, a);
        };
        for (int r = 0; r < 3; r++) launch();
        (void)hipEventRecord(e0, stream);
        for (int r = 0; r < reps; r++) launch();
        (void)hipEventRecord(e1, stream);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *us = 1.0e3 * (double)ms / reps;
        if (hipGetLastError() != hipSuccess) rc = fail(2, "pair_floor: launch failed");
    }
    (void)hipFree(hits); (void)hipFree(nh);
    return rc;
}

} // namespace meso
