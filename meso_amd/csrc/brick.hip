// Brick kernels: the MI355X-first layout of the DPD hot path.
//
// Local atoms are sorted by [border bit][Morton(bin)][Morton(sub-cell)] (atom_meso.cu:268-308), so every
// aligned group of 32 Morton codes is a 4x4x2 brick of bins whose atoms are contiguous in memory, and ghosts
// are sorted by Morton(bin) behind them.  One workgroup owns one (section, brick): it copies the brick's
// 6x6x4-bin halo (~1270 atoms at rho=4) from HBM into LDS ONCE, as whole bin runs, and every neighbour
// gather of the force kernel and of the list builder then hits LDS instead of L2 (the lane-per-atom kernel
// measured 48 % L1 / 54 % L2 hit rates and 4x the algorithmic bytes at the fabric: profiles/r01_pmc_*).
// Neighbour rows hold 16-bit halo-local indices (half the table traffic of the reference's int rows).
// Row layout stays transposed per 64-atom tile: entry p of atom i at table16[((i>>6)*n_col + p)*64 + (i&63)].
//
// Replaces gpu_build_neighbor_list + gpu_join/transpose (neigh_build_meso.cu:20-240) and gpu_dpd /
// gpu_dpd_fast (pair_dpd_meso.cu:91-205, pair_dpd_fast_meso.cu:91-205); membership test and per-pair
// arithmetic are unchanged, so results agree with the lane-per-atom kernels to summation order.
#include "kernels.h"
#include "meso_device.h"
#include <type_traits>

namespace meso {

#define BRK_HX 6
#define BRK_HY 6
#define BRK_HZ 4
#define BRK_NHB (BRK_HX * BRK_HY * BRK_HZ)
#define BRK_THREADS 320
#define BRK_WAVES (BRK_THREADS / 64)
#define BRK_MAXH 1664          // halo atoms staged per brick (mean 1267 at rho=4, sigma ~36)
#define BRK_MAXOWN 512
#define BRK_RING 128

struct BrickHdr {
    int hoff[BRK_NHB + 1];
    int hs0[BRK_NHB], hl0[BRK_NHB], hs1[BRK_NHB], hl1[BRK_NHB], hs2[BRK_NHB];
    int ostart[33];
    int wtot[4];
    int nh, n_own, o0, sec;
};

// Row layout: 8 consecutive 16-bit entries of one atom are one 16-byte word; a wave reads 64 such words
// (1 KiB, fully coalesced) per 8 candidates:  word(i, c) = ((i>>6)*(n_col/8) + c)*64 + (i&63),  c = p>>3.
__device__ inline size_t row_word(int i, int c, int n_col) { return ((size_t)(i >> 6) * (n_col >> 3) + c) * 64 + (i & 63); }

__device__ inline float dist2(float4 a, float4 b)
{
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

__device__ inline u32 compact3(u32 x)
{
    x &= 0x09249249;
    x = (x ^ (x >> 2)) & 0x030c30c3;
    x = (x ^ (x >> 4)) & 0x0300f00f;
    x = (x ^ (x >> 8)) & 0xff0000ff;
    x = (x ^ (x >> 16)) & 0x000003ff;
    return x;
}

// returns false (block-uniform) when the brick owns no atoms
__device__ inline bool brick_setup(const BrickArgs &g, BrickHdr &H, int *overflow)
{
    const int tid = threadIdx.x;
    // XCD-aware order: blocks b and b+8 share an L2, so each XCD walks a contiguous run of the active list
    const int nb2 = gridDim.x;
    int slot = (nb2 & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (nb2 >> 3) + (blockIdx.x >> 3));
    if (slot >= g.nactive) return false;
    const int blk = g.active[slot];
    const int sec = blk / g.nbricks, B = blk % g.nbricks;
    if (tid <= 32) H.ostart[tid] = g.estart[(size_t)sec * g.M + 32 * B + tid];
    __syncthreads();
    const int o0 = H.ostart[0], n_own = H.ostart[32] - o0;
    if (n_own <= 0) return false;
    const u32 code0 = 32u * (u32)B;
    const int bx0 = (int)compact3(code0), by0 = (int)compact3(code0 >> 1), bz0 = (int)compact3(code0 >> 2);
    int tot = 0;
    if (tid < BRK_NHB) {
        int hx = bx0 - 1 + tid % BRK_HX, hy = by0 - 1 + (tid / BRK_HX) % BRK_HY, hz = bz0 - 1 + tid / (BRK_HX * BRK_HY);
        int s0 = 0, l0 = 0, s1 = 0, l1 = 0, s2 = 0, l2 = 0;
        if (hx >= 0 && hx < g.mbin[0] && hy >= 0 && hy < g.mbin[1] && hz >= 0 && hz < g.mbin[2]) {
            u32 m = interleave3((u32)hx, (u32)hy, (u32)hz);
            s0 = g.estart[m]; l0 = g.estart[m + 1] - s0;
            s1 = g.estart[(size_t)g.M + m]; l1 = g.estart[(size_t)g.M + m + 1] - s1;
            s2 = g.ghost_base + g.gstart[m]; l2 = g.gstart[m + 1] - g.gstart[m];
        }
        H.hs0[tid] = s0; H.hl0[tid] = l0; H.hs1[tid] = s1; H.hl1[tid] = l1; H.hs2[tid] = s2;
        tot = l0 + l1 + l2;
    }
    // exclusive scan of tot over the first 3 waves
    int incl = tot;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63 && w < 3) H.wtot[w] = incl;
    __syncthreads();
    if (tid < BRK_NHB) {
        int base = 0;
        for (int k = 0; k < w; k++) base += H.wtot[k];
        H.hoff[tid] = base + incl - tot;
        if (tid == BRK_NHB - 1) H.hoff[BRK_NHB] = base + incl;
    }
    __syncthreads();
    const int nh = H.hoff[BRK_NHB];
    if (nh > BRK_MAXH || n_own > BRK_MAXOWN) {
        if (tid == 0) atomicMax(overflow, 100000 + (nh > BRK_MAXH ? nh : n_own));
        return false;
    }
    if (tid == 0) { H.nh = nh; H.n_own = n_own; H.o0 = o0; H.sec = sec; }
    __syncthreads();
    return true;
}

// halo-local index -> global (cell-order) atom index
__device__ inline int halo_src(const BrickHdr &H, int h)
{
    int lo = 0, hi = BRK_NHB;            // largest hb with hoff[hb] <= h
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (H.hoff[mid] <= h) lo = mid; else hi = mid;
    }
    int off = h - H.hoff[lo];
    if (off < H.hl0[lo]) return H.hs0[lo] + off;
    off -= H.hl0[lo];
    if (off < H.hl1[lo]) return H.hs1[lo] + off;
    return H.hs2[lo] + (off - H.hl1[lo]);
}

// own atom o (0-based inside the brick section) -> its brick-local bin k (0..31) and halo-local index
__device__ inline int own_loc(const BrickHdr &H, int o, int &hb_out)
{
    const int i = H.o0 + o;
    int lo = 0, hi = 32;                 // largest k with ostart[k] <= i
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (H.ostart[mid] <= i) lo = mid; else hi = mid;
    }
    const int k = lo;
    const int kx = (k & 1) | (((k >> 3) & 1) << 1), ky = ((k >> 1) & 1) | (((k >> 4) & 1) << 1), kz = (k >> 2) & 1;
    const int hb = (kx + 1) + BRK_HX * ((ky + 1) + BRK_HY * (kz + 1));
    hb_out = hb;
    return H.hoff[hb] + (H.sec ? H.hl0[hb] : 0) + (i - H.ostart[k]);
}

// =========================================================================================
// neighbour table builder
// =========================================================================================
__global__ void __launch_bounds__(BRK_THREADS) k_brick_build(BrickArgs g, const float4 *__restrict__ coord4,
                                                            float rc2, int n_col, int *__restrict__ count,
                                                            unsigned short *__restrict__ table16,
                                                            int *__restrict__ overflow)
{
    __shared__ BrickHdr H;
    __shared__ float4 hc[BRK_MAXH];
    if (!brick_setup(g, H, overflow)) return;
    const int tid = threadIdx.x;
    for (int h = tid; h < H.nh; h += BRK_THREADS) hc[h] = coord4[halo_src(H, h)];
    __syncthreads();
    for (int o = tid; o < H.n_own; o += BRK_THREADS) {
        int hb;
        const int loc = own_loc(H, o, hb);
        const int i = H.o0 + o;
        const float4 ci = hc[loc];
        uint4 *rows = (uint4 *)table16;
        int n = 0;
        u64 lo = 0, hi = 0;
        auto push = [&](int k) {
            int q = n & 7;
            if (q < 4) lo |= (u64)(u32)k << (16 * q);
            else hi |= (u64)(u32)k << (16 * (q - 4));
            if (q == 7) {
                if (n < n_col) rows[row_word(i, n >> 3, n_col)] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
                lo = 0; hi = 0;
            }
            n++;
        };
#pragma unroll 1
        for (int r = 0; r < 9; r++) {
            // x-adjacent halo bins are consecutive halo-bin indices: one contiguous run of halo-local slots
            const int hrow = hb + (r % 3 - 1) * BRK_HX + (r / 3 - 1) * BRK_HX * BRK_HY;
            const int kb = H.hoff[hrow - 1], ke = H.hoff[hrow + 2];
            int k = kb;
            for (; k + 4 <= ke; k += 4) {       // 4 LDS gathers in flight per lane
                float4 c0 = hc[k], c1 = hc[k + 1], c2 = hc[k + 2], c3 = hc[k + 3];
                float d0 = dist2(ci, c0), d1 = dist2(ci, c1), d2 = dist2(ci, c2), d3 = dist2(ci, c3);
                if (k != loc && d0 <= rc2) push(k);
                if (k + 1 != loc && d1 <= rc2) push(k + 1);
                if (k + 2 != loc && d2 <= rc2) push(k + 2);
                if (k + 3 != loc && d3 <= rc2) push(k + 3);
            }
            for (; k < ke; k++) {
                float d0 = dist2(ci, hc[k]);
                if (k != loc && d0 <= rc2) push(k);
            }
        }
        if ((n & 7) && n < n_col) rows[row_word(i, n >> 3, n_col)] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
        if (n > n_col) { atomicMax(overflow, n); n = n_col; }
        count[i] = n;
    }
}

// halo-local rows -> global-index rows (for the lane-per-atom kernels, energy/virial steps and the tests)
__global__ void __launch_bounds__(BRK_THREADS) k_brick_convert(BrickArgs g, int n_col, const int *__restrict__ count,
                                                              const unsigned short *__restrict__ table16,
                                                              int *__restrict__ table32, int *__restrict__ overflow)
{
    __shared__ BrickHdr H;
    if (!brick_setup(g, H, overflow)) return;
    for (int o = threadIdx.x; o < H.n_own; o += BRK_THREADS) {
        const int i = H.o0 + o;
        const size_t base = ((size_t)(i >> 6) * n_col) * 64 + (i & 63);
        const int n = count[i];
        for (int p = 0; p < n; p++)
            table32[base + (size_t)p * 64] = halo_src(H, (int)table16[row_word(i, p >> 3, n_col) * 8 + (p & 7)]);
    }
}

// =========================================================================================
// pair force
// =========================================================================================
template <bool FAST>
__global__ void __launch_bounds__(BRK_THREADS) k_brick_pair(BrickArgs g, PairArgs a,
                                                           const unsigned short *__restrict__ table16,
                                                           int *__restrict__ overflow)
{
    // fp32 style: 64-bit fixed-point sums (ds_add_u64; float LDS atomics serialise on gfx950, see pair_ring.hip)
    typedef typename std::conditional<FAST, u64, double>::type acc_t;
    __shared__ BrickHdr H;
    __shared__ float4 hc[BRK_MAXH];
    __shared__ float4 hv[BRK_MAXH];
    __shared__ int ring[BRK_WAVES][BRK_RING];
    __shared__ unsigned short oloc[BRK_MAXOWN];
    __shared__ acc_t facc[3][BRK_MAXOWN];
    extern __shared__ double cf_dyn[];
    double *cf64 = cf_dyn;
    float *cf32 = (float *)cf_dyn;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ncf = a.ntypes * a.ntypes * N_COEFF;
    for (int p = tid; p < ncf; p += BRK_THREADS) {
        if (FAST) cf32[p] = a.coeff32[p];
        else cf64[p] = a.coeff64[p];
    }
    if (!brick_setup(g, H, overflow)) return;
    // work-range filter (compute_bulk / compute_border): section 0 = bulk, 1 = border
    if (H.o0 >= a.end || H.o0 + H.n_own <= a.beg) return;
    for (int h = tid; h < H.nh; h += BRK_THREADS) {
        int src = halo_src(H, h);
        hc[h] = a.coord4[src];
        hv[h] = a.veloc4[src];
    }
    for (int o = tid; o < H.n_own; o += BRK_THREADS) {
        int hb;
        oloc[o] = (unsigned short)own_loc(H, o, hb);
        facc[0][o] = 0; facc[1][o] = 0; facc[2][o] = 0;
    }
    __syncthreads();
    if (a.debug == 1) return;

    const u64 lt = (1ULL << lane) - 1ULL;
    const float dtis32 = (float)a.dt_inv_sqrt;
    int *myring = ring[w];

    for (int obase = w * 64; obase < H.n_own; obase += BRK_WAVES * 64) {
        const int o = obase + lane;
        const bool mine = o < H.n_own;
        const int i = H.o0 + o;
        int n = 0, loc = 0;
        float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mine) { loc = oloc[o]; c1 = hc[loc]; n = a.count[i]; }
        const u32 t1 = __float_as_uint(c1.w);
        const uint4 *rows = (const uint4 *)table16;
        int nmax = n;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) nmax = max(nmax, __shfl_xor(nmax, s, 64));
        int qhead = 0, qtail = 0;

        auto drain = [&](int nb) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < nb && a.debug != 2) {
                int pk = myring[(qhead + lane) & (BRK_RING - 1)];
                int j = pk & 0xFFFF, oo = pk >> 16;
                int li = oloc[oo];
                float4 ci = hc[li], vi = hv[li], cj = hc[j], vj = hv[j];
                u32 si = __float_as_uint(vi.w), sj = __float_as_uint(vj.w);
                int cidx = __float_as_uint(ci.w) * a.ntypes + __float_as_uint(cj.w);
                if (FAST) {
                    const float *cf = cf32 + cidx * N_COEFF;
                    float dx = ci.x - cj.x, dy = ci.y - cj.y, dz = ci.z - cj.z;
                    float rsq = dx * dx + dy * dy + dz * dz;
                    float rn = gaussian_tea_fast(si, sj);
                    float rinv = __builtin_amdgcn_rsqf(rsq);
                    float r = rsq * rinv;
                    float dvx = vi.x - vj.x, dvy = vi.y - vj.y, dvz = vi.z - vj.z;
                    float dot = dx * dvx + dy * dvy + dz * dvz;
                    float wc = 1.0f - r * cf[P_CUTINV];
                    float ew = cf[P_EXPW];
                    float wr = (ew == 1.0f) ? wc : __powf(wc, ew);
                    float fpair = cf[P_A0] * wc - (cf[P_GAMMA] * wr * wr * dot * rinv) + (cf[P_SIGMA] * wr * rn * dtis32);
                    fpair *= rinv;
                    __hip_atomic_fetch_add((u64 *)&facc[0][oo], to_fixed(dx * fpair), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add((u64 *)&facc[1][oo], to_fixed(dy * fpair), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add((u64 *)&facc[2][oo], to_fixed(dz * fpair), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
                    const double *cf = cf64 + cidx * N_COEFF;
                    double dx = (double)ci.x - (double)cj.x, dy = (double)ci.y - (double)cj.y, dz = (double)ci.z - (double)cj.z;
                    double rsq = dx * dx + dy * dy + dz * dz;
                    double rn = gaussian_tea(si, sj);
                    double rinv = rsqrt(rsq);
                    double r = rsq * rinv;
                    double dvx = (double)vi.x - (double)vj.x, dvy = (double)vi.y - (double)vj.y, dvz = (double)vi.z - (double)vj.z;
                    double dot = dx * dvx + dy * dvy + dz * dvz;
                    double wc = 1.0 - r * cf[P_CUTINV];
                    double ew = cf[P_EXPW];
                    double wr = (ew == 1.0) ? wc : powd_poly(wc, ew);
                    double fpair = cf[P_A0] * wc - (cf[P_GAMMA] * wr * wr * dot * rinv) + (cf[P_SIGMA] * wr * rn * a.dt_inv_sqrt);
                    fpair *= rinv;
                    __hip_atomic_fetch_add((double *)&facc[0][oo], dx * fpair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add((double *)&facc[1][oo], dy * fpair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add((double *)&facc[2][oo], dz * fpair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            qhead += nb;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };

        const int nchunk = (nmax + 7) >> 3;
        uint4 wcur = make_uint4(0, 0, 0, 0);
        if (mine && n > 0) wcur = rows[row_word(i, 0, a.n_col)];
        for (int c = 0; c < nchunk; c++) {
            // prefetch the next 8 entries of my row while this chunk is tested
            uint4 wnext = make_uint4(0, 0, 0, 0);
            if (mine && (c + 1) * 8 < n) wnext = rows[row_word(i, c + 1, a.n_col)];
            const u32 ww[4] = {wcur.x, wcur.y, wcur.z, wcur.w};
            int jj[8];
            float4 cc[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                jj[q] = (int)((ww[q >> 1] >> (16 * (q & 1))) & 0xFFFFu);
                cc[q] = hc[jj[q]];            // entries past my count are 0: a valid slot, masked below
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const bool active = c * 8 + q < n;
                const float4 c2 = cc[q];
                const int cidx = t1 * a.ntypes + __float_as_uint(c2.w);
                bool hit;
                if (FAST) {
                    float dx = c1.x - c2.x, dy = c1.y - c2.y, dz = c1.z - c2.z;
                    float rsq = dx * dx + dy * dy + dz * dz;
                    hit = active && rsq < cf32[cidx * N_COEFF + P_CUTSQ] && rsq >= (float)MESO_EPSILON_SQ;
                } else {
                    double dx = (double)c1.x - (double)c2.x, dy = (double)c1.y - (double)c2.y, dz = (double)c1.z - (double)c2.z;
                    double rsq = dx * dx + dy * dy + dz * dz;
                    hit = active && rsq < cf64[cidx * N_COEFF + P_CUTSQ] && rsq >= MESO_EPSILON_SQ;
                }
                const u64 m = __ballot(hit);
                if (m) {
                    if (hit) myring[(qtail + __popcll(m & lt)) & (BRK_RING - 1)] = jj[q] | (o << 16);
                    qtail += __popcll(m);
                    if (qtail - qhead >= 64) drain(64);
                }
            }
            wcur = wnext;
        }
        if (qtail > qhead) drain(qtail - qhead);

        if (mine && i >= a.beg && i < a.end) {
            double fx, fy, fz;
            if (FAST) { fx = from_fixed((u64)facc[0][o]); fy = from_fixed((u64)facc[1][o]); fz = from_fixed((u64)facc[2][o]); }
            else { fx = (double)facc[0][o]; fy = (double)facc[1][o]; fz = (double)facc[2][o]; }
            if (a.accumulate) { a.f[0][i] += fx; a.f[1][i] += fy; a.f[2][i] += fz; }
            else { a.f[0][i] = fx; a.f[1][i] = fy; a.f[2][i] = fz; }
        }
    }
}

// =========================================================================================
// cell structure kernels
// =========================================================================================
// estart[e] = first local index whose extended code (sorted reorder key >> 12) is >= e, e in [0, ncodes]
// (one thread per code, binary search: codes of the border section and of the ghosts are sparse, so the
// "fill the gap" formulation of gpu_find_bin_boundary would serialise on single threads)
template <typename K>
__global__ void __launch_bounds__(256) k_code_starts(const K *__restrict__ key, int n, int shift, int ncodes,
                                                     int *__restrict__ start)
{
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > ncodes) return;
    int lo = 0, hi = n;                       // first index with (key >> shift) >= e
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((long)(key[mid] >> shift) < (long)e) lo = mid + 1; else hi = mid;
    }
    start[e] = lo;
}

// Morton code of each ghost's bin (ghost rule of gpu_assign_bin_id, neighbor_meso.cu:413-417)
__global__ void __launch_bounds__(256) k_ghost_morton(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, BinGeom g, int nlocal, int nghost,
                                                      u32 *__restrict__ key, int *__restrict__ val)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nghost) return;
    const double c[3] = {x[nlocal + k], y[nlocal + k], z[nlocal + k]};
    int b[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        b[d] = clampi((int)((c[d] - g.lo[d]) * g.bininv[d] + 1.0), 0, g.mbin[d]);
        b[d] = (c[d] >= g.lo[d]) ? (c[d] <= g.hi[d] ? b[d] : g.mbin[d] - 1) : 0;
    }
    key[k] = interleave3((u32)b[0], (u32)b[1], (u32)b[2]);
    val[k] = k;
}

static inline int brick_grid(const BrickArgs &g) { return (g.nactive + 7) / 8 * 8; }

// flag[b] = 1 if (section, brick) b owns atoms; the engine scans the flags and compacts the ids
__global__ void __launch_bounds__(256) k_brick_flags(const int *__restrict__ estart, int M, int nb2, int *__restrict__ flag)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb2) return;
    int nbricks = M / 32, sec = b / nbricks, B = b % nbricks;
    size_t e0 = (size_t)sec * M + 32 * (size_t)B;
    flag[b] = estart[e0 + 32] > estart[e0] ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_brick_compact(const int *__restrict__ flag, const int *__restrict__ pos, int nb2,
                                                       int *__restrict__ active, int *__restrict__ nactive)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb2) return;
    if (flag[b]) active[pos[b]] = b;
    if (b == nb2 - 1) *nactive = pos[b] + flag[b];
}
void launch_brick_flags(const int *estart, int M, int *flag, hipStream_t s)
{
    int nb2 = 2 * (M / 32);
    hipLaunchKernelGGL(k_brick_flags, dim3((nb2 + 255) / 256), dim3(256), 0, s, estart, M, nb2, flag);
}
void launch_brick_compact(const int *flag, const int *pos, int M, int *active, int *nactive, hipStream_t s)
{
    int nb2 = 2 * (M / 32);
    hipLaunchKernelGGL(k_brick_compact, dim3((nb2 + 255) / 256), dim3(256), 0, s, flag, pos, nb2, active, nactive);
}

void launch_estart(const uint32_t *sorted_key, int n, int key_shift, int ncodes, int *estart, hipStream_t s)
{
    hipLaunchKernelGGL(k_code_starts<u32>, dim3((ncodes + 1 + 255) / 256), dim3(256), 0, s, sorted_key, n, key_shift, ncodes,
                       estart);
}

void launch_code_starts_u32(const uint32_t *sorted_key, int n, int ncodes, int *start, hipStream_t s)
{
    hipLaunchKernelGGL(k_code_starts<u32>, dim3((ncodes + 1 + 255) / 256), dim3(256), 0, s, sorted_key, n, 0, ncodes, start);
}

void launch_ghost_morton(const AtomSoA &a, const BinGeom &g, int nlocal, int nghost, uint32_t *key, int *val,
                         hipStream_t s)
{
    if (nghost > 0)
        hipLaunchKernelGGL(k_ghost_morton, dim3((nghost + 255) / 256), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], g, nlocal,
                           nghost, key, val);
}

void launch_brick_build(const BrickArgs &g, const float4 *coord4, float rc2, int n_col, int *count,
                        unsigned short *table16, int *overflow, hipStream_t s)
{
    if (g.nactive <= 0) return;
    hipLaunchKernelGGL(k_brick_build, dim3(brick_grid(g)), dim3(BRK_THREADS), 0, s, g, coord4, rc2, n_col, count, table16,
                       overflow);
}

void launch_brick_convert(const BrickArgs &g, int n_col, const int *count, const unsigned short *table16, int *table32,
                          int *overflow, hipStream_t s)
{
    if (g.nactive <= 0) return;
    hipLaunchKernelGGL(k_brick_convert, dim3(brick_grid(g)), dim3(BRK_THREADS), 0, s, g, n_col, count, table16, table32,
                       overflow);
}

void launch_brick_pair(const BrickArgs &g, const PairArgs &p, const unsigned short *table16, int fast, int *overflow,
                       hipStream_t s)
{
    if (p.end <= p.beg || g.nactive <= 0) return;
    size_t sm = (size_t)p.ntypes * p.ntypes * N_COEFF * (fast ? 4 : 8);
    if (fast) hipLaunchKernelGGL((k_brick_pair<true>), dim3(brick_grid(g)), dim3(BRK_THREADS), sm, s, g, p, table16, overflow);
    else hipLaunchKernelGGL((k_brick_pair<false>), dim3(brick_grid(g)), dim3(BRK_THREADS), sm, s, g, p, table16, overflow);
}

} // namespace meso
