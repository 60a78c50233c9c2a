// Brick kernels: the neighbour-table builder of the DPD hot path and its per-rebuild plan.
//
// Local atoms are sorted by [border bit][Morton(bin)][Morton(sub-cell)] (atom_meso.cu:268-308), so every aligned
// group of 64 Morton codes is a 4x4x4 brick of bins whose atoms are contiguous in memory (one run per section:
// bulk, border), and ghosts are sorted by Morton(bin) behind them.  One workgroup owns one brick (or shares it with a few
// others in small boxes).  The PLAN - for every brick the map "halo slot -> global atom index" of its 6x6x6-bin
// neighbourhood (~1900 atoms at rho=4, whole bin runs in halo-bin order) - is computed in the builder's prologue (or by
// k_brick_plan, option tile_plan); the builder copies the neighbourhood's coordinates from HBM into LDS ONCE per launch
// with coalesced reads and every candidate read then hits LDS.
//
// Replaces gpu_build_neighbor_list + gpu_join/transpose (neigh_build_meso.cu:20-240).  (Round 1 also kept a force kernel
// and a 16-bit-row builder on this layout - `layout=1`; retired in round 2, the cell-ordered ring kernel is faster.)
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "meso_device.h"

namespace meso {

#define BRK_CODES 64
#define BRK_H 6
#define BRK_NHB (BRK_H * BRK_H * BRK_H)
#define BRK_THREADS 640
#define BRK_WAVES (BRK_THREADS / 64)
#define BRK_MAXH 2176          // halo atoms staged per brick (mean ~1900 at rho=4; Poisson sigma ~44)
#define BRK_MAXOWN 704         // atoms owned per brick (mean ~563, sigma ~24)
#define BRK_HOFF_PITCH 448      // hoff[0..216] + hloc[224 + hb]: locals (bulk + border runs) of each halo bin
#define BRK_HLOC 224
#define BRK_HDR_PITCH 8
#define BRK_RING 256
#define BRK_OWNER_SHIFT 16

__device__ inline size_t row_word(int i, int c, int n_col) { return ((size_t)(i >> 6) * (n_col >> 3) + c) * 64 + (i & 63); }

__device__ inline u32 compact3(u32 x)
{
    x &= 0x09249249;
    x = (x ^ (x >> 2)) & 0x030c30c3;
    x = (x ^ (x >> 4)) & 0x0300f00f;
    x = (x ^ (x >> 8)) & 0xff0000ff;
    x = (x ^ (x >> 16)) & 0x000003ff;
    return x;
}

// XCD-aware order: workgroups b and b+8 share an L2, so each XCD walks a contiguous run of the ACTIVE list (the
// compacted, Morton-ordered ids of the bricks that own atoms).  The count may live on the device (no host round trip
// between the compaction and the launches): the grid then covers every brick and the surplus workgroups exit.
__device__ inline int brick_slot(const BrickArgs &g)
{
    // identity list (list builder of the cell-ordered layout): consecutive workgroup ids are spread over the XCDs by the
    // hardware, so consecutive Morton bricks - full and empty ones alike - are shared out evenly with no compaction pass
    if (!g.active) return (int)blockIdx.x < g.nactive ? (int)blockIdx.x : -1;
    const int na = g.nactive_dev ? *g.nactive_dev : g.nactive;
    const int per = (na + 7) >> 3;
    const int r = (int)(blockIdx.x >> 3);
    if (r >= per) return -1;
    const int slot = (int)(blockIdx.x & 7) * per + r;
    return slot < na ? slot : -1;
}

// =========================================================================================
// plan: halo map, halo-bin offsets, own-atom slots (once per rebuild)
// =========================================================================================
__global__ void __launch_bounds__(256) k_brick_plan(BrickArgs g, int *__restrict__ overflow)
{
    __shared__ int hoff[BRK_NHB + 1];
    __shared__ int hs0[BRK_NHB], hl0[BRK_NHB], hs1[BRK_NHB], hl1[BRK_NHB], hs2[BRK_NHB];
    __shared__ int ostart[2][BRK_CODES + 1];
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int slot = brick_slot(g);
    if (slot < 0) return;
    const int B = g.active ? g.active[slot] : slot;
    if (tid <= BRK_CODES) {
        ostart[0][tid] = g.estart[(size_t)BRK_CODES * B + tid];
        ostart[1][tid] = g.estart[(size_t)g.M + (size_t)BRK_CODES * B + tid];
    }
    const u32 code0 = (u32)BRK_CODES * (u32)B;
    const int bx0 = (int)compact3(code0), by0 = (int)compact3(code0 >> 1), bz0 = (int)compact3(code0 >> 2);
    int tot = 0;
    if (tid < BRK_NHB) {
        const int hx = bx0 - 1 + tid % BRK_H, hy = by0 - 1 + (tid / BRK_H) % BRK_H, hz = bz0 - 1 + tid / (BRK_H * BRK_H);
        int s0 = 0, l0 = 0, s1 = 0, l1 = 0, s2 = 0, l2 = 0;
        if (hx >= 0 && hx < g.mbin[0] && hy >= 0 && hy < g.mbin[1] && hz >= 0 && hz < g.mbin[2]) {
            const u32 m = interleave3((u32)hx, (u32)hy, (u32)hz);
            s0 = g.estart[m]; l0 = g.estart[m + 1] - s0;
            s1 = g.estart[(size_t)g.M + m]; l1 = g.estart[(size_t)g.M + m + 1] - s1;
            s2 = g.ghost_base + g.gstart[m]; l2 = g.gstart[m + 1] - g.gstart[m];
        }
        hs0[tid] = s0; hl0[tid] = l0; hs1[tid] = s1; hl1[tid] = l1; hs2[tid] = s2;
        tot = l0 + l1 + l2;
    }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; k++) base += wtot[k];
    if (tid < BRK_NHB) hoff[tid] = base + incl - tot;
    if (tid == BRK_NHB - 1) hoff[BRK_NHB] = base + incl;
    __syncthreads();
    const int nh = hoff[BRK_NHB];
    const int o0 = ostart[0][0], n0 = ostart[0][BRK_CODES] - o0, o1 = ostart[1][0], n1 = ostart[1][BRK_CODES] - o1;
    int *hdr = g.hdr + (size_t)slot * BRK_HDR_PITCH;
    if (n0 + n1 == 0) {                       // a brick that owns nothing (identity active list)
        if (tid == 0) { hdr[0] = 0; hdr[1] = 0; hdr[2] = 0; hdr[3] = 0; hdr[4] = 0; }
        return;
    }
    if (nh > g.maxh || (g.maxown > 0 && n0 + n1 > g.maxown)) {
        if (tid == 0) {
            atomicMax(overflow, 100000 + (nh > g.maxh ? nh : n0 + n1));
            hdr[0] = 0; hdr[1] = 0; hdr[2] = 0; hdr[3] = 0; hdr[4] = 0;
        }
        return;
    }
    if (tid == 0) { hdr[0] = nh; hdr[1] = o0; hdr[2] = n0; hdr[3] = o1; hdr[4] = n1; }
    if (tid == 0 && nh * 8 > g.maxh * 7) atomicMax(overflow + 5, nh);      // high-water mark: the engine grows the capacity ahead of an overflow
    for (int t = tid; t <= BRK_NHB; t += 256) g.hoff[(size_t)slot * BRK_HOFF_PITCH + t] = hoff[t];
    if (tid < BRK_NHB) g.hoff[(size_t)slot * BRK_HOFF_PITCH + BRK_HLOC + tid] = hl0[tid] + hl1[tid];
    // halo slot -> global (cell-order) atom index
    for (int h = tid; h < nh; h += 256) {
        int lo = 0, hi = BRK_NHB;            // largest hb with hoff[hb] <= h
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (hoff[mid] <= h) lo = mid; else hi = mid;
        }
        int off = h - hoff[lo], src;
        if (off < hl0[lo]) src = hs0[lo] + off;
        else if ((off -= hl0[lo]) < hl1[lo]) src = hs1[lo] + off;
        else src = hs2[lo] + (off - hl1[lo]);
        g.hmap[(size_t)slot * g.maxh + h] = (u32)src;
    }
    // own atom -> (halo slot, halo bin)
    for (int s = 0; s < 2; s++) {
        const int ns = s ? n1 : n0;
        for (int o = tid; o < ns; o += 256) {
            const int i = ostart[s][0] + o;
            int lo = 0, hi = BRK_CODES;      // largest k with ostart[s][k] <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (ostart[s][mid] <= i) lo = mid; else hi = mid;
            }
            const int k = lo;
            const int kx = (k & 1) | (((k >> 3) & 1) << 1), ky = ((k >> 1) & 1) | (((k >> 4) & 1) << 1),
                      kz = ((k >> 2) & 1) | (((k >> 5) & 1) << 1);
            const int hb = (kx + 1) + BRK_H * ((ky + 1) + BRK_H * (kz + 1));
            const int hslot = hoff[hb] + (s ? hl0[hb] : 0) + (i - ostart[s][k]);
            g.own_info[i] = (u32)hslot | ((u32)hb << 16);
        }
    }
}

// =========================================================================================
// neighbour table builder, wave-per-bin ballot stenciling (cell-ordered layout, global-index rows)
// =========================================================================================
// The reference's formulation (gpu_build_neighbor_list, neigh_build_meso.cu:20-119: a warp walks the stencil of one
// bin, ballot + popc compaction, no global atomics) on wave64 with the neighbourhood in LDS:
//   lanes = candidates: the 27-bin stencil of an own bin is 9 contiguous runs of halo slots (x-adjacent halo bins
//     are consecutive), ~237 atoms = 4 batches of 64, each candidate read from LDS once per group of own atoms;
//   own atoms of the bin are taken 8 at a time: their coordinates are held in SGPRs, every (own atom, batch) step is 6 VALU for the distance, one compare, ballot, mbcnt and a 2-byte LDS
//     store of the candidate's slot into the atom's staged row;
//   rows leave LDS as whole 32-byte chunks of the chunked-8 table (8 lanes per atom), slots translated to global
//     indices on the way out, tail slots padded with the atom itself.
// Entry order inside a row is (batch, lane): deterministic, and different from the lane-per-atom builders.
#define TB_G 4
#ifndef TB2_THREADS
#define TB2_THREADS 256      // workgroup of the 2x2x2-brick builder: 4 waves, 2 bins each
#endif
#ifndef TB_EXPANDED
#define TB_EXPANDED 0        // 1: distances in the expanded form on brick-relative coordinates (see the scan); 0: (o - c)^2 on absolute ones
#endif
#define TB_ROWPAD 64               // entries behind the last staged row (see the scan step)
#define TB_ROWCAP_MAX 1024
#define TBQ_N 32                   // persistent workgroups: brick counters (BrickArgs::queue)
#define TBQ_PITCH 32               // ... ints between two of them (a 128-byte line each); [0]: workgroups done         // rows are staged in LDS at their full capacity n_col (2 bytes per entry)

// E: brick edge in bins.  4: the 4x4x4 brick (64 Morton codes, 6x6x6-bin neighbourhood, 10 waves) of rounds 1-2.  2 (default):
// a 2x2x2 brick (8 codes, 4x4x4-bin neighbourhood, 4 waves of 2 bins each): eight times as many workgroups of a fifth of
// the LDS, eight to a CU - a 32^3 box (343 4-bricks = 2.7 workgroups per CU, one round) fills the chip evenly, and large boxes
// gain as well although 8 atoms are staged per own atom instead of 3.4 (staging is a small part; the finer grain hides the
// staging latency of one workgroup behind the scans of the seven others): 32^3 77 -> 52 us, 64^3 303 -> 265, 128^3 2257 -> 1784.
// The 4-brick remains the fallback when a 2-brick neighbourhood nears its LDS stage (which does not grow).
// PART: rows in two sections (RowPartArgs, kernels.h); false: plain rows - every entry keeps its place, which spares the row-out the
// classification and the per-entry address arithmetic (64^3: 177 against 201 us per build; what decks that rebuild on every step use)
// (the body of the kernel for ONE brick: vblock = the workgroup id a launch of one workgroup per brick would have had)
// (G, P: BrickArgs and RowPartArgs where the caller has them - the by-value kernel argument, or the kernel-argument segment itself
// read through a constant-address-space reference, see the persistent loop)
struct TileKArgs {
    BrickArgs g;
    const float4 *coord4;
    float rc2;
    int n_col;
    int *count, *table, *overflow;
    int split, dbg;
    RowPartArgs pt;
};
template <int E, bool PART, typename G, typename P>
__device__ __forceinline__ void tile_build_brick(G &g, const float4 *__restrict__ coord4, float rc2,
                                                 int n_col, int *__restrict__ count, int *__restrict__ table,
                                                 int *__restrict__ overflow, int split, int dbg, P &pt, int vblock)
{
#pragma clang fp contract(fast)
    constexpr int CODES = E * E * E, H = E + 2, NHB = H * H * H, THREADS = E == 4 ? BRK_THREADS : TB2_THREADS, WAVES = THREADS / 64;
    const int maxh = E == 4 ? g.maxh : g.maxh2;
    // staged neighbourhood, SoA (candidate reads are consecutive slots); sized at launch for g.maxh halo atoms, so denser
    // systems trade occupancy for capacity instead of failing
    extern __shared__ float tb_dyn[];
    float *hx = tb_dyn, *hy = hx + maxh, *hz = hy + maxh;
    u32 *hgi = (u32 *)(hz + maxh);
    unsigned short *rowbuf = (unsigned short *)(hgi + maxh);          // [wave][TB_G][n_col]
    __shared__ int hoff[NHB + 1];
    __shared__ int hloc[NHB];
    // (opaque to the optimiser: under the persistent loop everything derived from the thread id would otherwise be hoisted out of the
    // loop and kept in registers across the whole brick - 95 VGPRs, five waves per SIMD, instead of 70 and seven)
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_, lane = tid & 63, w = tid >> 6;
    // few bricks (small boxes, sub-boxes of many ranks): `split` workgroups share one brick, each staging the neighbourhood
    // and taking every split-th group of own bins - a brick then finishes sooner, which is what the launch waits for
    const int part = vblock % split;
    const int slot = vblock / split;
    if (slot >= g.nactive) return;
    int nh;
    if (g.plan_inline) {
        // the brick's plan (k_brick_plan) computed here, in LDS: the halo-bin runs from estart / gstart, their prefix, and
        // the halo slot -> global index map by binary search - no plan launch, no round trip of the map through HBM
        __shared__ int wtot[4];
        int *hs0 = (int *)rowbuf, *hl0 = hs0 + NHB, *hs1 = hl0 + NHB, *hl1 = hs1 + NHB, *hs2 = hl1 + NHB;   // rows are not staged yet
        // 2-bricks: the bricks that own real cells, heaviest first (order2, built by the engine from the geometry): the light
        // bricks at the faces, edges and corners of the bin grid then form the launch's tail instead of whole bricks
        const int B = (E == 2 && g.order2) ? g.order2[slot] : slot;
        const size_t e0 = (size_t)CODES * B;
        // (a brick without own atoms leaves; the 2-brick path asks after its bins' loads are on their way - one round trip less)
        if (E != 2 && g.estart[e0 + CODES] - g.estart[e0] + g.estart[(size_t)g.M + e0 + CODES] - g.estart[(size_t)g.M + e0] == 0) return;
        const u32 code0 = (u32)e0;
        const int bx0 = (int)compact3(code0), by0 = (int)compact3(code0 >> 1), bz0 = (int)compact3(code0 >> 2);
        // coordinates are staged relative to the corner of the brick's neighbourhood (see the scan)
        const float rx = TB_EXPANDED ? g.org[0] + (float)(bx0 - 1) * g.binw[0] : 0.f, ry = TB_EXPANDED ? g.org[1] + (float)(by0 - 1) * g.binw[1] : 0.f,
                    rz = TB_EXPANDED ? g.org[2] + (float)(bz0 - 1) * g.binw[2] : 0.f;
        if constexpr (E == 2) {
            // 2-brick: 64 halo bins, 4 lanes each (256 threads).  Every lane reads its bin's three runs itself (same-address loads
            // of four lanes), the prefix over the bins is a wave scan of the first lane's total, and the lanes of a bin then
            // stage its atoms four at a time: no search for the bin of a slot (six dependent LDS reads per staged atom before,
            // a fifth of the kernel's vector instructions), no run table in LDS.
            static_assert(E != 2 || TB2_THREADS == 256, "4 lanes per halo bin");
            const int hbn = tid >> 2, sub = tid & 3;
            const int qx = bx0 - 1 + (hbn & 3), qy = by0 - 1 + ((hbn >> 2) & 3), qz = bz0 - 1 + (hbn >> 4);
            int s0 = 0, l0 = 0, s1 = 0, l1 = 0, s2 = 0, l2 = 0;
            if (qx >= 0 && qx < g.mbin[0] && qy >= 0 && qy < g.mbin[1] && qz >= 0 && qz < g.mbin[2]) {
                const u32 m = interleave3((u32)qx, (u32)qy, (u32)qz);
                s0 = g.estart[m]; l0 = g.estart[m + 1] - s0;
                s1 = g.estart[(size_t)g.M + m]; l1 = g.estart[(size_t)g.M + m + 1] - s1;
                if (g.gcnt) {
                    const bool ghostcell = qx == 0 || qx == g.mbin[0] - 1 || qy == 0 || qy == g.mbin[1] - 1 || qz == 0 || qz == g.mbin[2] - 1;
                    if (ghostcell) { s2 = g.ghost_base + g.gstart[m]; l2 = g.gcnt[m]; }
                } else { s2 = g.ghost_base + g.gstart[m]; l2 = g.gstart[m + 1] - g.gstart[m]; }
            }
            const int nown = g.estart[e0 + CODES] - g.estart[e0] + g.estart[(size_t)g.M + e0 + CODES] - g.estart[(size_t)g.M + e0];
            const int tot = l0 + l1 + l2;
            if (nown == 0) return;
            int incl = sub == 0 ? tot : 0;
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) {            // (the other three lanes of a bin carry zeros: steps 1 and 2 add nothing)
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            // lanes 1..3 of a bin: the inclusive sum of their first lane
            incl = __shfl(incl, lane & ~3, 64);
            if (lane == 63) wtot[w] = incl;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) base += k < w ? wtot[k] : 0;
            nh = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            const int first = base + incl - tot;
            if (sub == 0) { hoff[hbn] = first; hloc[hbn] = l0 + l1; }
            if (tid == 0) hoff[NHB] = nh;
            if (nh > maxh) {
                if (tid == 0) atomicMax(overflow, 100000 + (int)(((long)nh * g.maxh + g.maxh2 - 1) / g.maxh2));
                return;
            }
            if (tid == 0 && nh * 4 > maxh * 3) overflow[6] = 1;
            if (tid == 0 && (long)nh * 27 > (long)g.maxh * 7) atomicMax(overflow + 5, (int)((long)nh * 27 / 8));
            for (int off = sub; off < tot; off += 4) {
                const int o1 = off - l0, o2 = o1 - l1;
                const u32 src = (u32)(o1 < 0 ? s0 + off : o2 < 0 ? s1 + o1 : s2 + o2);
                const int h = first + off;
                hgi[h] = src;
                const float4 c = coord4[src];
                hx[h] = c.x - rx; hy[h] = c.y - ry; hz[h] = c.z - rz;
            }
        } else {
        int tot = 0;
        if (tid < NHB) {
            const int qx = bx0 - 1 + tid % H, qy = by0 - 1 + (tid / H) % H, qz = bz0 - 1 + tid / (H * H);
            int s0 = 0, l0 = 0, s1 = 0, l1 = 0, s2 = 0, l2 = 0;
            if (qx >= 0 && qx < g.mbin[0] && qy >= 0 && qy < g.mbin[1] && qz >= 0 && qz < g.mbin[2]) {
                const u32 m = interleave3((u32)qx, (u32)qy, (u32)qz);
                s0 = g.estart[m]; l0 = g.estart[m + 1] - s0;
                s1 = g.estart[(size_t)g.M + m]; l1 = g.estart[(size_t)g.M + m + 1] - s1;
                if (g.gcnt) {
                    // (fused rebuild: only the tiles of ghost cells were written - start and count of a ghost cell, nothing for others)
                    const bool ghostcell = qx == 0 || qx == g.mbin[0] - 1 || qy == 0 || qy == g.mbin[1] - 1 || qz == 0 || qz == g.mbin[2] - 1;
                    if (ghostcell) { s2 = g.ghost_base + g.gstart[m]; l2 = g.gcnt[m]; }
                } else { s2 = g.ghost_base + g.gstart[m]; l2 = g.gstart[m + 1] - g.gstart[m]; }
            }
            hs0[tid] = s0; hl0[tid] = l0; hs1[tid] = s1; hl1[tid] = l1; hs2[tid] = s2;
            hloc[tid] = l0 + l1;
            tot = l0 + l1 + l2;
        }
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63 && w < 4) wtot[w] = incl;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < w && k < 4; k++) base += wtot[k];
        if (tid < NHB) hoff[tid] = base + incl - tot;
        if (tid == NHB - 1) hoff[NHB] = base + incl;
        __syncthreads();
        nh = hoff[NHB];
        if (nh > maxh) {
            // (2-brick: reported as the 4-brick neighbourhood of the same density would be - the engine grows both stages)
            if (tid == 0) atomicMax(overflow, 100000 + (E == 4 ? nh : (int)(((long)nh * g.maxh + g.maxh2 - 1) / g.maxh2)));
            return;
        }
        if (E == 4 && tid == 0 && nh * 8 > maxh * 7) atomicMax(overflow + 5, nh);      // high-water mark (see k_brick_plan)
        // the 2-brick's stage does not grow: at three quarters of it the engine goes back to the 4-brick, whose stage does (a
        // 4^3-bin neighbourhood feels a local compression more than a 6^3-bin one, so this comes first)
        if (E == 2 && tid == 0 && nh * 4 > maxh * 3) overflow[6] = 1;
        // ... and the 4-brick's stage keeps growing meanwhile: a 6^3-bin neighbourhood around this one holds at most 27/8 as many
        if (E == 2 && tid == 0 && (long)nh * 27 > (long)g.maxh * 7) atomicMax(overflow + 5, (int)((long)nh * 27 / 8));
        for (int h = tid; h < nh; h += THREADS) {
            int lo = 0, hi = NHB;            // largest halo bin with hoff[bin] <= h
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (hoff[mid] <= h) lo = mid; else hi = mid;
            }
            int off = h - hoff[lo];
            u32 src;
            if (off < hl0[lo]) src = (u32)(hs0[lo] + off);
            else if ((off -= hl0[lo]) < hl1[lo]) src = (u32)(hs1[lo] + off);
            else src = (u32)(hs2[lo] + (off - hl1[lo]));
            hgi[h] = src;
            const float4 c = coord4[src];
            hx[h] = c.x - rx; hy[h] = c.y - ry; hz[h] = c.z - rz;
        }
        }
    } else {
        const int *hdr = g.hdr + (size_t)slot * BRK_HDR_PITCH;
        nh = hdr[0];
        if (hdr[2] + hdr[4] == 0) return;
        for (int t = tid; t <= NHB; t += THREADS) hoff[t] = g.hoff[(size_t)slot * BRK_HOFF_PITCH + t];
        for (int t = tid; t < NHB; t += THREADS) hloc[t] = g.hoff[(size_t)slot * BRK_HOFF_PITCH + BRK_HLOC + t];
        float4 ref = coord4[g.hmap[(size_t)slot * g.maxh]];          // (any point of the brick serves as the origin)
        if (!TB_EXPANDED) ref = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int h = tid; h < nh; h += THREADS) {
            const u32 src = g.hmap[(size_t)slot * g.maxh + h];
            hgi[h] = src;
            const float4 c = coord4[src];
            hx[h] = c.x - ref.x; hy[h] = c.y - ref.y; hz[h] = c.z - ref.z;
        }
    }
    __shared__ int nextcell;         // (the bin queue of the 2-brick, see the main loop)
    constexpr bool QUEUE = E == 2;
    if (QUEUE && tid == 0) nextcell = WAVES;
    __syncthreads();
    unsigned short *myrow0 = rowbuf + (size_t)__builtin_amdgcn_readfirstlane(w) * TB_G * n_col;      // (scalar: row addresses are SALU work)
    auto myrow = [&](int t) { return myrow0 + t * n_col; };
    if (dbg == 1) return;        // timing ablation: staging only

    // Distances in the expanded form |c|^2 - 2 o.c <= rc^2 - |o|^2 on coordinates RELATIVE to the brick's corner (|c| < 8 bin
    // widths, so the cancellation costs < 1e-4 absolute; rc2 arrives widened by more than that - a list may hold a pair too many
    // at its outer edge, never one too few, and the force kernel's own test decides): 3 fma + 1 compare per (own atom, batch)
    // with -2 o and rc^2 - |o|^2 in SGPRs, instead of 3 sub + 3 mul/fma + compare.
    // 2-brick: the waves take the bins from a counter (a wave's first bin is its own number): the atoms per bin vary (9.5 +- 3), the
    // workgroup's LDS and wave slots are free again only when its slowest wave is done, and a wave with two light bins takes a
    // third one (32^3: 44 -> 39.5 us per build, 64^3: 180 -> 177).  4-brick: bins dealt out by number, as before.  (The counter is
    // set before the barrier that ends the staging.)
    // (2-brick with split > 1 - boxes whose bricks fill little more than one round of workgroups: `split` workgroups stage the same
    // neighbourhood and share its eight bins, draw d of part p = bin d * split + p: the launch has `split` times the workgroups of
    // a fraction of the work, so the staging of one overlaps the scans of the others instead of all staging at once)
    for (int k = QUEUE ? __builtin_amdgcn_readfirstlane(w) * split + part : w + WAVES * part; k < CODES; k += QUEUE ? 0 : WAVES * split) {
        const int kx = (k & 1) | (((k >> 3) & 1) << 1), ky = ((k >> 1) & 1) | (((k >> 4) & 1) << 1),
                  kz = ((k >> 2) & 1) | (((k >> 5) & 1) << 1);
        const int hb = (kx + 1) + H * ((ky + 1) + H * (kz + 1));
        // wave-uniform values are forced into SGPRs: counters, branches and the own coordinates then stay scalar
        const int own0 = __builtin_amdgcn_readfirstlane(hoff[hb]), na = __builtin_amdgcn_readfirstlane(hloc[hb]);
        // (the next bin is drawn when this iteration ends, whichever way it ends)
        struct Next {
            int &k; int lane; int *q; int kcur; int split, part;
            __device__ ~Next() {
                if (!QUEUE) return;
                int v = 0;
                if (lane == 0) v = __hip_atomic_fetch_add(q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                k = __builtin_amdgcn_readfirstlane(v) * split + part;
            }
        } next_{k, lane, &nextcell, k, split, part};
        if (na == 0) continue;
        // the 9 candidate runs of the bin's stencil and their prefix: lanes 0..8 read the two offsets of "their" run in one LDS
        // round trip, v_readlane moves the 18 values to SGPRs.  The table lives only while candidates are loaded - 19 SGPRs
        // that the scan must not pay for with spills - and is rebuilt for the rare bin with more than 4 batches
        auto load_cand = [&](int b0, int nb, int *cs, float *cx, float *cy, float *cz, float *c2) {
            int rstart[9], pre[10];
            pre[0] = 0;
            const int rl = lane < 9 ? lane : 0;
            const int hrow_l = hb + (rl % 3 - 1) * H + (rl / 3 - 1) * H * H;
            const int lo_l = hoff[hrow_l - 1], hi_l = hoff[hrow_l + 2];
#pragma unroll
            for (int r = 0; r < 9; r++) {
                rstart[r] = __builtin_amdgcn_readlane(lo_l, r);
                pre[r + 1] = pre[r] + (__builtin_amdgcn_readlane(hi_l, r) - rstart[r]);
            }
            const int ncand_ = pre[9];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (q < nb) {
                    const int id = ((b0 + q) << 6) + lane;
                    int base = rstart[0];
#pragma unroll
                    for (int r = 1; r < 9; r++) base = id >= pre[r] ? rstart[r] - pre[r] : base;
                    const bool valid = id < ncand_;
                    cs[q] = valid ? id + base : 0;
                    cx[q] = hx[cs[q]]; cy[q] = hy[cs[q]]; cz[q] = hz[cs[q]];
                    c2[q] = valid ? __builtin_fmaf(cx[q], cx[q], __builtin_fmaf(cy[q], cy[q], cz[q] * cz[q])) : 1.0e30f;   // never inside
                    if (!TB_EXPANDED && !valid) cx[q] = 1.0e18f;
                }
            }
            return ncand_;
        };
        // candidates of the first 4 batches (256 atoms: the usual stencil holds ~237) stay in registers for all groups
        // of this bin; later batches (denser systems) are reloaded per group
        int cs4[4];
        float cx4[4], cy4[4], cz4[4], cq4[4];
#pragma unroll
        for (int b = 0; b < 4; b++) { cs4[b] = 0; cx4[b] = cy4[b] = cz4[b] = 0.f; cq4[b] = 1.0e30f; }
        const int ncand = __builtin_amdgcn_readfirstlane(load_cand(0, 4, cs4, cx4, cy4, cz4, cq4));
        const int nbatch = (ncand + 63) >> 6;

        float ax_l = 0.f, ay_l = 0.f, az_l = 0.f, th_l = 0.f;
        for (int g0 = 0; g0 < na; g0 += TB_G) {
            const int ng = min(TB_G, na - g0);
            if ((g0 & 63) == 0) {
                // the next 64 own atoms of the bin, one per lane: -2 o and rc^2 - |o|^2, once; a group takes its four by v_readlane
                const int tl = own0 + g0 + (g0 + lane < na ? lane : 0);
                const float ox_l = hx[tl], oy_l = hy[tl], oz_l = hz[tl];
                if (TB_EXPANDED) {
                    ax_l = -2.f * ox_l; ay_l = -2.f * oy_l; az_l = -2.f * oz_l;
                    th_l = rc2 - __builtin_fmaf(ox_l, ox_l, __builtin_fmaf(oy_l, oy_l, oz_l * oz_l));
                } else { ax_l = ox_l; ay_l = oy_l; az_l = oz_l; th_l = rc2; }
            }
            float ax[TB_G], ay[TB_G], az[TB_G], th[TB_G];
#pragma unroll
            for (int t = 0; t < TB_G; t++) {
                const int sl = (g0 & 63) + (t < ng ? t : 0);
                ax[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ax_l), sl));
                ay[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ay_l), sl));
                az[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(az_l), sl));
                th[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(th_l), sl));
            }
            int nrow[TB_G];
#pragma unroll
            for (int t = 0; t < TB_G; t++) nrow[t] = 0;
            // one (own atom, 64-candidate batch) step: 3 fma, lane mask straight from the compares (LLVM predicates: 5 = OLE,
            // 33 = NE), slot of every hit in the atom's LDS row.  No "any hit?" branch: a batch almost always holds one, and
            // straight-line code lets the chains of the group's atoms overlap.  FULL: all TB_G own atoms present (16 independent
            // chains for 4 batches).  Special-bond partners are dropped by k_filter_exclusion afterwards.
            // (two own atoms per packed instruction: the differences, squares and sums of atoms t and t + 1 are the two halves of
            // v_pk_add / v_pk_mul / v_pk_fma_f32 - 6 instructions for two distances - with the candidate's coordinate broadcast)
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto scan = [&](auto full, const int cs, const float cx, const float cy, const float cz, const float c2) {
#pragma unroll
                for (int t0 = 0; t0 < TB_G; t0 += 2) {
                    if (!(decltype(full)::value || t0 < ng)) continue;
                    f2 d2;
                    if (TB_EXPANDED) {
                        d2[0] = __builtin_fmaf(ax[t0], cx, __builtin_fmaf(ay[t0], cy, __builtin_fmaf(az[t0], cz, c2)));
                        d2[1] = __builtin_fmaf(ax[t0 + 1], cx, __builtin_fmaf(ay[t0 + 1], cy, __builtin_fmaf(az[t0 + 1], cz, c2)));
                    } else {
                        const f2 dx = (f2){ax[t0], ax[t0 + 1]} - (f2){cx, cx}, dy = (f2){ay[t0], ay[t0 + 1]} - (f2){cy, cy},
                                 dz = (f2){az[t0], az[t0 + 1]} - (f2){cz, cz};
                        d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int t = t0 + u;
                        if (!(decltype(full)::value || t < ng)) continue;
                        const float d = d2[u];
                        const u64 m = __builtin_amdgcn_fcmpf(d, th[t], 5) & __builtin_amdgcn_uicmp((u32)cs, (u32)(own0 + g0 + t), 33);
                        const bool hit = (d <= th[t]) & (cs != own0 + g0 + t);
                        // (row position = entries so far [scalar, SALU work, kept inside the row] + hits in lower lanes [mbcnt x 2,
                        // shift-add].  The lane part is not clamped - one instruction of eleven per step, 4 % of the kernel: a row
                        // that overflows writes up to 63 entries into the next row, or into the TB_ROWPAD entries behind the last.
                        // Every entry written is a valid halo slot, the overflow is reported below and ends the run, as it did.)
                        const int have = min(nrow[t], n_col - 1);
                        const u32 cnt = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                        if (hit) (myrow(t) + have)[cnt] = (unsigned short)cs;
                        nrow[t] += __popcll(m);
                    }
                }
            };
            if (nbatch >= 4 && ng == TB_G && dbg != 3) {
                // the common case (4 full batches, 4 own atoms) without a single branch: 16 independent chains
#pragma unroll
                for (int b = 0; b < 4; b++) scan(std::true_type{}, cs4[b], cx4[b], cy4[b], cz4[b], cq4[b]);
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (b < nbatch && dbg != 3) scan(std::false_type{}, cs4[b], cx4[b], cy4[b], cz4[b], cq4[b]);
            }
            for (int b = 4; b < nbatch; b++) {
                int cs[4];
                float cx[4], cy[4], cz[4], cq[4];
                load_cand(b, 1, cs, cx, cy, cz, cq);
                scan(std::false_type{}, cs[0], cx[0], cy[0], cz[0], cq[0]);
            }
            if constexpr (!PART) {
                // plain rows out, 16 lanes per own atom of the group: 8 lanes fill one 32-byte chunk of the chunked-8 table, slots become
                // global indices, the tail of the last chunk is padded with the atom itself
                const int t_l = lane >> 4, el = lane & 15;
                const bool on = t_l < ng;
                const int n_l = t_l == 0 ? nrow[0] : t_l == 1 ? nrow[1] : t_l == 2 ? nrow[2] : nrow[3];
                const int i_l = (int)hgi[own0 + g0 + (on ? t_l : 0)];
                const int nn_l = (dbg == 2 || !on) ? 0 : min(n_l, n_col);
                const int pad_l = (nn_l + 7) & ~7;
                // (row_word8(i, 0, n_col) with unsigned factors: one 32 x 32 -> 64 multiply-add)
                int *dst = table + (((size_t)((u32)i_l >> 6) * (u32)(n_col >> 3)) * 64 + ((u32)i_l & 63u)) * 8 + (el & 7) + (size_t)(el >> 3) * 512;
                const unsigned short *row_l = myrow0 + t_l * n_col;
                const int nmax = __builtin_amdgcn_readfirstlane(max(max(nrow[0], nrow[1]), max(nrow[2], nrow[3])));
                const int padmax = (min(nmax, n_col) + 7) & ~7;
                for (int e0 = 0; e0 < padmax; e0 += 16) {
                    const int e = e0 + el;
                    if (e < pad_l) {
                        int val = i_l;
                        if (e < nn_l) val = (int)hgi[row_l[e]];
                        dst[(size_t)(e0 >> 3) * 512] = val;
                    }
                }
                if (on && el == 0) {
                    if (n_l > n_col) atomicMax(overflow, n_l);
                    count[i_l] = nn_l;
                }
            } else
            // rows out, 16 lanes per own atom of the group, four atoms at once: slots become global indices and every entry goes to
            // one of the row's two sections (RowPartArgs, kernels.h) - front: what this atom evaluates, back: mirrored entries
            // (partner in the same pairing group, both local, and the balanced rule gives the pair to the partner).  One ballot
            // per 16 entries places both: a front entry moves up by the mirrored entries before it, a mirrored one takes that
            // number as its place in the back row.  Both sections are padded with the atom itself to whole 32-byte chunks.
            {
                const int t_l = lane >> 4, el = lane & 15;
                const bool on = t_l < ng;
                const int n_l = t_l == 0 ? nrow[0] : t_l == 1 ? nrow[1] : t_l == 2 ? nrow[2] : nrow[3];
                const u32 i_l = hgi[own0 + g0 + (on ? t_l : 0)];
                const int nn_l = (dbg == 2 || !on) ? 0 : min(n_l, n_col);
                // (row_word8(i, 0, n_col) in bytes with unsigned factors: one 32 x 32 -> 64 multiply-add per table)
                char *dstF = (char *)table + ((size_t)(i_l >> 6) * (u32)(n_col * 256) + ((i_l & 63u) << 5));
                char *dstB = (char *)pt.back + ((size_t)(i_l >> 6) * (u32)(pt.nb_col * 256) + ((i_l & 63u) << 5));
                const unsigned short *row_l = myrow0 + t_l * n_col + el;
                const int nmax = min(n_col, __builtin_amdgcn_readfirstlane(max(max(nrow[0], nrow[1]), max(nrow[2], nrow[3]))));
                // pairing happens inside aligned groups that lie wholly below nlocal (the force kernel applies the same test to its
                // workgroup): an atom of the last, incomplete group has no mirrored entries
                const u32 grp_l = (i_l | (u32)(pt.group - 1)) < (u32)pt.nlocal ? (u32)pt.group : 0u;
                // my 16-lane group inside my half of the wave's lane mask: the lanes below me, and all of it
                const u32 below16 = ((1u << el) - 1u) << (16 * (t_l & 1)), grp16 = 0xFFFFu << (16 * (t_l & 1));
                const int hsh = lane & 32;
                int nbr = 0;                     // back entries so far (the same in the 16 lanes of an atom)
                int pfe = el;                    // my entry's place if nothing before it were mirrored
                const int rem = nn_l - el;       // my entry of iteration e0 exists while e0 < rem
                // byte offset of entry p inside a chunked-8 row: (p >> 3) * 2048 + (p & 7) * 4 = 4 p + 252 (p & ~7)
                auto rowoff = [](int p) -> u32 { return __umul24((u32)p & ~7u, 252u) + ((u32)p << 2); };
                for (int e0 = 0; e0 < nmax; e0 += 16) {
                    // (a lane past its row reads a stale slot - a valid one, every slot ever staged is - and neither counts nor stores)
                    const u32 j = hgi[row_l[e0]];
                    const u32 y = j - i_l, z = (y << 31) + y;                       // sign of z = (j < i) != ((j - i) & 1)
                    // lane masks straight from the compares (LLVM predicates: 36 = ULT, 38 = SGT, 40 = SLT): mirrored entries of the 64 lanes
                    const u64 vm = __builtin_amdgcn_sicmp(rem, e0, 38);
                    const u64 m = __builtin_amdgcn_uicmp(j ^ i_l, grp_l, 36) & __builtin_amdgcn_sicmp((int)z, 0, 40) & vm;
                    const u32 half = (u32)(m >> hsh);
                    const int cc = nbr + __popc(half & below16);
                    // (a back row that overflows - reported below, the run ends - keeps writing into its last entry)
                    const int pf = pfe - cc, pb = min(cc, pt.nb_col - 1);
                    int pos;
                    char *d;
                    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(pos) : "v"(pf), "v"(pb), "s"(m));
                    asm("v_cndmask_b32 %0, %2, %4, %6\n\tv_cndmask_b32 %1, %3, %5, %6" : "=&v"(((u32 *)&d)[0]), "=&v"(((u32 *)&d)[1])
                        : "v"((u32)(size_t)dstF), "v"((u32)((size_t)dstF >> 32)), "v"((u32)(size_t)dstB), "v"((u32)((size_t)dstB >> 32)), "s"(m));
                    if (rem > e0) *(__attribute__((address_space(1))) int *)(size_t)(d + rowoff(pos)) = (int)j;      // (a global store, not a flat one)
                    nbr += __popc(half & grp16);
                    pfe += 16;
                }
                // the tails of the two sections' last chunks: the atom itself (r = 0: no kernel takes it for a neighbour) - lanes 0..7 of
                // an atom's 16 write behind the front section, lanes 8..15 behind the back section
                const int nf = nn_l - nbr, nb = min(nbr, pt.nb_col);
                {
                    const bool bk = el >= 8;
                    const int ns = bk ? nb : nf, ps = ns + (el & 7);
                    char *dp = bk ? dstB : dstF;
                    if (on && ps < ((ns + 7) & ~7)) *(__attribute__((address_space(1))) int *)(size_t)(dp + rowoff(ps)) = (int)i_l;
                }
                if (on && el == 0) {
                    if (n_l > n_col || nbr > pt.nb_col) atomicMax(overflow, max(n_l, n_col + 1));
                    count[i_l] = nf;
                    if (pt.group) pt.nback[i_l] = nb;
                }
            }
        }
    }
}

// One workgroup per brick (queue null: the grid covers every brick), or PERSISTENT workgroups (round 6, BrickArgs::queue: the grid is
// what the card holds at once and every workgroup draws bricks from a counter until none is left - the reference's builder is a
// grid-stride loop over bins too, neigh_build_meso.cu:58).  The draw for the NEXT brick is posted before this one is staged, so its
// round trip is hidden; the last workgroup to leave puts the counter back to zero for the next launch (no memset between launches).
template <int E, bool PART, bool PERSIST = false>
__global__ void __launch_bounds__(E == 4 ? BRK_THREADS : TB2_THREADS, E == 4 ? 3 : (PERSIST ? 7 : 2)) k_tile_build(TileKArgs a)
{
    if constexpr (!PERSIST) {
        tile_build_brick<E, PART>(a.g, a.coord4, a.rc2, a.n_col, a.count, a.table, a.overflow, a.split, a.dbg, a.pt, (int)blockIdx.x);
        return;
    }
    const BrickArgs &g = a.g;
    const int split = a.split;
    // (the first brick is drawn too: should the runtime's occupancy answer exceed what the card really holds at once, the workgroups
    // that start late find the counters past the end and leave - with blockIdx as the first brick they would add a whole round)
    // TBQ_N counters, each in a 128-byte line of its own: 15 600 draws from ONE address took as long as the whole build (same-address
    // atomics of eight XCDs meet at the memory side, ~17 ns each); counter q hands out the bricks q, q + TBQ_N, q + 2 TBQ_N, ...
    __shared__ int next_vb;
    const int total = g.nactive * split;
    const int nq = min(TBQ_N, (int)gridDim.x);      // (every counter needs a workgroup that draws from it)
    const int q = (int)blockIdx.x % nq;
    int *ctr = g.queue + TBQ_PITCH * (1 + q);
    if (threadIdx.x == 0) next_vb = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * nq + q;
    __syncthreads();
    int vb = __builtin_amdgcn_readfirstlane(next_vb);      // (scalar, as blockIdx is: the brick's geometry stays SALU work)
    __syncthreads();
    while (vb < total) {
        int drawn = 0;
        if (threadIdx.x == 0) drawn = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        {
            // the arguments are read from the kernel-argument segment anew for every brick (the pointer is opaque to the optimiser):
            // kept in scalar registers across the loop they were spilled - 92 SGPRs into lanes of ten more VGPRs, a wave per SIMD less
            typedef const __attribute__((address_space(4))) TileKArgs *KP;
            KP ka = (KP)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
            tile_build_brick<E, PART>(ka->g, ka->coord4, ka->rc2, ka->n_col, ka->count, ka->table, ka->overflow, ka->split, ka->dbg, ka->pt, vb);
        }
        if (threadIdx.x == 0) next_vb = drawn * nq + q;
        __syncthreads();           // every wave is through with the brick's LDS; the next brick's number is there
        vb = __builtin_amdgcn_readfirstlane(next_vb);
        __syncthreads();           // ... and read by all before thread 0 writes the one after it
    }
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(g.queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        // the last workgroup out: every other one has made its last draw
        __hip_atomic_store(g.queue, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < TBQ_N; k++) __hip_atomic_store(g.queue + TBQ_PITCH * (1 + k), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// Ghost binning without a sort (the ghosts' comparison sort was ~15 launches of ~6 us): count per Morton code with the rank
// of each ghost inside its code as the atomic's return value, exclusive scan -> gstart, placement, and a per-code pass
// that puts the (few) ghosts of a code in ascending ghost index - the order a stable sort gives, so storage stays
// deterministic - and writes gslot (ghost -> slot).
__global__ void __launch_bounds__(256) k_ghost_count(const double *__restrict__ x, const double *__restrict__ y,
                                                     const double *__restrict__ z, BinGeom g, int nlocal, int nghost,
                                                     u32 *__restrict__ code, int *__restrict__ rank, int *__restrict__ cnt,
                                                     const int *__restrict__ nghost_dev)
{
    if (nghost_dev) nghost = *nghost_dev;      // nghost only sized the grid; the count is on the device
    // whole waves take part in every trip (run_rank is a wave-level operation)
    for (int base = blockIdx.x * blockDim.x; base < nghost; base += gridDim.x * blockDim.x) {
        const int k = base + (int)threadIdx.x;
        const bool valid = k < nghost;
        u32 m = 0;
        if (valid) {
            const double c[3] = {x[nlocal + k], y[nlocal + k], z[nlocal + k]};
            int b[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                b[d] = clampi((int)((c[d] - g.lo[d]) * g.bininv[d] + 1.0), 0, g.mbin[d]);
                b[d] = (c[d] >= g.lo[d]) ? (c[d] <= g.hi[d] ? b[d] : g.mbin[d] - 1) : 0;
            }
            m = interleave3((u32)b[0], (u32)b[1], (u32)b[2]);
            code[k] = m;
        }
        const int r = run_rank(m, valid, cnt);
        if (valid) rank[k] = r;
    }
}
void launch_ghost_count(const AtomSoA &a, const BinGeom &g, int nlocal, int nghost, uint32_t *code, int *rank, int *cnt,
                        const int *nghost_dev, hipStream_t s)
{
    if (nghost > 0)
        hipLaunchKernelGGL(k_ghost_count, dim3((nghost + 255) / 256), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], g, nlocal, nghost, code,
                           rank, cnt, nghost_dev);
}
static inline int brick_grid(const BrickArgs &g) { return (g.nactive + 7) / 8 * 8; }

int brick_codes() { return BRK_CODES; }
int brick_static_maxh() { return BRK_MAXH; }        // capacity of the brick-layout kernels (static LDS arrays)
int brick_static_maxown() { return BRK_MAXOWN; }
// largest halo the tile builder can stage: 160 KB of LDS minus its static part (row staging, bin offsets), 16 B per atom
int tile_build_maxh_limit(int n_col, int) { return (int)((160 * 1024 - (BRK_WAVES * TB_G * n_col * 2 + TB_ROWPAD * 2 + 2 * (BRK_NHB + 1) * 4 + 1024)) / 16); }
size_t brick_hoff_pitch() { return BRK_HOFF_PITCH; }
size_t brick_hdr_pitch() { return BRK_HDR_PITCH; }

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

// gpu_filter_exclusion (neigh_build_meso.cu:497-544): special-bond partners leave the rows by TAG (an atom and its periodic
// images carry the same tag).  A pass of its own over the finished rows, like the reference's: a wave looks at 64 consecutive
// atoms, and for each one that has special partners (most atoms of a solution have none) its row is read lane = entry, the
// entry's tag comes from the cell-ordered tag array, the atom's special list sits in the first lanes of its 16-lane group and
// is handed out with a lane shuffle, kept entries are compacted in place with ballot + popcount (writes never pass the reads of
// a later batch) and the tail of the last chunk is padded with the atom itself again.  Inside the list builder the same filter cost 50 us per build
// even on a deck without a single exclusion (scalar-register spills of the larger kernel) and a third of its occupancy
// while the tags of the neighbourhood were staged in LDS.
__global__ void __launch_bounds__(256) k_filter_exclusion(ExclArgs ex, int nlocal, int n_col, int *__restrict__ count,
                                                          int *__restrict__ table, int nb_col, int *__restrict__ nback, int *__restrict__ back)
{
    // (nback != null: partitioned rows, RowPartArgs in kernels.h - the front row and the back row are compacted one after the other)
    // FOUR atoms at a time per wave, 16 lanes each: the filter is a chain of dependent memory round trips per atom (special
    // list and count, row entries, their tags), so its speed is the number of atoms in flight (one atom per wave: 250 us on a
    // melt of 1 M chain beads, four: see profiles/r02_notes.md)
    __shared__ unsigned char todo_s[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + w) * 64;
    if (i0 >= nlocal) return;
    const int il = i0 + lane;
    const int nsp_own = il < nlocal ? ex.nspecial[il] : 0;
    const int cnt_own = il < nlocal ? count[il] : 0;       // (loaded for all 64 atoms at once: one round trip less per atom below)
    const int bck_own = (nback && il < nlocal) ? nback[il] : 0;
    const u64 todo = __builtin_amdgcn_ballot_w64(nsp_own > 0);
    const int ntodo = __popcll(todo);
    if (ntodo == 0) return;
    // the atoms that have special partners, in order: slot k of todo_s = lane offset of the k-th one
    if (nsp_own > 0) todo_s[w][__builtin_amdgcn_mbcnt_hi((u32)(todo >> 32), __builtin_amdgcn_mbcnt_lo((u32)todo, 0u))] = (unsigned char)lane;
    const int g = lane >> 4, l16 = lane & 15;
    const u32 below = (1u << l16) - 1u;
    for (int r = 0; r < ntodo; r += 4) {
        const bool on = r + g < ntodo;
        const int al = on ? (int)todo_s[w][r + g] : 0;
        const int i = i0 + al;
        // (shuffles outside any branch: a lane that is switched off hands out nothing)
        const int nsp_a = __shfl(nsp_own, al, 64), n_a = __shfl(cnt_own, al, 64), nb_a = __shfl(bck_own, al, 64);
        const int nsp = on ? nsp_a : 0;
        // (requested together with the row entries: the list load does not wait for anything; lists longer than 16: rest from memory)
        int sp_l = (on && l16 < ex.msp) ? ex.special[(size_t)i * ex.msp + l16] : -1;
        if (l16 >= nsp) sp_l = -1;
        int nspmax = nsp;
#pragma unroll
        for (int o = 32; o >= 16; o >>= 1) nspmax = max(nspmax, __shfl_xor(nspmax, o, 64));
        nspmax = __builtin_amdgcn_readfirstlane(nspmax);
        // the n entries of a row, compacted in place (writes never pass the reads of a later batch), tail padded with the atom itself
        auto section = [&](int *row, int n) -> int {
            int nmax = n;
#pragma unroll
            for (int o = 32; o >= 16; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
            nmax = __builtin_amdgcn_readfirstlane(nmax);
            int nout = 0;
            for (int e0 = 0; e0 < nmax; e0 += 16) {
                const int e = e0 + l16;
                const bool in = e < n;
                const int j = in ? row[(size_t)(e >> 3) * 512 + (e & 7)] : i;
                const int tg = ex.tagc[j];
                bool keep = in;
                for (int sp = 0; sp < min(nspmax, 16); sp++) keep = keep & (__shfl(sp_l, (lane & 48) + sp, 64) != tg);
                for (int sp = 16; sp < nsp; sp++) keep = keep & (ex.special[(size_t)i * ex.msp + sp] != tg);
                const u32 m16 = (u32)(__builtin_amdgcn_ballot_w64(keep) >> (16 * g)) & 0xffffu;
                const int pos = nout + __popc(m16 & below);
                if (keep) row[(size_t)(pos >> 3) * 512 + (pos & 7)] = j;
                nout += __popc(m16);
            }
            if (on) {
                const int e = nout + l16;
                if (e < ((nout + 7) & ~7)) row[(size_t)(e >> 3) * 512 + (e & 7)] = i;
            }
            return nout;
        };
        // Rows of ordinary length (front <= 64 entries, back <= 32 - wave-uniform test): every entry of both sections is requested
        // before anything is looked at, then every tag: two round trips for the atom instead of two per batch of 16 entries and section
        // (the filter is a chain of dependent round trips: that is its whole cost).  All reads precede all writes here.
        int *rowF = table + row_word8(i, 0, n_col) * 8, *rowB = nback ? back + row_word8(i, 0, nb_col) * 8 : rowF;
        const int nF = on ? n_a : 0, nB = (on && nback) ? nb_a : 0;
        int nFmax = nF, nBmax = nB;
#pragma unroll
        for (int o = 32; o >= 16; o >>= 1) { nFmax = max(nFmax, __shfl_xor(nFmax, o, 64)); nBmax = max(nBmax, __shfl_xor(nBmax, o, 64)); }
        nFmax = __builtin_amdgcn_readfirstlane(nFmax); nBmax = __builtin_amdgcn_readfirstlane(nBmax);
        if (nFmax <= 64 && nBmax <= 32 && nspmax <= 16) {
            int jf[4], jb[2], tf[4], tb[2];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int e = 16 * q + l16; jf[q] = e < nF ? rowF[(size_t)(e >> 3) * 512 + (e & 7)] : i; }
#pragma unroll
            for (int q = 0; q < 2; q++) { const int e = 16 * q + l16; jb[q] = e < nB ? rowB[(size_t)(e >> 3) * 512 + (e & 7)] : i; }
#pragma unroll
            for (int q = 0; q < 4; q++) tf[q] = ex.tagc[jf[q]];
#pragma unroll
            for (int q = 0; q < 2; q++) tb[q] = ex.tagc[jb[q]];
            auto compact = [&](int *row, int n, const int *jj, const int *tt, int iters, int nmax) -> int {
                int nout = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (q >= iters || 16 * q >= nmax) break;
                    bool keep = 16 * q + l16 < n;
                    for (int sp = 0; sp < nspmax; sp++) keep = keep & (__shfl(sp_l, (lane & 48) + sp, 64) != tt[q]);
                    const u32 m16 = (u32)(__builtin_amdgcn_ballot_w64(keep) >> (16 * g)) & 0xffffu;
                    const int pos = nout + __popc(m16 & below);
                    if (keep) row[(size_t)(pos >> 3) * 512 + (pos & 7)] = jj[q];
                    nout += __popc(m16);
                }
                if (on) {
                    const int e = nout + l16;
                    if (e < ((nout + 7) & ~7)) row[(size_t)(e >> 3) * 512 + (e & 7)] = i;
                }
                return nout;
            };
            const int nf = compact(rowF, nF, jf, tf, 4, nFmax);
            if (on && l16 == 0) count[i] = nf;
            if (nback) {
                const int nb = compact(rowB, nB, jb, tb, 2, nBmax);
                if (on && l16 == 0) nback[i] = nb;
            }
            continue;
        }
        const int nf = section(rowF, nF);
        if (on && l16 == 0) count[i] = nf;
        if (nback) {
            const int nb = section(rowB, nB);
            if (on && l16 == 0) nback[i] = nb;
        }
    }
}

void launch_brick_plan(const BrickArgs &g, int *overflow, hipStream_t s)
{
    if (g.nactive <= 0) return;
    hipLaunchKernelGGL(k_brick_plan, dim3(brick_grid(g)), dim3(256), 0, s, g, overflow);
}

void launch_tile_build(const BrickArgs &g, const float4 *coord4, float rc2, int n_col, int *count, int *table, int *overflow,
                       const ExclArgs *excl, int nlocal, int dbg, hipStream_t s, const RowPartArgs *part)
{
    if (g.nactive <= 0) return;
    RowPartArgs pt = {0, 0, nullptr, nullptr, 8};
    if (part) pt = *part;
    if (!pt.nback || !pt.back) { pt.group = 0; pt.back = table; pt.nback = count; }      // (plain rows: nothing is ever written behind)
    auto kargs = [&](const BrickArgs &gg, int split_, int dbg_) { return TileKArgs{gg, coord4, rc2, n_col, count, table, overflow, split_, dbg_, pt}; };
    // (the scan's expanded distance form loses < 1e-4 absolute to cancellation: the list cutoff is widened by more than that)
    if (TB_EXPANDED) rc2 += 4.0e-4f;
    // few bricks (small boxes, sub-boxes of many ranks): several workgroups share a brick as long as all of them still fit the
    // card in one round (3 workgroups per CU); only the bricks that overlap the bin grid own atoms (measured: 25^3 best with 4
    // workgroups per brick, 32^3 with 2, from 48^3 on with 1)
    const int occupied = ((g.mbin[0] + 3) / 4) * ((g.mbin[1] + 3) / 4) * ((g.mbin[2] + 3) / 4);
    if (dbg != 99 && g.plan_inline && g.maxh2 > 0 && occupied <= g.brick2_limit && g.M >= 64) {
        // 2x2x2 bricks: eight times as many workgroups of 4 waves
        BrickArgs g2 = g;
        g2.nactive = g.order2 ? g.norder2 : g.M / 8;
        dim3 tgrid2((g2.nactive + 7) / 8 * 8);
        const size_t dyn2 = (size_t)g.maxh2 * 16 + (size_t)(TB2_THREADS / 64) * TB_G * n_col * 2 + TB_ROWPAD * 2;
        if (g2.queue) {
            // persistent workgroups: as many as are resident at once (asked of the runtime once per stage size and row form)
            static int per_cu[2] = {0, 0}, ncu = 0;
            static size_t per_cu_dyn[2] = {0, 0};
            const int f = pt.group ? 1 : 0;
            if (!ncu) { int dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); }
            if (!per_cu[f] || per_cu_dyn[f] != dyn2) {
                if (dyn2 > 48 * 1024) (void)hipFuncSetAttribute(f ? (const void *)k_tile_build<2, true, true> : (const void *)k_tile_build<2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
                int nb = 0;
                if (f) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile_build<2, true, true>, TB2_THREADS, dyn2);
                else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tile_build<2, false, true>, TB2_THREADS, dyn2);
                per_cu[f] = nb > 0 ? nb : 1; per_cu_dyn[f] = dyn2;
            }
            const int resident = per_cu[f] * (ncu > 0 ? ncu : 256);
            if (g2.nactive > resident) tgrid2 = dim3(resident); else g2.queue = nullptr;      // (a launch of one round needs no queue)
        }
        if (g2.queue) {
            if (pt.group) hipLaunchKernelGGL((k_tile_build<2, true, true>), tgrid2, dim3(TB2_THREADS), dyn2, s, kargs(g2, 1, dbg));
            else hipLaunchKernelGGL((k_tile_build<2, false, true>), tgrid2, dim3(TB2_THREADS), dyn2, s, kargs(g2, 1, dbg));
        } else if (pt.group) {
            if (dyn2 > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_tile_build<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
            hipLaunchKernelGGL((k_tile_build<2, true>), tgrid2, dim3(TB2_THREADS), dyn2, s, kargs(g2, 1, dbg));
        } else {
            if (dyn2 > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_tile_build<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
            hipLaunchKernelGGL((k_tile_build<2, false>), tgrid2, dim3(TB2_THREADS), dyn2, s, kargs(g2, 1, dbg));
        }
    } else {
        BrickArgs g4 = g;
        g4.queue = nullptr;          // (the 4-brick fallback: one workgroup per brick)
        int split = 1;
        while (split < 4 && occupied * split * 2 <= 900) split *= 2;
        if (dbg >= 100) { split = dbg - 100; dbg = 0; }     // timing experiments: pair_debug 110 + split
        const dim3 tgrid((g.nactive * split + 7) / 8 * 8);
        const size_t dyn = (size_t)g.maxh * 16 + (size_t)BRK_WAVES * TB_G * n_col * 2 + TB_ROWPAD * 2;
        if (getenv("MESO_DEBUG_BUILD")) fprintf(stderr, "tile build: bricks %d split %d maxh %d n_col %d LDS %zu mbin %d %d %d\n", g.nactive, split, g.maxh, n_col, dyn, g.mbin[0], g.mbin[1], g.mbin[2]);
        if (pt.group) {
            if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_tile_build<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            hipLaunchKernelGGL((k_tile_build<4, true>), tgrid, dim3(BRK_THREADS), dyn, s, kargs(g4, split, dbg));
        } else {
            if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_tile_build<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            hipLaunchKernelGGL((k_tile_build<4, false>), tgrid, dim3(BRK_THREADS), dyn, s, kargs(g4, split, dbg));
        }
    }
    if (excl && excl->tagc && nlocal > 0) {
        const int nw = (nlocal + 63) / 64;                                  // one wave per 64 consecutive atoms
        hipLaunchKernelGGL(k_filter_exclusion, dim3((nw + 3) / 4), dim3(256), 0, s, *excl, nlocal, n_col, count, table, pt.nb_col,
                           pt.group ? pt.nback : nullptr, pt.group ? pt.back : nullptr);
    }
}

int tile_build_rowcap() { return TB_ROWCAP_MAX; }
int tile_build_queue_ints() { return TBQ_PITCH * (1 + TBQ_N); }

} // namespace meso
