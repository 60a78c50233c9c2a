// Neighbour rebuild of one rank in THREE launches (+ the list builder): reorder of the locals, ghost creation, ghost
// binning and the ghosts' merged pairs.
//
// What the reference does with host OpenMP + MPI + 7-8 radix passes per rebuild (MesoDomain::pbc domain_meso.cu:30-145,
// MesoAtom::sort_local atom_meso.cu:343-384, MesoComm::borders comm_meso.cu:41-186, binning_meso neighbor_meso.cu:535-711,
// gpu_merge_xvt atom_vec_meso.cu:142-167) and what rounds 1-2 of this repository did with a chain of 16 small dependent
// launches (count, scan x2, place, order, gather, border count/scan/fill, pack, ghost count, scan x2, place, order, ghost
// merge: ~95 us of dispatch latency at 32^3, where a whole rebuild interval is 285 us).  A dependent launch costs 2-5 us
// whatever it does (MI355X_MICROARCH.md, row "boundary"), a grid-wide hand-off inside a launch costs the same, so the only
// way to a cheaper rebuild is FEWER PHASES:
//
//   k_fr_count   lane = atom (old order): periodic wrap, extended code e = [border][Morton(bin)], rank inside the code from ONE
//                atomic per run of equal codes in a wave (the atoms arrive nearly sorted), the atom's old index into the code's
//                bucket (fixed capacity + a short overflow list), and the run into the total of its tile of 64 codes (hot
//                addresses are avoided: ~35 ns per same-address atomic made a per-supertile counter cost 70 us at 32^3);
//   k_fr_place   workgroup = tile of 64 codes (one brick of the list builder): first index of the tile = sum of the tile
//                totals in front of it (every workgroup adds them up itself: no scan launch, no look-back spin; beyond 4096
//                tiles a small extra launch sums them per supertile of 256 first),
//                estart of its codes, its atoms ordered by (sub-cell Morton key, old index) in LDS - the order the reference's
//                radix sort gives - and gathered to their new places together with the merged float4 pair; border atoms emit
//                their periodic images: ghost bin code, rank inside it (one atomic per run of equal ghost codes in a wave),
//                (new index, direction) into the ghost bucket;
//   k_fr_ghosts  workgroup = tile of 64 ghost codes: first ghost slot from the tile totals, gstart, ghosts ordered by
//                (source index, direction) - deterministic - and written in their FINAL slot order: x + shift, tag, type,
//                mask, the merged pair in the receiver's frame (what k_pack_border + k_ghost_count/place/order +
//                k_pack_forward did), send list, direction bytes and the image table of the step-boundary epilogue.
//
// No global atomics touch the neighbour rows (north_star); the atomics here count atoms per cell.  Counters are left clean for
// the next rebuild (the tile totals are double-buffered by rebuild parity: a tile zeroes the other buffer's entry).
#include "kernels.h"
#include "meso_device.h"

namespace meso {

#define FR_TILE 64            // codes per tile (= one brick of the list builder)
#define FR_SUPER 256          // tiles per supertile
#define FR_THREADS 256       // k_fr_ghosts, k_fr_super
#ifndef FR_PLACE_THREADS
#define FR_PLACE_THREADS 256  // k_fr_place: the ~600 atoms of a tile in one or two trips (few workgroups have atoms: latency counts)
#endif
#define FR_DIRECT_TILES 4096  // up to this many tiles every tile adds up the tile totals in front of it directly

__device__ inline u32 compact3(u32 x)      // inverse of bit_space3: every third bit
{
    x &= 0x09249249;
    x = (x ^ (x >> 2)) & 0x030c30c3;
    x = (x ^ (x >> 4)) & 0x0300f00f;
    x = (x ^ (x >> 8)) & 0xff0000ff;
    x = (x ^ (x >> 16)) & 0x000003ff;
    return x;
}

__device__ inline int fr_block_sum(int v, int *wsum)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int k = 0; k < (int)blockDim.x / 64; k++) t += wsum[k];
    __syncthreads();
    return t;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 1: count
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_fr_count(FusedArgs a)
{
    const int i = blockDim.x * blockIdx.x + threadIdx.x;
    const bool valid = i < a.n;
    u32 e = 0;
    if (valid) {
        double c[3] = {a.src.x[0][i], a.src.x[1][i], a.src.x[2][i]};
        if (a.wrap) {
            const int img = a.src.image[i];
            int im[3] = {img & 1023, (img >> 10) & 1023, img >> 20};
            bool moved = false;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (!a.per[d]) continue;
                const double p = a.boxhi[d] - a.boxlo[d];
                if (c[d] < a.boxlo[d]) { c[d] += p; im[d] = (im[d] - 1) & 1023; moved = true; }
                if (c[d] >= a.boxhi[d]) { c[d] -= p; c[d] = fmax(c[d], a.boxlo[d]); im[d] = (im[d] + 1) & 1023; moved = true; }
            }
            if (moved) {
                a.src.x[0][i] = c[0]; a.src.x[1][i] = c[1]; a.src.x[2][i] = c[2];
                a.src.image[i] = im[0] | (im[1] << 10) | (im[2] << 20);
            }
        }
        u32 b[3];
#pragma unroll
        for (int d = 0; d < 3; d++) b[d] = (u32)clampi((int)((c[d] - a.g.lo[d]) * a.g.bininv[d] + 1), 0, a.g.mbin[d]);
        e = interleave3(b[0], b[1], b[2]);
        const bool border = c[0] <= a.sl.lo[0] || c[0] >= a.sl.hi[0] || c[1] <= a.sl.lo[1] || c[1] >= a.sl.hi[1] || c[2] <= a.sl.lo[2] ||
                            c[2] >= a.sl.hi[2];
        if (border) e += (u32)a.M;
    }
    // rank inside the code, one atomic per run of equal codes; tile totals: one atomic per tile and wave
    const int rank = run_rank(e, valid, a.cnt);
    wave_group_add(e / FR_TILE, valid, a.ttot);
    if (!valid) return;
    if (rank < a.cap) a.bucket[(size_t)e * a.cap + rank] = i;
    else {
        const int o = atomicAdd(a.novf, 1);
        if (o < a.ovf_cap) { a.ovf[2 * o] = (int)e; a.ovf[2 * o + 1] = i; }
        else atomicMax(a.flags, 300000);
    }
}

// the tile's first index: supertile totals in front of its supertile + tile totals in front of it inside the supertile
__device__ inline int fr_tile_base(const int *__restrict__ ttot, const int *__restrict__ stot, int t, int *wsum)
{
    const int st = stot ? t / FR_SUPER : 0;
    int part = 0;
    for (int k = threadIdx.x; k < st; k += blockDim.x) part += stot[k];
    for (int k = st * FR_SUPER + (int)threadIdx.x; k < t; k += blockDim.x) part += ttot[k];
    return fr_block_sum(part, wsum);
}

// many tiles (large boxes): totals per supertile of 256 tiles, so that a tile adds up <= 256 + ntiles / 256 numbers
__global__ void __launch_bounds__(FR_THREADS) k_fr_super(const int *__restrict__ ttot, int ntiles, int *__restrict__ stot)
{
    __shared__ int wsum[FR_THREADS / 64];
    const int k = blockIdx.x * FR_SUPER + (int)threadIdx.x;
    const int v = fr_block_sum(k < ntiles ? ttot[k] : 0, wsum);
    if (threadIdx.x == 0) stot[blockIdx.x] = v;
}

// members of code e beyond the bucket's capacity: the r-th one (r >= cap) in list order... any order is fine, the ordering
// pass sorts by (key, index); the list is short (normally empty), every lane that needs it scans it
__device__ inline int fr_overflow_member(const int *__restrict__ ovf, int novf, int e, int r)
{
    int seen = 0;
    for (int o = 0; o < novf; o++)
        if (ovf[2 * o] == e) {
            if (seen == r) return ovf[2 * o + 1];
            seen++;
        }
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 2: place + gather + ghost emission
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FR_PLACE_THREADS) k_fr_place(FusedArgs a)
{
    extern __shared__ unsigned long long fr_pairs[];      // (sub-cell key << 32) | old index, per pass
    __shared__ int lstart[FR_TILE + 1];
    __shared__ int wsum[FR_PLACE_THREADS / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int code0 = t * FR_TILE;
    const int base = fr_tile_base(a.ttot, a.stot, t, wsum);
    // counts of my codes -> local starts (one wave), estart, clean counters
    if (tid < 64) {
        const int c = a.cnt[code0 + tid];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        lstart[tid] = incl - c;
        if (tid == 63) lstart[64] = incl;
        a.estart[code0 + tid] = base + incl - c;
        if (tid == 63 && code0 + 64 == 2 * a.M) a.estart[2 * a.M] = base + incl;
        a.cnt[code0 + tid] = 0;
    }
    if (tid == 0) a.ttot_next[t] = 0;
    __syncthreads();
    const int total = lstart[FR_TILE];
    if (total == 0) return;
    const int novf = *a.novf;
    const int sub_bits = a.sub_bits, res = 1 << (sub_bits / 3);
    const bool border_tile = code0 >= a.M;
    // passes over sub-ranges of codes whose atoms fit the LDS stage (one pass at ordinary densities)
    int cb = 0;
    while (cb < FR_TILE) {
        int ce = cb + 1;
        while (ce < FR_TILE && lstart[ce + 1] - lstart[cb] <= a.lds_cap) ce++;
        const int s0 = lstart[cb], ns = lstart[ce] - s0;
        const bool staged = ns <= a.lds_cap;          // (a single code beyond the stage: ordered straight from global memory)
        for (int p = tid; p < ns; p += FR_PLACE_THREADS) {
            int lo = cb, hi = ce;                     // code of slot s0 + p: largest c with lstart[c] <= s0 + p
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (lstart[mid] <= s0 + p) lo = mid; else hi = mid;
            }
            const int r = s0 + p - lstart[lo];
            const int e = code0 + lo;
            int j = r < a.cap ? a.bucket[(size_t)e * a.cap + r] : fr_overflow_member(a.ovf, min(novf, a.ovf_cap), e, r - a.cap);
            u32 key = 0;
            if (j >= 0) {
                const double c[3] = {a.src.x[0][j], a.src.x[1][j], a.src.x[2][j]};
                u32 sc[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const u32 b = (u32)clampi((int)((c[d] - a.g.lo[d]) * a.g.bininv[d] + 1), 0, a.g.mbin[d]);
                    sc[d] = (u32)clampi((int)((c[d] - a.g.lo[d] - ((double)b - 1) * a.g.binsize[d]) * (res * a.g.bininv[d])), 0, res);
                }
                key = interleave3(sc[0], sc[1], sc[2]);
            } else { j = 0; atomicMax(a.flags, 300001); }
            if (staged) fr_pairs[p] = ((unsigned long long)key << 32) | (u32)j;
            else a.scratch[(size_t)base + s0 + p] = ((unsigned long long)key << 32) | (u32)j;
        }
        __syncthreads();
        // (whole waves take every trip: the ghost emission ranks its atomics per run of equal ghost codes in a wave)
        for (int p = tid; p < ((ns + 63) & ~63); p += FR_PLACE_THREADS) {
            const bool act = p < ns;
            int j = 0, n = 0, lo_code = 0;
            if (act) {
                int lo = cb, hi = ce;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (lstart[mid] <= s0 + p) lo = mid; else hi = mid;
                }
                lo_code = lo;
                const int sb = lstart[lo] - s0, se = lstart[lo + 1] - s0;
                const unsigned long long *pairs = staged ? fr_pairs : a.scratch + (size_t)base + s0;
                const unsigned long long mine = pairs[p];
                int pos = 0;
                for (int q = sb; q < se; q++) pos += pairs[q] < mine ? 1 : 0;
                j = (int)(u32)mine;
                n = base + s0 + sb + pos;             // the atom's new place
                permute_one(a.src, a.dst, j, n, a.with_f, a.mg);
                if (a.perm) a.perm[n] = j;
            }
            if (border_tile && a.gttot) {
                // periodic images of a border atom: k_fr_ghosts PULLS the ghosts of a ghost cell from the cell they are images of,
                // so all that is needed here is the number of ghosts per tile of 64 ghost cells (its first slot is the sum of the
                // totals in front of it).  The ghost cell of an image is the geometric image of the atom's own cell.  No returning
                // atomics: nothing waits (a chain of rank atomics, one per direction, cost 25-40 us here)
                int fl = 0;
                if (act) fl = near_flags(a.src.x[0][j], a.src.x[1][j], a.src.x[2][j], a.sl.lo, a.sl.hi);
                if (__ballot(fl != 0) != 0ull) {
                    const u32 lc = (u32)(code0 - a.M) + (u32)lo_code;          // Morton code of the atom's cell
                    const int bx = (int)compact3(lc), by = (int)compact3(lc >> 1), bz = (int)compact3(lc >> 2);
#pragma unroll 1
                    for (int dir = 0; dir < 27; dir++) {
                        const int sx = dir % 3 - 1, sy = (dir / 3) % 3 - 1, sz = dir / 9 - 1;
                        // an image sent up (s = +1) comes from the last cell below the high face and lands in ghost cell 0; k_fr_ghosts
                        // pulls exactly these (an atom ON the slab plane of a box whose cells are exactly one ghost cutoff wide can
                        // sit in the cell next to it: not an image for either kernel)
                        const bool cellok = (sx == 0 || bx == (sx > 0 ? a.g.mbin[0] - 2 : 1)) && (sy == 0 || by == (sy > 0 ? a.g.mbin[1] - 2 : 1)) &&
                                            (sz == 0 || bz == (sz > 0 ? a.g.mbin[2] - 2 : 1));
                        const bool em = dir != 13 && fl && in_dir(fl, dir) && ((a.dir_mask >> dir) & 1u) && cellok;
                        if (__ballot(em) == 0ull) continue;
                        const u32 gc = interleave3((u32)(sx == 0 ? bx : (sx > 0 ? 0 : a.g.mbin[0] - 1)), (u32)(sy == 0 ? by : (sy > 0 ? 0 : a.g.mbin[1] - 1)),
                                                   (u32)(sz == 0 ? bz : (sz > 0 ? 0 : a.g.mbin[2] - 1)));
                        wave_group_add(em ? gc / FR_TILE : 0u, em, a.gttot);
                    }
                }
            }
        }
        __syncthreads();
        cb = ce;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3: ghosts
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FR_THREADS) k_fr_ghosts(FusedArgs a)
{
    extern __shared__ unsigned char fr_em[];      // per candidate of the pass: 1 = becomes a ghost of this cell
    __shared__ int cstart[FR_TILE + 1];           // candidates (atoms of the source cell) in front of each of my cells
    __shared__ int gl[FR_TILE + 1];               // ghosts in front of each of my cells
    __shared__ int src0[FR_TILE];                 // first atom of the source cell (new order), direction
    __shared__ int sdir[FR_TILE];
    __shared__ int wsum[FR_THREADS / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int ntiles = a.M / FR_TILE;
    const int code0 = t * FR_TILE;
    const int base = fr_tile_base(a.gttot, a.gstot, t, wsum);
    if (tid < 64) {
        // the cell my ghosts are images of, and the direction they were sent in (geometry only)
        const u32 g = (u32)(code0 + tid);
        const int b[3] = {(int)compact3(g), (int)compact3(g >> 1), (int)compact3(g >> 2)};
        int sb[3], sd[3];
        bool ghostcell = false, inside = true;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            inside = inside && b[d] < a.g.mbin[d];
            if (b[d] == 0) { sb[d] = a.g.mbin[d] - 2; sd[d] = 1; ghostcell = true; }                 // below the low face: sent up
            else if (b[d] == a.g.mbin[d] - 1) { sb[d] = 1; sd[d] = -1; ghostcell = true; }         // above the high face: sent down
            else { sb[d] = b[d]; sd[d] = 0; }
        }
        const int dir = (sd[0] + 1) + 3 * (sd[1] + 1) + 9 * (sd[2] + 1);
        int nc = 0, first = 0;
        if (ghostcell && inside && ((a.dir_mask >> dir) & 1u)) {
            const u32 ms = interleave3((u32)sb[0], (u32)sb[1], (u32)sb[2]);
            first = a.estart[a.M + ms];
            nc = a.estart[a.M + ms + 1] - first;
        }
        src0[tid] = first; sdir[tid] = dir;
        int incl = nc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        cstart[tid] = incl - nc;
        if (tid == 63) cstart[64] = incl;
        gl[tid] = 0;
    }
    if (tid == 0) { a.gttot_next[t] = 0; gl[64] = 0; }
    __syncthreads();
    // candidates -> ghost flags, ghosts per cell (passes over sub-ranges of cells whose candidates fit the LDS stage)
    int cb = 0;
    while (cb < FR_TILE) {
        int ce = cb + 1;
        while (ce < FR_TILE && cstart[ce + 1] - cstart[cb] <= a.lds_cap) ce++;
        const int s0 = cstart[cb], ns = cstart[ce] - s0;
        if (ns > a.lds_cap) { if (tid == 0) atomicMax(a.flags, 300003); cb = ce; continue; }      // one cell beyond the stage
        for (int p = tid; p < ns; p += FR_THREADS) {
            int lo = cb, hi = ce;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (cstart[mid] <= s0 + p) lo = mid; else hi = mid;
            }
            const int j = src0[lo] + (s0 + p - cstart[lo]);
            const int fl = near_flags(a.dst.x[0][j], a.dst.x[1][j], a.dst.x[2][j], a.sl.lo, a.sl.hi);
            const bool em = fl && in_dir(fl, sdir[lo]);
            fr_em[p] = em ? 1 : 0;
            if (em) atomicAdd(&gl[lo], 1);        // (LDS)
        }
        __syncthreads();
        cb = ce;
    }
    // ghosts per cell -> first slot of each cell
    if (tid < 64) {
        const int c = gl[tid];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        a.gstart[code0 + tid] = base + incl - c;
        gl[tid] = incl - c;
        if (tid == 63) gl[64] = incl;
    }
    __syncthreads();
    const int total = gl[FR_TILE];
    if (t == ntiles - 1 && tid < 28) {
        // the rebuild's counts: on the device for the kernels that follow, in pinned host memory for the engine (four words: every
        // store to host memory is a trip over the host link on this tile's critical path)
        const int ng = base + total;
        a.dir_start[tid] = tid == 27 ? ng : 0;
        if (tid == 0) {
            a.gstart[a.M] = ng;
            int f = a.flags[0];
            if (ng > a.ghost_cap) f = 200000;
            if (f) a.flags[0] = f;
            *a.novf = 0;
            a.report[8] = f;
            a.report[9] = a.estart[a.M];            // n_bulk
            a.report[10] = a.flags[5];              // fullest brick neighbourhood of the previous list build
            a.report[16 + 27] = ng;
        }
    }
    if (total == 0) return;
    // second walk over the candidates: the ghosts, in (cell, source index) order - deterministic, no atomics
    cb = 0;
    while (cb < FR_TILE) {
        int ce = cb + 1;
        while (ce < FR_TILE && cstart[ce + 1] - cstart[cb] <= a.lds_cap) ce++;
        const int s0 = cstart[cb], ns = cstart[ce] - s0;
        if (ns > a.lds_cap) { cb = ce; continue; }
        const bool refill = !(cb == 0 && ce == FR_TILE);      // several passes: the flags of this pass again
        if (refill) {
            __syncthreads();
            for (int p = tid; p < ns; p += FR_THREADS) {
                int lo = cb, hi = ce;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (cstart[mid] <= s0 + p) lo = mid; else hi = mid;
                }
                const int j = src0[lo] + (s0 + p - cstart[lo]);
                const int fl = near_flags(a.dst.x[0][j], a.dst.x[1][j], a.dst.x[2][j], a.sl.lo, a.sl.hi);
                fr_em[p] = (fl && in_dir(fl, sdir[lo])) ? 1 : 0;
            }
            __syncthreads();
        }
        for (int p = tid; p < ns; p += FR_THREADS) {
            if (!fr_em[p]) continue;
            int lo = cb, hi = ce;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (cstart[mid] <= s0 + p) lo = mid; else hi = mid;
            }
            const int q0 = cstart[lo] - s0;
            int rank = 0;
            for (int q = q0; q < p; q++) rank += fr_em[q];
            const int k = base + gl[lo] + rank;       // ghost slot
            if (k >= a.ghost_cap) continue;           // (reported through flags[0] by the last tile)
            const int j = src0[lo] + (p - q0), d = sdir[lo];
            const int gi = a.n + k;
            // pack_border_vel (atom_vec_dpd_atomic_meso.cu:61-135): x + shift, tag, type, mask
            const double gx = a.dst.x[0][j] + a.sh.s[d][0], gy = a.dst.x[1][j] + a.sh.s[d][1], gz = a.dst.x[2][j] + a.sh.s[d][2];
            a.dst.x[0][gi] = gx; a.dst.x[1][gi] = gy; a.dst.x[2][gi] = gz;
            const int tg = a.dst.tag[j], ty = a.dst.type[j];
            a.dst.tag[gi] = tg; a.dst.type[gi] = ty; a.dst.mask[gi] = a.dst.mask[j];
            // pack_comm_vel + gpu_merge_xvt for the ghost (k_pack_forward's expressions: same bits)
            float4 c, v;
            c.x = (float)(gx - a.ce.c[d][0]); c.y = (float)(gy - a.ce.c[d][1]); c.z = (float)(gz - a.ce.c[d][2]);
            c.w = __uint_as_float((u32)(ty - 1));
            v.x = (float)a.dst.v[0][j]; v.y = (float)a.dst.v[1][j]; v.z = (float)a.dst.v[2][j];
            v.w = __uint_as_float(signature(a.mg.seed, tg, v.x, v.y, v.z));
            a.mg.coord4[gi] = c;
            a.mg.veloc4[gi] = v;
            a.sendlist[k] = j;
            a.senddir[k] = (unsigned char)d;
            if (a.img_cnt) {      // where the images of atom j live (the step-boundary epilogue refreshes them between rebuilds)
                const int slot = atomicAdd(&a.img_cnt[j], 1);
                if (slot < 8) a.img[(size_t)j * 8 + slot] = gi | (d << 26);
            }
        }
        cb = ce;
    }
}

void launch_fused_rebuild(const FusedArgs &a, hipStream_t s)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_fr_count, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
    const int ntl = 2 * a.M / FR_TILE, ntg = a.M / FR_TILE;
    if (a.stot) hipLaunchKernelGGL(k_fr_super, dim3((ntl + FR_SUPER - 1) / FR_SUPER), dim3(FR_THREADS), 0, s, a.ttot, ntl, a.stot);
    const size_t dyn2 = (size_t)a.lds_cap * 8, dyn3 = (size_t)a.lds_cap;
    if (dyn2 > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_fr_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
    hipLaunchKernelGGL(k_fr_place, dim3(2 * a.M / FR_TILE), dim3(FR_PLACE_THREADS), dyn2, s, a);
    if (a.gttot && a.gstot) hipLaunchKernelGGL(k_fr_super, dim3((ntg + FR_SUPER - 1) / FR_SUPER), dim3(FR_THREADS), 0, s, a.gttot, ntg, a.gstot);
    if (a.gttot) hipLaunchKernelGGL(k_fr_ghosts, dim3(ntg), dim3(FR_THREADS), dyn3, s, a);
}

int fused_direct_tiles() { return FR_DIRECT_TILES; }
int fused_tile_codes() { return FR_TILE; }
int fused_super_tiles() { return FR_SUPER; }

} // namespace meso
