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

#ifndef FR_TILE
#define FR_TILE 64            // cells per tile of k_fr_place (<= 64)
#endif
#ifndef FR_U
#define FR_U 1                // trips of k_fr_place whose loads are issued together (measured 1..4: the registers of more
                              // trips cost occupancy - 64^3 141 us with 1, 158 us with 4; 32^3 the same)
#endif
#ifndef FR_GTILE
#define FR_GTILE 32           // ghost cells per tile of k_fr_ghosts (<= 64)
#endif
static_assert(FR_GTILE == FR_COUNT_GTILE, "ghost tile of the count's image booking (meso_device.h) and of k_fr_ghosts");
#define FR_SUPER 256          // tiles per supertile
#define FR_THREADS 256       // k_fr_ghosts, k_fr_super
#ifndef FR_PLACE_THREADS
#define FR_PLACE_THREADS 256  // k_fr_place: the ~600 atoms of a tile in one or two trips (few workgroups have atoms: latency counts)
#endif
#ifndef FR_DIRECT_TILES
#define FR_DIRECT_TILES 4096  // up to this many tiles every tile adds up the tile totals in front of it directly
#endif

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
FrCountArgs fused_count_args(const FusedArgs &a)
{
    FrCountArgs c;
    for (int d = 0; d < 3; d++) {
        c.x[d] = a.src.x[d];
        c.boxlo[d] = a.boxlo[d]; c.boxhi[d] = a.boxhi[d]; c.per[d] = a.per[d];
        c.sl_lo[d] = a.sl.lo[d]; c.sl_hi[d] = a.sl.hi[d];
    }
    c.image = a.src.image;
    c.wrap = a.wrap;
    c.g = a.g;
    c.sub_bits = a.sub_bits; c.M = a.M;
    c.cnt = a.cnt; c.cap = a.cap;
    c.bucket = a.bucket; c.ovf = a.ovf; c.novf = a.novf; c.ovf_cap = a.ovf_cap;
    c.ttot = a.ttot;
    c.gttot = a.img_booked ? a.gttot : nullptr; c.dir_mask = a.dir_mask;
    c.flags = a.flags;
    return c;
}

__global__ void __launch_bounds__(256) k_fr_count(FrCountArgs a, const int *__restrict__ skip, int skip_n, int n)
{
    static_assert(FR_COUNT_TILE == FR_TILE, "tile of the count and of the placing kernel");
    const int i = blockDim.x * blockIdx.x + threadIdx.x;
    const bool valid = i < n && !(skip && i < skip_n && skip[i] != 13);
    double c[3] = {0.0, 0.0, 0.0};
    if (valid) { c[0] = a.x[0][i]; c[1] = a.x[1][i]; c[2] = a.x[2][i]; }
    fr_count_atom(a, i, valid, c[0], c[1], c[2]);
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
__device__ inline unsigned long long fr_overflow_member(const unsigned long long *__restrict__ ovf, int novf, int e, int r)
{
    int seen = 0;
    for (int o = 0; o < novf; o++)
        if (ovf[2 * o] == (unsigned long long)e) {
            if (seen == r) return ovf[2 * o + 1];
            seen++;
        }
    return ~0ull;
}

// ---------------------------------------------------------------------------------------------------------------------------
// 2: place + gather + ghost emission
// ---------------------------------------------------------------------------------------------------------------------------
#ifndef FR_PLACE_OCC
#define FR_PLACE_OCC 8        // waves per SIMD the placing kernel is compiled for: its gather is bound by the bytes in flight per CU
#endif
// LISTS: bonded systems - the general gather (topology lists travel too); an instantiation of its own so that its registers do not
// count against the common path
// GATHER false (large one-rank boxes): the kernel only ORDERS - it writes the permutation new place -> old place - and the payload
// moves in a streaming pass of its own (k_fr_gather: lane = atom, nothing but the permutation between it and its loads).  The
// fused form keeps a tile's ~600 atoms x 100 bytes behind four dependent round trips (tile base, counts, buckets, gather) with
// eight tiles resident per CU: 77 us for 190 MB at 64^3, where the streaming gather of round 2 moved the same bytes in 38.
template <bool LISTS, bool GATHER>
__global__ void __launch_bounds__(FR_PLACE_THREADS, LISTS ? 5 : FR_PLACE_OCC) k_fr_place(FusedArgs a)
{
    extern __shared__ unsigned long long fr_pairs[];      // (sub-cell key << 32) | old index, per pass
    __shared__ int lstart[FR_TILE + 1];
    __shared__ int wsum[FR_PLACE_THREADS / 64];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int code0 = t * FR_TILE;
    const int base = fr_tile_base(a.ttot, a.stot, t, wsum);
    // counts of my codes -> local starts (one wave), estart, clean counters
    if (tid < 64) {
        const int c = tid < FR_TILE ? a.cnt[code0 + tid] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        if (tid < FR_TILE) {
            lstart[tid] = incl - c;
            a.estart[code0 + tid] = base + incl - c;
            a.cnt[code0 + tid] = 0;
        }
        if (tid == FR_TILE - 1) {
            lstart[FR_TILE] = incl;
            if (code0 + FR_TILE == 2 * a.M) a.estart[2 * a.M] = base + incl;
        }
    }
    if (tid == 0) a.ttot_next[t] = 0;
    __syncthreads();
    const int total = lstart[FR_TILE];
    if (total == 0) return;
    const int novf = *a.novf;
    const bool border_tile = code0 >= a.M;
    // passes over sub-ranges of codes whose atoms fit the LDS stage (one pass at ordinary densities)
    int cb = 0;
    while (cb < FR_TILE) {
        // (the usual case first: the rest of the tile fits the stage - a walk over the 64 starts is 64 dependent LDS reads, 3 us)
        int ce = FR_TILE;
        if (lstart[FR_TILE] - lstart[cb] > a.lds_cap) {
            ce = cb + 1;
            while (ce < FR_TILE && lstart[ce + 1] - lstart[cb] <= a.lds_cap) ce++;
        }
        const int s0 = lstart[cb], ns = lstart[ce] - s0;
        const bool staged = ns <= a.lds_cap;          // (a single code beyond the stage: ordered straight from global memory)
        // FR_U trips at a time, every load of the group in flight before the first store: a wave's loads wait for its older stores
        // (one vmcnt counter), so trip after trip cost one store acknowledgement + one load round trip each (~3 us)
        for (int p0 = tid; p0 < ns; p0 += FR_U * FR_PLACE_THREADS) {
            unsigned long long ent[FR_U];
#pragma unroll
            for (int u = 0; u < FR_U; u++) {
                const int p = p0 + u * FR_PLACE_THREADS;
                ent[u] = 0;
                if (p < ns) {
                    int lo = cb, hi = ce;             // code of slot s0 + p: largest c with lstart[c] <= s0 + p
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (lstart[mid] <= s0 + p) lo = mid; else hi = mid;
                    }
                    const int r = s0 + p - lstart[lo];
                    const int e = code0 + lo;
                    ent[u] = r < a.cap ? a.bucket[(size_t)e * a.cap + r] : fr_overflow_member(a.ovf, min(novf, a.ovf_cap), e, r - a.cap);
                }
            }
#pragma unroll
            for (int u = 0; u < FR_U; u++) {
                const int p = p0 + u * FR_PLACE_THREADS;
                if (p < ns) {
                    if (ent[u] == ~0ull) { ent[u] = 0; atomicMax(a.flags, 300001); }
                    if (staged) fr_pairs[p] = ent[u];
                    else a.scratch[(size_t)base + s0 + p] = ent[u];
                }
            }
        }
        __syncthreads();
        // (whole waves take every trip: the image counting below is a wave-level operation)
        constexpr bool lists = LISTS;
        for (int p0 = tid; p0 < ((ns + 63) & ~63); p0 += FR_U * FR_PLACE_THREADS) {
            int jj[FR_U], nn[FR_U], cc[FR_U];
            double X[FR_U][3], V[FR_U][3], F[FR_U][3], MS[FR_U];
            int TG[FR_U], TY[FR_U], MK[FR_U], IM[FR_U];
#pragma unroll
            for (int u = 0; u < FR_U; u++) {
                const int p = p0 + u * FR_PLACE_THREADS;
                jj[u] = -1; nn[u] = 0; cc[u] = 0;
                if (p < ns) {
                    int lo = cb, hi = ce;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (lstart[mid] <= s0 + p) lo = mid; else hi = mid;
                    }
                    cc[u] = lo;
                    const int sb = lstart[lo] - s0, se = lstart[lo + 1] - s0;
                    const unsigned long long *pairs = staged ? fr_pairs : a.scratch + (size_t)base + s0;
                    const unsigned long long mine = pairs[p];
                    int pos = 0;
                    for (int q = sb; q < se; q++) pos += pairs[q] < mine ? 1 : 0;
                    jj[u] = (int)(u32)mine;
                    nn[u] = base + s0 + sb + pos;     // the atom's new place
                }
            }
            if (!GATHER) {
#pragma unroll
                for (int u = 0; u < FR_U; u++)
                    if (jj[u] >= 0) {
                        a.perm[nn[u]] = jj[u];
                        if (a.merged_ghosts && border_tile && a.mg.zero) a.mg.zero[nn[u]] = 0;      // (see fr_gather_block)
                    }
                continue;
            }
#pragma unroll
            for (int u = 0; u < FR_U; u++) {
                X[u][0] = X[u][1] = X[u][2] = 0.0;
                if (jj[u] >= 0 && !lists) {
                    const int j = jj[u];
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        X[u][d] = a.src.x[d][j]; V[u][d] = a.src.v[d][j];
                        if (a.with_f) F[u][d] = a.src.f[d][j];
                    }
                    TG[u] = a.src.tag[j]; TY[u] = a.src.type[j]; MK[u] = a.src.mask[j]; IM[u] = a.src.image[j]; MS[u] = a.src.mass[j];
                }
            }
#pragma unroll
            for (int u = 0; u < FR_U; u++) {
                if (jj[u] < 0) continue;
                const int j = jj[u], n = nn[u];
                if (lists) {                          // bonded systems: the general gather (topology lists travel too)
                    permute_one(a.src, a.dst, j, n, a.with_f, a.mg, X[u]);
                } else {
                    // permute_one (meso_device.h) with the loads hoisted: the same values to the same places
                    if (a.mg.zero) a.mg.zero[n] = 0;
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        a.dst.x[d][n] = X[u][d]; a.dst.v[d][n] = V[u][d];
                        if (a.with_f) a.dst.f[d][n] = F[u][d];
                    }
                    float4 c, v;
                    c.x = (float)(X[u][0] - a.mg.cx); c.y = (float)(X[u][1] - a.mg.cy); c.z = (float)(X[u][2] - a.mg.cz);
                    c.w = __uint_as_float((u32)(TY[u] - 1));
                    v.x = (float)V[u][0]; v.y = (float)V[u][1]; v.z = (float)V[u][2];
                    v.w = __uint_as_float(signature(a.mg.seed, TG[u], v.x, v.y, v.z));
                    a.mg.coord4[n] = c;
                    a.mg.veloc4[n] = v;
                    a.dst.tag[n] = TG[u]; a.dst.type[n] = TY[u]; a.dst.mask[n] = MK[u]; a.dst.image[n] = IM[u]; a.dst.mass[n] = MS[u];
                }
                if (a.perm) a.perm[n] = j;
            }
            if (border_tile && a.gttot && !a.img_booked) {
                // periodic images of a border atom, booked per tile of ghost cells (book_images, meso_device.h) - unless the count did
#pragma unroll
                for (int u = 0; u < FR_U; u++) {
                    if (__ballot(jj[u] >= 0) == 0ull) continue;
                    u32 emask = 0;
                    const u32 lc = (u32)(code0 - a.M) + (u32)cc[u];           // Morton code of the atom's cell
                    const int bx = (int)compact3(lc), by = (int)compact3(lc >> 1), bz = (int)compact3(lc >> 2);
                    if (jj[u] >= 0) emask = image_mask(near_flags(X[u][0], X[u][1], X[u][2], a.sl.lo, a.sl.hi), bx, by, bz, a.g.mbin, a.dir_mask);
                    book_images(emask, bx, by, bz, a.g.mbin, a.gttot);
                }
            }
        }
        __syncthreads();
        cb = ce;
    }
}

// 2b: the payload of the order-only form: atom perm[n] of the old order becomes atom n of the new one (lane = n; gpu_permute_copy,
// atom_vec_meso.h:11-67, with gpu_merge_xvt folded in), border atoms book their periodic images per tile of ghost cells exactly as
// the fused form does (the cell from the coordinate, as k_fr_count computed it)
// (LISTS: bonded systems - the general gather, topology lists travel too: permute_one, meso_device.h)
template <bool LISTS>
__device__ __forceinline__ void fr_gather_block(const FusedArgs &a, const int bidx)
{
    const int n = (int)blockDim.x * bidx + (int)threadIdx.x;
    const bool valid = n < a.n;
    // A cell bucket + the overflow list too small (flag >= 300000, raised by the count or the ordering kernel in front of this launch):
    // atoms were lost on the way and the permutation has holes that hold whatever the memory held before - nothing is gathered
    // through it.  The engine deepens the buckets and builds again from the arrays this launch would have read (Engine::reneighbor,
    // prepare_redo).
    if (*a.flags >= 300000) return;
    double X[3] = {0.0, 0.0, 0.0};
    if (valid && LISTS) {
        MergeOut mg = a.mg;
        if (a.merged_ghosts && n >= a.estart[a.M]) mg.zero = nullptr;      // (border atoms: cleared by the ordering kernel, see below)
        permute_one(a.src, a.dst, a.perm[n], n, a.with_f, mg, X);
    } else if (valid) {
        const int j = a.perm[n];
        double V[3], F[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            X[d] = a.src.x[d][j]; V[d] = a.src.v[d][j];
            if (a.with_f) F[d] = a.src.f[d][j];
        }
        const int TG = a.src.tag[j], TY = a.src.type[j], MK = a.src.mask[j], IM = a.src.image[j];
        const double MS = a.src.mass[j];
        // (ghost tiles of the same launch write the counters of border atoms: those were cleared by the ordering kernel)
        if (a.mg.zero && !(a.merged_ghosts && n >= a.estart[a.M])) a.mg.zero[n] = 0;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            a.dst.x[d][n] = X[d]; a.dst.v[d][n] = V[d];
            if (a.with_f) a.dst.f[d][n] = F[d];
        }
        float4 c, v;
        c.x = (float)(X[0] - a.mg.cx); c.y = (float)(X[1] - a.mg.cy); c.z = (float)(X[2] - a.mg.cz);
        c.w = __uint_as_float((u32)(TY - 1));
        v.x = (float)V[0]; v.y = (float)V[1]; v.z = (float)V[2];
        v.w = __uint_as_float(signature(a.mg.seed, TG, v.x, v.y, v.z));
        a.mg.coord4[n] = c;
        a.mg.veloc4[n] = v;
        a.dst.tag[n] = TG; a.dst.type[n] = TY; a.dst.mask[n] = MK; a.dst.image[n] = IM; a.dst.mass[n] = MS;
    }
    if (!a.gttot || a.img_booked) return;
    // (whole waves: the image counting is a wave-level operation; the border section starts at estart[M], written by the ordering kernel)
    const int nb = a.estart[a.M];
    const bool bord = valid && n >= nb;
    if (__ballot(bord) == 0ull) return;
    u32 emask = 0;
    int bx = 0, by = 0, bz = 0;
    if (bord) {
        bx = clampi((int)((X[0] - a.g.lo[0]) * a.g.bininv[0] + 1), 0, a.g.mbin[0]);
        by = clampi((int)((X[1] - a.g.lo[1]) * a.g.bininv[1] + 1), 0, a.g.mbin[1]);
        bz = clampi((int)((X[2] - a.g.lo[2]) * a.g.bininv[2] + 1), 0, a.g.mbin[2]);
        emask = image_mask(near_flags(X[0], X[1], X[2], a.sl.lo, a.sl.hi), bx, by, bz, a.g.mbin, a.dir_mask);
    }
    book_images(emask, bx, by, bz, a.g.mbin, a.gttot);
}

template <bool LISTS> __global__ void __launch_bounds__(256) k_fr_gather(FusedArgs a) { fr_gather_block<LISTS>(a, (int)blockIdx.x); }

// ---------------------------------------------------------------------------------------------------------------------------
// 3: ghosts
// ---------------------------------------------------------------------------------------------------------------------------
// (bidx: the tile's place in the launch; via_perm: the gather of this rebuild runs in the SAME launch - the atoms a ghost is an image
// of are read from the old order through the permutation the ordering kernel wrote)
template <bool VIA_PERM>
__device__ __forceinline__ void fr_ghosts_tile(const FusedArgs &a, const int bidx)
{
    extern __shared__ unsigned char fr_em[];      // per candidate of the pass: 1 = becomes a ghost of this cell
    __shared__ int cstart[FR_GTILE + 1];           // candidates (atoms of the source cell) in front of each of my cells
    __shared__ int gl[FR_GTILE + 1];               // ghosts in front of each of my cells
    __shared__ int src0[FR_GTILE];                 // first atom of the source cell (new order), direction
    __shared__ int sdir[FR_GTILE];
    __shared__ int scell[FR_GTILE];                // the source cell's coordinates, 10 bits each
    __shared__ int wsum[FR_THREADS / 64];
    // (gorder: only the tiles that hold ghost cells run - nine tenths of the tiles of a large box are interior; the list builder
    // then takes a cell's ghosts from (gstart, gcnt) of ghost cells only)
    const int t = a.gorder ? a.gorder[bidx] : bidx, tid = threadIdx.x;
    const bool reorder_lost = *a.flags >= 300000;      // (the reorder in front lost atoms: see fr_gather_block)
    const int ntiles = a.M / FR_GTILE;
    const int code0 = t * FR_GTILE;
    int nc = 0;
    if (tid < FR_GTILE) {
        // the cell my ghosts are images of, and the direction they were sent in (geometry only); its atoms sit in the border
        // section of the new order (these loads are in flight while the tile's first slot is added up)
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
        int first = 0;
        if (ghostcell && inside && ((a.dir_mask >> dir) & 1u)) {
            const u32 ms = interleave3((u32)sb[0], (u32)sb[1], (u32)sb[2]);
            first = a.estart[a.M + ms];
            nc = a.estart[a.M + ms + 1] - first;
        }
        src0[tid] = first; sdir[tid] = dir; scell[tid] = sb[0] | (sb[1] << 10) | (sb[2] << 20);
    }
    const int base = fr_tile_base(a.gttot, a.gstot, t, wsum);
    if (bidx == 0) {
        // the rebuild's counts, reported by the tile that starts first: on the device for the kernels that follow, in pinned host
        // memory for the engine (four words: every store to host memory is a trip over the host link)
        int part = 0;
        for (int k = tid; k < ntiles; k += FR_THREADS) part += a.gttot[k];
        const int ng = fr_block_sum(part, wsum);
        if (tid < 28) a.dir_start[tid] = tid == 27 ? ng : 0;
        if (tid == 0) {
            a.gstart[a.M] = ng;
            // (another tile of this launch may be raising an overflow code right now: the write-back is an atomic maximum, never a
            // plain store that could undo it; a code raised behind this snapshot stays in flags[0] - check_overflow at the end of
            // run() and the next rebuild's report see it: the run fails, it never carries on with a truncated ghost list)
            int f = a.flags[0];
            if (ng > a.ghost_cap) f = 200000;
            if (f) atomicMax(a.flags, f);
            *a.novf = 0;
            a.report[8] = f;
            a.report[9] = a.estart[a.M];            // n_bulk
            a.report[10] = a.flags[5];              // fullest brick neighbourhood of the previous list build
            a.report[11] = a.flags[6];              // ... a 2-brick neighbourhood neared its stage
            a.report[16 + 27] = ng;
            // (the host polls this word instead of waiting for an event behind the launch: an event record between two kernels of a
            // stream is a marker packet the command processor takes ~6 us to pass - a bubble in front of the list builder on every
            // rebuild, profiles/r06_notes.md section 10)
            if (a.report_seq) { __threadfence_system(); __hip_atomic_store(&a.report[12], a.report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
    }
    if (tid < 64) {
        int incl = nc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        if (tid < FR_GTILE) { cstart[tid] = incl - nc; gl[tid] = 0; }
        if (tid == FR_GTILE - 1) cstart[FR_GTILE] = incl;
    }
    if (tid == 0) { a.gttot_next[t] = 0; gl[FR_GTILE] = 0; }
    __syncthreads();
    if (reorder_lost) return;      // (the report above went out; no ghost is read through a permutation with holes)
    // candidates -> ghost flags, ghosts per cell (passes over sub-ranges of cells whose candidates fit the LDS stage)
    int cb = 0;
    while (cb < FR_GTILE) {
        int ce = FR_GTILE;
        if (cstart[FR_GTILE] - cstart[cb] > a.lds_cap) {
            ce = cb + 1;
            while (ce < FR_GTILE && cstart[ce + 1] - cstart[cb] <= a.lds_cap) ce++;
        }
        const int s0 = cstart[cb], ns = cstart[ce] - s0;
        if (ns > a.lds_cap) { if (tid == 0) atomicMax(a.flags, 300003); cb = ce; continue; }      // one cell beyond the stage
        for (int p = tid; p < ns; p += FR_THREADS) {
            int lo = cb, hi = ce;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (cstart[mid] <= s0 + p) lo = mid; else hi = mid;
            }
            const int j = src0[lo] + (s0 + p - cstart[lo]);
            const int jo = VIA_PERM ? a.perm[j] : j;
            const AtomSoA &from = VIA_PERM ? a.src : a.dst;
            const int fl = near_flags(from.x[0][jo], from.x[1][jo], from.x[2][jo], a.sl.lo, a.sl.hi);
            const bool em = fl && in_dir(fl, sdir[lo]);
            fr_em[p] = em ? 1 : 0;
            if (em) atomicAdd(&gl[lo], 1);        // (LDS)
        }
        __syncthreads();
        cb = ce;
    }
    // ghosts per cell -> first slot of each cell
    if (tid < 64) {
        const int c = tid < FR_GTILE ? gl[tid] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (tid >= o) incl += u;
        }
        if (tid < FR_GTILE) {
            a.gstart[code0 + tid] = base + incl - c; gl[tid] = incl - c;
            if (a.gcnt_out) a.gcnt_out[code0 + tid] = c;
        }
        if (tid == FR_GTILE - 1) gl[FR_GTILE] = incl;
    }
    __syncthreads();
    const int total = gl[FR_GTILE];
    if (total == 0) return;
    // second walk over the candidates: the ghosts, in (cell, source index) order - deterministic, no atomics
    cb = 0;
    while (cb < FR_GTILE) {
        int ce = FR_GTILE;
        if (cstart[FR_GTILE] - cstart[cb] > a.lds_cap) {
            ce = cb + 1;
            while (ce < FR_GTILE && cstart[ce + 1] - cstart[cb] <= a.lds_cap) ce++;
        }
        const int s0 = cstart[cb], ns = cstart[ce] - s0;
        if (ns > a.lds_cap) { cb = ce; continue; }
        const bool refill = !(cb == 0 && ce == FR_GTILE);      // several passes: the flags of this pass again
        if (refill) {
            __syncthreads();
            for (int p = tid; p < ns; p += FR_THREADS) {
                int lo = cb, hi = ce;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (cstart[mid] <= s0 + p) lo = mid; else hi = mid;
                }
                const int j = src0[lo] + (s0 + p - cstart[lo]);
                const int jo = VIA_PERM ? a.perm[j] : j;
                const AtomSoA &from = VIA_PERM ? a.src : a.dst;
                const int fl = near_flags(from.x[0][jo], from.x[1][jo], from.x[2][jo], a.sl.lo, a.sl.hi);
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
            const int jo = VIA_PERM ? a.perm[j] : j;
            const AtomSoA &from = VIA_PERM ? a.src : a.dst;
            const double xj = from.x[0][jo], yj = from.x[1][jo], zj = from.x[2][jo];
            const double gx = xj + a.sh.s[d][0], gy = yj + a.sh.s[d][1], gz = zj + a.sh.s[d][2];
            a.dst.x[0][gi] = gx; a.dst.x[1][gi] = gy; a.dst.x[2][gi] = gz;
            const int tg = from.tag[jo], ty = from.type[jo];
            a.dst.tag[gi] = tg; a.dst.type[gi] = ty; a.dst.mask[gi] = from.mask[jo];
            // pack_comm_vel + gpu_merge_xvt for the ghost (k_pack_forward's expressions: same bits)
            float4 c, v;
            c.x = (float)(gx - a.ce.c[d][0]); c.y = (float)(gy - a.ce.c[d][1]); c.z = (float)(gz - a.ce.c[d][2]);
            c.w = __uint_as_float((u32)(ty - 1));
            v.x = (float)from.v[0][jo]; v.y = (float)from.v[1][jo]; v.z = (float)from.v[2][jo];
            v.w = __uint_as_float(signature(a.mg.seed, tg, v.x, v.y, v.z));
            a.mg.coord4[gi] = c;
            a.mg.veloc4[gi] = v;
            a.sendlist[k] = j;
            a.senddir[k] = (unsigned char)d;
            if (a.img_cnt) {
                // where the images of atom j live (the step-boundary epilogue refreshes them between rebuilds).  The atom's slot for
                // this image = the number of its images in lower directions - no atomic, no round trip; the image in its lowest
                // direction also books their number
                const int sc = scell[lo];
                const u32 emask = image_mask(near_flags(xj, yj, zj, a.sl.lo, a.sl.hi), sc & 1023, (sc >> 10) & 1023, sc >> 20, a.g.mbin, a.dir_mask);
                const int slot = __popc(emask & ((1u << d) - 1u)), nimg = __popc(emask);
                if (slot < 8) a.img[(size_t)j * 8 + slot] = gi | (d << 26);
                if (slot == 0) a.img_cnt[j] = nimg;
            }
        }
        cb = ce;
    }
}

__global__ void __launch_bounds__(FR_THREADS) k_fr_ghosts(FusedArgs a) { fr_ghosts_tile<false>(a, (int)blockIdx.x); }

// gather + ghosts as ONE launch (one rank, order-only placing kernel, images booked by the count): the first nghost_blocks workgroups
// are ghost tiles - latency chains that need nothing of the gather, only the permutation and the cell starts of the ordering kernel
// and the ghost totals of the count - the rest stream the payload
template <bool LISTS>
__global__ void __launch_bounds__(FR_THREADS) k_fr_gather_ghosts(FusedArgs a, int nghost_blocks)
{
    static_assert(FR_THREADS == 256, "one workgroup size for both halves");
    if ((int)blockIdx.x < nghost_blocks) fr_ghosts_tile<true>(a, (int)blockIdx.x);
    else fr_gather_block<LISTS>(a, (int)blockIdx.x - nghost_blocks);
}

void launch_fused_rebuild(const FusedArgs &a, hipStream_t s, bool counted)
{
    if (a.n <= 0) return;
    if (!counted) hipLaunchKernelGGL(k_fr_count, dim3((a.n + 255) / 256), dim3(256), 0, s, fused_count_args(a), a.skip, a.skip_n, a.n);
    const int ntl = 2 * a.M / FR_TILE, ntg = a.M / FR_GTILE;
    if (a.stot) hipLaunchKernelGGL(k_fr_super, dim3((ntl + FR_SUPER - 1) / FR_SUPER), dim3(FR_THREADS), 0, s, a.ttot, ntl, a.stot);
    const size_t dyn2 = (size_t)a.lds_cap * 8, dyn3 = (size_t)a.lds_cap;
    const bool lists = a.src.bpa > 0 || a.src.apa > 0 || a.src.msp > 0;
    if (dyn2 > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)k_fr_place<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
        (void)hipFuncSetAttribute((const void *)k_fr_place<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
        (void)hipFuncSetAttribute((const void *)k_fr_place<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);
    }
    // large boxes of one rank: order, then a streaming gather (split_gather: the engine's choice; needs the permutation array and
    // no holes in the old order)
    const bool split = a.split_gather && a.perm && !a.skip;
    if (lists && !split) hipLaunchKernelGGL((k_fr_place<true, true>), dim3(2 * a.M / FR_TILE), dim3(FR_PLACE_THREADS), dyn2, s, a);
    else if (split) {
        // (merged_ghosts, the engine's wish, holds only with the order-only placing kernel and images booked by the count)
        FusedArgs b = a;
        b.merged_ghosts = (a.merged_ghosts && a.gttot && a.img_booked) ? 1 : 0;
        hipLaunchKernelGGL((k_fr_place<false, false>), dim3(2 * a.M / FR_TILE), dim3(FR_PLACE_THREADS), dyn2, s, b);
        if (b.merged_ghosts) {
            if (a.gstot) hipLaunchKernelGGL(k_fr_super, dim3((ntg + FR_SUPER - 1) / FR_SUPER), dim3(FR_THREADS), 0, s, a.gttot, ntg, a.gstot);
            const int ngb = a.gorder ? a.ngorder : ntg;
            if (lists) hipLaunchKernelGGL(k_fr_gather_ghosts<true>, dim3(ngb + (a.n + 255) / 256), dim3(256), dyn3, s, b, ngb);
            else hipLaunchKernelGGL(k_fr_gather_ghosts<false>, dim3(ngb + (a.n + 255) / 256), dim3(256), dyn3, s, b, ngb);
            return;
        }
        if (lists) hipLaunchKernelGGL(k_fr_gather<true>, dim3((a.n + 255) / 256), dim3(256), 0, s, b);
        else hipLaunchKernelGGL(k_fr_gather<false>, dim3((a.n + 255) / 256), dim3(256), 0, s, b);
    } else hipLaunchKernelGGL((k_fr_place<false, true>), dim3(2 * a.M / FR_TILE), dim3(FR_PLACE_THREADS), dyn2, s, a);
    if (a.gttot && a.gstot) hipLaunchKernelGGL(k_fr_super, dim3((ntg + FR_SUPER - 1) / FR_SUPER), dim3(FR_THREADS), 0, s, a.gttot, ntg, a.gstot);
    if (a.gttot) hipLaunchKernelGGL(k_fr_ghosts, dim3(a.gorder ? a.ngorder : ntg), dim3(FR_THREADS), dyn3, s, a);
    else if (!a.novf_later) (void)hipMemsetAsync(a.novf, 0, sizeof(int), s);      // (the ghost kernel clears the overflow count otherwise;
                                                                                  // novf_later: the caller's next kernel does)
}

int fused_direct_tiles() { return FR_DIRECT_TILES; }
int fused_tile_codes() { return FR_TILE; }
int fused_gtile_codes() { return FR_GTILE; }
int fused_super_tiles() { return FR_SUPER; }

} // namespace meso
