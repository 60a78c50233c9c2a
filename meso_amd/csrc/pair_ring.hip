// pair_style dpd/fast/meso force kernel, "ring" form (gfx950, wave64).
//
// Computes what gpu_dpd_fast<0> computes (/root/reference/src/USER-MESO/pair_dpd_fast_meso.cu:91-205): full list,
// newton off, fp32 arithmetic, TEA-keyed pair noise.  The lane-per-atom kernels are bound by VALU issue, not by HBM
// (profiles/r01_pmc_64_fast.txt: 77 M wave-instructions per launch = 87 % of the VALU issue slots of 1024 SIMDs):
// only ~45 % of the row entries are inside r_c and a wave executes the ~110-instruction TEA/Gaussian/weight
// sequence for a slot as soon as ONE of its lanes has a hit, for as many slots as its longest row has.  Here
//
//   light phase (lane = atom): the row is read 8 entries at a time (two 16-byte words of the chunked-8 table, the
//     next chunk's words already in flight), the 8 partner coordinates are gathered with buffer loads (a 32-bit
//     byte offset per lane, no 64-bit address arithmetic), the cutoff is tested, and each hit appends ONE 4-byte
//     record (owner lane << 26 | partner index) to a per-wave LDS ring with ballot + mbcnt;
//   heavy phase (lane = hit): whenever 64 records are queued their partner coordinate/velocity words are
//     requested (64 independent gathers), and the batch requested at the PREVIOUS drain point - whose data has
//     arrived meanwhile - is evaluated with every lane busy.  The owner's own coordinate/velocity come from LDS;
//     the three force components are added to per-wave LDS accumulators as 64-bit fixed point (2^-32 units,
//     ds_add_u64).  Measured on gfx950 (tools/micro/lds_atomic_bench.hip), CU cycles per 64-lane x 3-component
//     group: ds_add_f32 579, ds_add_f64 56, ds_add_u64 30, ds_add_u32 21, plain 12-byte stores 21 - the float
//     LDS atomics serialise, the integer ones run at store speed.
//
// Newton pairing inside a workgroup (SHARE): the 256 atoms of a workgroup are an aligned group of the cell-ordered
// storage order, and 64 % of the in-range pairs (56 % of the row entries) have both atoms in one such group
// (tools/measure_sharing.py).  For those the entry with the lower index is evaluated once and added to BOTH atoms
// (the accumulators are workgroup-shared; integer sums commute, so waves may add in any order), the mirrored entry
// (partner index < own index, same group) is neither gathered nor tested.  The decision needs only the two indices, so
// rows stay complete full-list rows for every other kernel.
//
// One wave owns its 64 atoms from start to end: no block barriers after the prologue, no global atomics; integer
// addition is associative, so the sums do not depend on the order in which hits are drained (bit-reproducible).
// Nothing is left to the compiler's contraction: the fused multiply-adds of the fp32 style are written out (see the kernel).
//
// Partitioned rows (LP = 1, round 5; RowPartArgs in kernels.h): the list builder has decided once per rebuild which of an atom's
// in-group pairs it evaluates (a balanced rule: about half of them, whatever the atom's place in the group) and has moved the
// mirrored entries behind the row's front section.  The light phase then walks nact[i] entries instead of count[i] (26 instead of
// 36 on average: five row chunks per wave instead of seven), every in-group entry it meets is one it evaluates for both atoms, and
// the four waves of a group carry the same share of the pairs (with "the lower index evaluates" the first wave of a group had 2.4
// times the hits of the last, and the group's accumulators are read only when the slowest wave is through).  Forces are
// bit-identical: the same pairs, each evaluated once from one side or once from each, summed as integers.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "kernels.h"
#include "meso_device.h"

namespace meso {

#ifndef RG_UNIT
#define RG_UNIT 1             // 1: the fp32 instantiations and the launcher; 2 (pair_ring_dp.hip): the fp64 instantiations
#endif
#ifndef RG_WAVES
#define RG_WAVES 4
#endif
#define RG_GROUP (64 * RG_WAVES)      // atoms of a workgroup = Newton-pairing group
#ifndef RG_DP_PREFILTER
#define RG_DP_PREFILTER 1           // fp64 style: fp32 cutoff filter in the row walk, exact fp64 test where the pair is evaluated
#endif
#ifndef RG_FIX_WAVES
#define RG_FIX_WAVES 0
#endif
#ifndef RG_FIX32
#define RG_FIX32 1                  // fp32 style: force sums as 32-bit fixed point (to_fixed16, meso_device.h); 0: the 64-bit sums of rounds 1-4
#endif
#ifndef RG_RING
#define RG_RING 256                 // records per wave; a drain check every 2 slots keeps the fill below 64 + 128
#endif
#ifndef RG_OCC
#define RG_OCC 20                   // waves per CU the fp32 kernel is compiled for
#endif
#ifndef RG_OCC_DP
#define RG_OCC_DP 8                 // ... the fp64 style (116 VGPRs: four waves per SIMD)
#endif
#ifndef RG_OCC_PARTS
#define RG_OCC_PARTS 20             // ... the variants with 2 / 4 lanes per atom (small launches: one round of waves; they prefetch the step boundary's inputs)
#endif
#ifndef RG_EXEC_GATHER
#define RG_EXEC_GATHER 0      // measured: lanes switched off for the gather 103 -> 109 us (out-of-range lanes are cheap already)
#endif
#define RG_OWNER_SHIFT 26
#define RG_SHARED_BIT 0x02000000u   // record: evaluate once, add to owner and partner
#define RG_INDEX_MASK 0x01FFFFFFu

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

__device__ inline float4 buf_load4(__amdgpu_buffer_rsrc_t r, u32 byte_off)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

// EW1: every pair type has weight exponent s = 1 (the usual DPD choice): w_R = w_C, no pow()
// FAST: dpd/fast/meso (fp32 arithmetic, contracted); otherwise dpd/meso (fp64 arithmetic on the fp32 operands through the
// uncontracted functions of meso_device.h, 36-fractional-bit fixed-point sums)
// TY: 0 one atom type (coefficients are kernel-argument constants), 1 several types with ONE cutoff (the cutoff test stays scalar),
// 2 several types, cutoff per type pair
// NPART_: lanes per atom (1, 2, 4); 0 = one lane per atom with WIDE records for more than 2^25 atoms on a rank: the record word
// is the whole 32-bit partner index, owner lane and pairing flag travel in a byte ring next to it
// PLAIN: the noise is the TEA-keyed Gaussian and the conservative force a0 w (dpd/meso, dpd/fast/meso): the wave-uniform
// switches for dpd/mini, dpd/polyforce and dpd/tableforce (PairArgs::rng / poly / ftab) are compiled out of the hot loop
#ifdef RG_STAMP
__device__ unsigned long long *g_stamp_dev = nullptr;
#endif
// LP: 0 plain rows - the row is walked over its whole extent and the pairing class of an entry comes from the two indices ("the lower
// index evaluates"); 1 partitioned rows (RowPartArgs, kernels.h; SHARE only): the walk covers the front section; 2 partitioned rows
// under a launch that cannot pair by the builder's rule (a range that splits a group): the walk of LP 0 over front and back section
template <bool FAST, int TY, bool EW1, bool SHARE, int NPART_, bool PLAIN, int LP>
__global__ void __launch_bounds__(64 * RG_WAVES, FAST ? ((NPART_ == 1 ? RG_OCC : RG_OCC_PARTS) / RG_WAVES > 0 ? (NPART_ == 1 ? RG_OCC : RG_OCC_PARTS) / RG_WAVES : 1) : (RG_OCC_DP / RG_WAVES > 0 ? RG_OCC_DP / RG_WAVES : 1))
#if RG_FIX_WAVES
    __attribute__((amdgpu_waves_per_eu(RG_FIX_WAVES, RG_FIX_WAVES)))      // (the register allocator otherwise aims for a wave more than the launch bounds ask and spills for it)
#endif
    k_pair_dpd_ring(PairArgs a)
{
    prefetch_kernargs<sizeof(PairArgs)>();
    // (requested now, looked at in front of the epilogue's stores: no wait of its own)
    const int poisoned = a.poison ? *a.poison : 0;
    // (no contraction left to the compiler: the pair evaluation is inlined at every drain point of the light phase, and copies that
    // fuse different multiply-adds would give one pair two forces that differ in the last bit, depending on which copy - and, for a
    // pair evaluated from both sides, which side - got it.  The fused operations of the fp32 style are written out below.)
#ifdef RG_STAMP
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ double smem[];
    float *cf32 = (float *)smem;
    double *cf64 = smem;
    constexpr bool NT1 = TY == 0, UCUT = TY <= 1;
    constexpr bool WIDE = NPART_ == 0;
    constexpr int NPART = WIDE ? 1 : NPART_;
    constexpr bool PARTED = LP == 1;
    static_assert(!PARTED || SHARE, "partitioned rows serve pairing launches");
    // fp64 style, one lane per atom (launches of several rounds of waves): fp32 cutoff filter in the row walk, exact fp64 test where
    // the pair is evaluated (64^3: 138.5 -> 133.3 us, identical forces; the two-lane form of small boxes lost 1 % with it)
    constexpr bool DPF = RG_DP_PREFILTER && !FAST && NPART_ == 1;
    constexpr int RING = RG_RING;
    // fp32 style: rows of 8 floats (a0, gamma, sigma, s | 1/rc, rc^2, rc, -): one 16-byte LDS read per evaluated pair
    constexpr int CFP = FAST ? 8 : N_COEFF;
    const int ncf = NT1 ? 0 : a.ntypes * a.ntypes * CFP;
    for (int p = threadIdx.x; p < ncf; p += blockDim.x) {
        if (FAST) {
            const int src[8] = {P_A0, P_GAMMA, P_SIGMA, P_EXPW, P_CUTINV, P_CUTSQ, P_CUT, P_CUT};
            cf32[p] = a.coeff32[(p >> 3) * N_COEFF + src[p & 7]];
        } else cf64[p] = a.coeff64[p];
    }
    const size_t off = ((size_t)ncf * (FAST ? 4 : 8) + 15) & ~(size_t)15;
    // (the wave's number in an SGPR: the addresses of its LDS areas are scalar then)
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const size_t per_wave = RING * 16 + (NT1 ? 0 : RING) + (WIDE ? RING : 0);
    // [3][256] force sums of the workgroup's atoms in fixed point: 64 bits (36 fractional) in the fp64 style, 32 bits (16) in the fp32 one
    constexpr bool F32 = FAST && RG_FIX32;
    typedef typename std::conditional<F32, u32, u64>::type acc_t;
    acc_t *facc = (acc_t *)((char *)smem + off);
    // the workgroup's atoms, coordinate and velocity records, by group-local index (wave w owns [64 w, 64 w + 64): one lane per atom):
    // own_c / own_v are this wave's part; a partner of another wave of the group is looked up in own_v_all (issue())
    float4 *own_c_all = (float4 *)((char *)smem + off + 3 * 64 * RG_WAVES * sizeof(acc_t));      // (accumulator area sized for NPART = 1)
    float4 *own_v_all = own_c_all + 64 * RG_WAVES;
    constexpr int APW = 64 / NPART;                          // atoms per wave (NPART lanes share one atom, see below)
    float4 *own_c = own_c_all + APW * w, *own_v = own_v_all + APW * w;
    char *wb = (char *)(own_v_all + 64 * RG_WAVES) + (size_t)w * per_wave;
    float4 *ring = (float4 *)wb;        // (partner x, y, z, record word): the coordinate is not gathered twice
    unsigned char *ringt = (unsigned char *)(ring + RING);        // several types: the partner's type next to its record
    unsigned char *ringm = ringt + (NT1 ? 0 : RING);              // WIDE: owner lane | pairing flag << 6

    // XCD-aware order (workgroups b and b + 8 share an L2): every XCD walks a contiguous range of the atoms - with a border section
    // (xcd_sb > 0) a contiguous range of the bulk workgroups and then one of the border workgroups: in Morton order the eighths of
    // both sections are the octants of the box, so an XCD's border atoms lie next to its bulk atoms, and the border atoms' extra work
    // (a fifth more pairs to evaluate, their periodic images) is shared by the eight XCDs instead of ending the launch on the last
    const int nbk = gridDim.x;
    int blk;
    if (a.xcd_sb > 0) {
        const int x = (int)(blockIdx.x & 7), r = (int)(blockIdx.x >> 3);
        blk = r < a.xcd_sb ? (x * a.xcd_sb + r < a.xcd_kb ? x * a.xcd_sb + r : 0x3FFFFF) : a.xcd_kb + x * a.xcd_sr + (r - a.xcd_sb);
    } else blk = (nbk & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3));
    // NPART lanes share one atom (small launches: more waves for the same atoms): lane = part * APW + slot, the parts of an
    // atom walk its row chunks part, part + NPART, ...; sums meet in the LDS accumulators like those of the other waves
    constexpr int NB = APW * RG_WAVES;                       // atoms per workgroup = Newton-pairing group of this launch
    const int slot = lane % APW, part = lane / APW;
    const int blockbase = a.beg + blk * NB;                  // SHARE: beg is a multiple of 256 (launcher)
    const int i = blockbase + w * APW + slot;
    const bool mine = i < a.end;
    float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = c1;
    int n = 0;
    // the first chunk of the row is requested together with the atom's own data (its address needs the index only; a row past
    // its count holds stale entries that are never looked at): one memory round trip less at the head of every wave
    const int4 *rows = (const int4 *)a.table + 2 * row_word8(mine ? i : a.beg, 0, a.n_col);      // (the front section, or the whole plain row)
    int4 first0 = make_int4(0, 0, 0, 0), first1 = first0;
    if (mine) {
        c1 = a.coord4[i]; v1 = a.veloc4[i]; n = a.count[i];
        first0 = rows[(size_t)part * 128]; first1 = rows[(size_t)part * 128 + 1];
    }
    // small launches (two or four lanes per atom: every wave's latency chain counts): what the step-boundary epilogue needs is
    // requested now and waits in registers (16 VGPRs: only the variants with registers to spare)
#ifndef RG_LDS_VELOC
#define RG_LDS_VELOC 1        // in-group partners' velocity records from the workgroup's LDS copy (one-lane variant only, see issue())
#endif
#ifndef RG_PRE_ALL
#define RG_PRE_ALL 0          // measured: the prefetch in the one-lane variant too (93 VGPRs): 64^3 unchanged, 48^3 -2.3 %
#endif
    constexpr bool PRE = NPART > 1 || (RG_PRE_ALL && FAST && TY == 0);
    NvePre npre;
    if (PRE && a.fuse_nve && mine && part == 0) nve_prefetch(a.nve, i, npre, (int)__float_as_uint(c1.w) + 1);
    const int ob = w * APW + slot;                           // my atom's slot in the workgroup's accumulators
    if (part == 0) {
        own_c[slot] = c1;
        own_v[slot] = v1;
        facc[ob] = 0; facc[NB + ob] = 0; facc[2 * NB + ob] = 0;
    }
    __syncthreads();   // coefficient table (multi-type), accumulators, this wave's own_c/own_v
#ifdef RG_STAMP
    const unsigned long long st_prolog = __builtin_amdgcn_s_memtime();
#endif

    // (num_records is 32 bits: launch_pair clamps nall below 2^28 atoms, 16 bytes each)
    const u32 nrec = (u32)min((unsigned long long)(u32)a.nall * 16ull, 0xFFFFFFFFull);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)a.coord4, 0, (int)nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void *)a.veloc4, 0, (int)nrec, 0x00020000);
    const u32 t1 = __float_as_uint(c1.w);
    const float dtis = (float)a.dt_inv_sqrt;
    const u32 lanehi = (u32)slot << RG_OWNER_SHIFT;
    const bool blk_full = blockbase + NB <= a.end;
    int nch, nchmax;
    // chunks part, part + NPART, ... of a row of n entries; the wave walks as many as its longest row has
    auto set_row = [&](int n_) {
        nch = (((n_ + 7) >> 3) - part + NPART - 1) / NPART;
        nchmax = nch;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nchmax = max(nchmax, __shfl_xor(nchmax, o, 64));
        nchmax = __builtin_amdgcn_readfirstlane(nchmax);
    };
    set_row(n);

    int qhead = 0, qtail = 0;     // wave-uniform
    int pn = 0;                   // records of the batch whose gathers are in flight
    u32 pe = 0, pm = 0;
    float4 pc2 = make_float4(0.f, 0.f, 0.f, 0.f), pv2 = pc2;

    // evaluate the pending batch (lane = hit)
#ifdef RG_STAMP
    // timing build (tools/build_variant.sh ... -DRG_STAMP): shader-clock cycles of one wave per phase, summed over the waves
    unsigned long long st_heavy = 0, st_issue = 0, st_cand = 0, st_push = 0, st_ncand = 0;
#define ST_BEGIN() const unsigned long long st_t0 = __builtin_amdgcn_s_memtime()
#define ST_END(acc) acc += __builtin_amdgcn_s_memtime() - st_t0
#else
#define ST_BEGIN()
#define ST_END(acc)
#endif
    auto compute = [&]() __attribute__((always_inline)) {
        if (pn > 0) {
            ST_BEGIN();
            if (lane < pn) {
                const u32 owner = WIDE ? (pm & 63u) : pe >> RG_OWNER_SHIFT;
                const float4 ci = own_c[owner], vi = own_v[owner];
                if (RG_LDS_VELOC && (NPART_ == 1 && FAST) && !WIDE && a.lds_veloc) {
                    // in-group partner: its velocity record is the LDS copy of the wave that owns it (see issue())
                    const bool inwg = SHARE && (WIDE ? (pm & 64u) : (pe & RG_SHARED_BIT)) != 0;
                    const u32 pl = (WIDE ? pe : (pe & RG_INDEX_MASK)) - (u32)blockbase;                // (in-group: < NB)
                    const float4 vl = own_v_all[inwg ? pl : (u32)(APW * w + slot)];
                    pv2.x = inwg ? vl.x : pv2.x; pv2.y = inwg ? vl.y : pv2.y; pv2.z = inwg ? vl.z : pv2.z; pv2.w = inwg ? vl.w : pv2.w;      // (component-wise: v_cndmask, not a trip through scratch)
                }
                acc_t qx, qy, qz;
                if (FAST) {
                    float c_cutinv, c_ew, c_a0, c_gamma, c_sigma;
                    if (NT1) {
                        c_cutinv = (float)a.cf1[P_CUTINV]; c_ew = (float)a.cf1[P_EXPW]; c_a0 = (float)a.cf1[P_A0];
                        c_gamma = (float)a.cf1[P_GAMMA]; c_sigma = (float)a.cf1[P_SIGMA];
                    } else {
                        const float *cf = cf32 + (__float_as_uint(ci.w) * a.ntypes + __float_as_uint(pc2.w)) * 8;
                        const float4 cq = *(const float4 *)cf;
                        c_a0 = cq.x; c_gamma = cq.y; c_sigma = cq.z; c_ew = cq.w;
                        c_cutinv = UCUT ? (float)a.cf1[P_CUTINV] : cf[4];
                    }
                    const float dx = ci.x - pc2.x, dy = ci.y - pc2.y, dz = ci.z - pc2.z;
                    const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    const float rn = PLAIN ? gaussian_tea_fast(__float_as_uint(vi.w), __float_as_uint(pv2.w))
                                           : pair_noise_fast(a.rng, __float_as_uint(vi.w), __float_as_uint(pv2.w));
                    const float rinv = __builtin_amdgcn_rsqf(rsq);
                    const float r = rsq * rinv;
                    const float dvx = vi.x - pv2.x, dvy = vi.y - pv2.y, dvz = vi.z - pv2.z;
                    const float dot = __builtin_fmaf(dz, dvz, __builtin_fmaf(dy, dvy, dx * dvx));
                    const float wc = __builtin_fmaf(-r, c_cutinv, 1.0f);
                    float wr = wc;
                    if (!EW1 && c_ew != 1.0f) wr = __builtin_amdgcn_exp2f(c_ew * __builtin_amdgcn_logf(wc));   // powf(wc, s), wc in (0,1)
                    float fcons = c_a0 * wc;
                    if (!PLAIN && a.poly)      // dpd/polyforce/meso (gpu_dpd_polyforce pair_dpd_polyforce_meso.cu:159-162)
                        fcons = polyval_f32(wc, a.poly + (NT1 ? 0 : (__float_as_uint(ci.w) * a.ntypes + __float_as_uint(pc2.w)) * MESO_POLY_PITCH));
                    if (!PLAIN && a.ftab)      // dpd/tableforce/meso (gpu_dpd_tableforce pair_dpd_tableforce_meso.cu:181)
                        fcons = table_force_f32(r * c_cutinv, a.ftab + (NT1 ? 0 : (__float_as_uint(ci.w) * a.ntypes + __float_as_uint(pc2.w)) * a.ftab_len), a.ftab_len);
                    float fpair = __builtin_fmaf(c_sigma * wr * rn, dtis, fcons - (c_gamma * wr * wr * dot * rinv));
                    fpair *= rinv;
                    if constexpr (F32) { qx = to_fixed16(dx * fpair); qy = to_fixed16(dy * fpair); qz = to_fixed16(dz * fpair); }
                    else { qx = to_fixed(dx * fpair); qy = to_fixed(dy * fpair); qz = to_fixed(dz * fpair); }
                } else {
                    PairCoeff64 pc;
                    if (NT1) {
                        pc.cutinv = a.cf1[P_CUTINV]; pc.expw = a.cf1[P_EXPW]; pc.a0 = a.cf1[P_A0]; pc.gamma = a.cf1[P_GAMMA]; pc.sigma = a.cf1[P_SIGMA];
                    } else {
                        const double *cf = cf64 + (__float_as_uint(ci.w) * a.ntypes + __float_as_uint(pc2.w)) * N_COEFF;
                        pc.cutinv = cf[P_CUTINV]; pc.expw = cf[P_EXPW]; pc.a0 = cf[P_A0]; pc.gamma = cf[P_GAMMA]; pc.sigma = cf[P_SIGMA];
                    }
                    double fx, fy, fz;
                    const double cutsq_x = !DPF ? -1.0 : UCUT ? a.cf1[P_CUTSQ]
                                         : cf64[(__float_as_uint(ci.w) * a.ntypes + __float_as_uint(pc2.w)) * N_COEFF + P_CUTSQ];
                    pair_dpd_f64<EW1>(ci, pc2, vi, pv2, pc, a.dt_inv_sqrt, fx, fy, fz, cutsq_x);
                    qx = to_fixed36(fx); qy = to_fixed36(fy); qz = to_fixed36(fz);
                }
                const u32 oo = (u32)(w * APW) + owner;
                __hip_atomic_fetch_add(&facc[oo], qx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&facc[NB + oo], qy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&facc[2 * NB + oo], qz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (SHARE && (WIDE ? (pm & 64u) : (pe & RG_SHARED_BIT))) {
                    // the partner is one of this workgroup's atoms: it receives the opposite force now and skips its own
                    // (mirrored) row entry
                    const u32 pj = (WIDE ? pe : (pe & RG_INDEX_MASK)) - (u32)blockbase;
                    __hip_atomic_fetch_sub(&facc[pj], qx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_sub(&facc[NB + pj], qy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_sub(&facc[2 * NB + pj], qz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            pn = 0;
            ST_END(st_heavy);
        }
    };
    // request the partner words of the next nb queued records
    auto issue = [&](int nb) __attribute__((always_inline)) {
        ST_BEGIN();
        if (lane < nb) {
            const float4 rec = ring[(qhead + lane) & (RING - 1)];
            pe = __float_as_uint(rec.w);
            pc2 = make_float4(rec.x, rec.y, rec.z, 0.f);
            if (WIDE) pm = ringm[(qhead + lane) & (RING - 1)];
            const u32 joff = (WIDE ? pe : (pe & RG_INDEX_MASK)) << 4;
            if (!NT1) pc2.w = __uint_as_float((u32)ringt[(qhead + lane) & (RING - 1)]);     // (a fourth gather per hit before)
            if (RG_LDS_VELOC && (NPART_ == 1 && FAST) && !WIDE && a.lds_veloc) {
            // a partner of this workgroup's group (the pairs evaluated once for both, 64 % of the hits) has its velocity record in
            // the LDS copy of the wave that owns it: those lanes read it there and give the gather an out-of-range offset (one lane
            // per atom, launches of several rounds of waves: 64^3 fused launch 121.9 -> 119.7 us, +1.2 % steps/s; 48^3 44.7 -> 46.6 us
            // alone and two lanes per atom 19.0 -> 20.4 us, so not there: PairArgs::lds_veloc, set by the launcher from the size;
            // the fp64 style lost 4 % with it at 64^3: fp32 kernels only)
            // (the LDS record is read - and chosen - when the batch is evaluated: a select here would wait for the gather at once)
            const bool inwg = SHARE && (WIDE ? (pm & 64u) : (pe & RG_SHARED_BIT)) != 0;
            pv2 = buf_load4(rv, inwg ? 0xFFFFFFF0u : joff);
            } else pv2 = buf_load4(rv, joff);
        }
        pn = nb;
        qhead += nb;
        ST_END(st_issue);
    };

    // chunk ch of my row: two 16-byte words (lanes past their row: zeros, never used)
    auto ldrow = [&](int ch, int4 &w0, int4 &w1) {
        w0 = make_int4(0, 0, 0, 0); w1 = w0;
        if (ch == 0) { if (nch > 0) { w0 = first0; w1 = first1; } }
        else if (ch < nch) { w0 = rows[(size_t)(ch * NPART + part) * 128]; w1 = rows[(size_t)(ch * NPART + part) * 128 + 1]; }
    };
    // the entries of one chunk: which of them are looked at, and their coordinate gathers on the way
    auto prep = [&](int ch, const int4 w0, const int4 w1, int (&j)[8], bool (&use)[8], bool (&shb)[8], float4 (&c2)[8]) {
        const bool active = ch < nch;
        j[0] = w0.x; j[1] = w0.y; j[2] = w0.z; j[3] = w0.w; j[4] = w1.x; j[5] = w1.y; j[6] = w1.z; j[7] = w1.w;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            use[q] = active;
            shb[q] = false;
            if (SHARE) {
                // same aligned group: lower partner index = mirrored entry, not looked at; higher (and one of this launch's atoms) =
                // evaluated once for both.  Two compares per entry; everything else is scalar logic on their lane masks
                // (lower-or-equal: the tail slots of a row hold the atom itself - not looked at either)
                if (PARTED) {
                    // (front section: every in-group entry is mine to evaluate for both; the section's tail slots hold the atom itself)
                    shb[q] = ((u32)j[q] ^ (u32)i) < (u32)NB;
                } else {
                    const bool same = ((u32)j[q] ^ (u32)i) < (u32)NB, lower = (u32)j[q] <= (u32)i;
                    use[q] = active & !(same & lower);
                    shb[q] = same & !lower;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; q++) c2[q] = buf_load4(rc, use[q] ? ((u32)j[q] << 4) : 0xFFFFFFF0u);   // out of range: returns 0, no fetch
    };
    // cutoff test of one chunk's entries, hits into the ring, drain checks
    auto proc = [&](const int (&j)[8], const bool (&use)[8], const bool (&shb)[8], const float4 (&c2)[8]) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            // tail slots hold i itself: rsq = 0.  One set of compares: the hit is a lane mask already, its ballot costs nothing
            bool hit;
            if (FAST) {
                const float dx = c1.x - c2[q].x, dy = c1.y - c2[q].y, dz = c1.z - c2[q].z;
                const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                const float cutsq = UCUT ? (float)a.cf1[P_CUTSQ] : cf32[(t1 * a.ntypes + __float_as_uint(c2[q].w)) * 8 + 5];
                hit = (rsq < cutsq) & (rsq >= (float)MESO_EPSILON_SQ) & use[q];
            } else if (DPF) {
                // fp64 style: the row walk filters in fp32 against a cutoff widened by 1e-5 (fp32 r^2 is within 3e-7 of the fp64 one);
                // the exact fp64 test - the reference's - is made where the pair is evaluated (pair_dpd_f64 returns zero for the few
                // pairs the widening lets through): the same pairs contribute the same forces, at a third of the light phase's fp64 work
                const float dx = c1.x - c2[q].x, dy = c1.y - c2[q].y, dz = c1.z - c2[q].z;
                const float rsq = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                const float cutsq = 1.00001f * (float)(UCUT ? a.cf1[P_CUTSQ] : cf64[(t1 * a.ntypes + __float_as_uint(c2[q].w)) * N_COEFF + P_CUTSQ]);
                hit = (rsq < cutsq) & (rsq >= 0.5f * (float)MESO_EPSILON_SQ) & use[q];
            } else {
                const double rsq = rsq_f64(c1, c2[q]);
                const double cutsq = UCUT ? a.cf1[P_CUTSQ] : cf64[(t1 * a.ntypes + __float_as_uint(c2[q].w)) * N_COEFF + P_CUTSQ];
                hit = (rsq < cutsq) & (rsq >= MESO_EPSILON_SQ) & use[q];
            }
            const u64 m = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const u32 pos = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, (u32)qtail)) & (RG_RING - 1);
                // (partitioned rows: pairs are shared inside groups that lie wholly below the end - the builder's own test, scalar here)
                const bool sh = SHARE && shb[q] && (PARTED ? blk_full : (u32)j[q] < (u32)a.end);
                ring[pos] = make_float4(c2[q].x, c2[q].y, c2[q].z, __uint_as_float(WIDE ? (u32)j[q] : ((u32)j[q] | (sh ? lanehi | RG_SHARED_BIT : lanehi))));   // record word last
                if (WIDE) ringm[pos] = (unsigned char)((u32)slot | (sh ? 64u : 0u));
                if (!NT1) ringt[pos] = (unsigned char)__float_as_uint(c2[q].w);
            }
            qtail += __popcll(m);
            if (RG_RING < 256 || (q & 1)) {
                while (qtail - qhead >= 64) { compute(); issue(64); }
            }
        }
    };
    // a launch that does not pair by the builder's rule on a partitioned table (ranges that split a group, the records of very
    // large systems) needs every neighbour: it walks the back section as a second row
    constexpr int npass = LP == 2 ? 2 : 1;
#pragma unroll 1
    for (int pass = 0; pass < npass; pass++) {
        if (pass == 1) {
            rows = (const int4 *)a.table_back + 2 * row_word8(mine ? i : a.beg, 0, a.nb_col);
            set_row(mine ? a.nback[i] : 0);
            first0 = make_int4(0, 0, 0, 0); first1 = first0;
            if (nch > 0) { first0 = rows[(size_t)part * 128]; first1 = rows[(size_t)part * 128 + 1]; }
        }
        int j[8];
        bool use[8], shb[8];
        float4 c2[8];
        int4 w0, w1;
        ldrow(0, w0, w1);
#pragma unroll 1
        for (int c = 0; c < nchmax; c++) {
            const int4 v0 = w0, v1 = w1;
            ldrow(c + 1, w0, w1);
            prep(c, v0, v1, j, use, shb, c2);
            proc(j, use, shb, c2);
        }
    }
#ifdef RG_STAMP
    const unsigned long long st_light = __builtin_amdgcn_s_memtime();
#endif
    compute();
    while (qtail > qhead) { issue(min(64, qtail - qhead)); compute(); }

#ifdef RG_STAMP
    const unsigned long long st_drained = __builtin_amdgcn_s_memtime();
#endif
#ifndef RG_LATE_PRE
#define RG_LATE_PRE 1         // variants without the early prefetch: the step boundary's inputs are requested in front of the barrier below
#endif
    // (nothing of the heavy phase is live any more: the requests cost no registers at the kernel's peak, and they are in flight while
    // the wave waits for the other waves of its workgroup)
    constexpr bool LATE = RG_LATE_PRE && !PRE;
    if (LATE && a.fuse_nve && mine && part == 0) nve_prefetch(a.nve, i, npre, (int)__float_as_uint(c1.w) + 1);
#ifndef RG_IMG_PRE
#define RG_IMG_PRE 1
#endif
    NveImgPre ipre;
    if (RG_IMG_PRE && a.fuse_nve && mine && part == 0) nve_prefetch_images(a.nve, i, ipre);
    if (SHARE) __syncthreads();      // partners in other waves may still be adding to my sums
    // (PairArgs::poison: an outgrown capacity was reported by this interval's rebuild - nothing is stored, the rebuild will be redone)
    if (poisoned) return;
    double xn0 = 0.0, xn1 = 0.0, xn2 = 0.0;       // the atom's position after the step boundary (for the rebuild's count below)
    if (mine && part == 0) {
        double fx, fy, fz;
        if constexpr (F32) {
            const int sx = (int)facc[ob], sy = (int)facc[NB + ob], sz = (int)facc[2 * NB + ob];
            fx = from_fixed16((u32)sx); fy = from_fixed16((u32)sy); fz = from_fixed16((u32)sz);
            // a sum beyond half the accumulator's range (|F| >= 16384 force units on one atom; a single term beyond the range saturates
            // in to_fixed16 and lands here too): reported, not wrapped silently (meso_hip.h, "units of the fp32 styles")
            if (a.range_flag && max(max(abs(sx), abs(sy)), abs(sz)) > 0x3FFFFFFF) *a.range_flag = 1;
        }
        else if (FAST) { fx = from_fixed(facc[ob]); fy = from_fixed(facc[NB + ob]); fz = from_fixed(facc[2 * NB + ob]); }
        else { fx = from_fixed36(facc[ob]); fy = from_fixed36(facc[NB + ob]); fz = from_fixed36(facc[2 * NB + ob]); }
        if (a.fuse_nve) {
            // final(s) + initial(s+1) (+ merge for s+1 into the other merged buffer: this step's is still being read)
            if (a.accumulate) { fx += a.f[0][i]; fy += a.f[1][i]; fz += a.f[2][i]; }
            if (a.bond.nbond) {
                // bonds of this atom (Bond::compute of the same step): the same uncontracted function the bond kernel calls
                const int nb = a.bond.nbond[i];
                if (nb > 0) {
                    double bx, by, bz, be;
                    const int *bi = a.bond.bond_idx + (size_t)i * a.bond.bpa, *bt = a.bond.bond_type + (size_t)i * a.bond.bpa;
                    if (a.bond.style == 1)
                        bond_forces_of_atom<1, false>(a.coord4, c1, nb, bi, bt, a.bond.cf, a.bond.nbt, a.bond.prd[0], a.bond.prd[1], a.bond.prd[2], bx, by, bz, be);
                    else
                        bond_forces_of_atom<0, false>(a.coord4, c1, nb, bi, bt, a.bond.cf, a.bond.nbt, a.bond.prd[0], a.bond.prd[1], a.bond.prd[2], bx, by, bz, be);
                    fx += bx; fy += by; fz += bz;
                }
            }
            if (PRE || LATE) nve_boundary_atom(a.nve, i, fx, fy, fz, &npre, xn0, xn1, xn2, 0, RG_IMG_PRE ? &ipre : nullptr);
            else nve_boundary_atom(a.nve, i, fx, fy, fz, nullptr, xn0, xn1, xn2, (int)__float_as_uint(c1.w) + 1, RG_IMG_PRE ? &ipre : nullptr);
        } else if (a.accumulate) { a.f[0][i] += fx; a.f[1][i] += fy; a.f[2][i] += fz; }
        else { a.f[0][i] = fx; a.f[1][i] = fy; a.f[2][i] = fz; }
    }
    // the step in front of a rebuild: the rebuild's first kernel - wrap, cell code, rank inside the cell, bucket entry, tile totals
    // (fr_count_atom, meso_device.h) - over the position the step boundary has just produced; every lane of the wave takes part
    if (a.fuse_nve && a.frc_on) fr_count_atom(a.frc, i, mine && part == 0, xn0, xn1, xn2);
#ifdef RG_STAMP
    if (g_stamp_dev && lane == 0) {
        const unsigned long long st_end = __builtin_amdgcn_s_memtime();
        unsigned long long *o = g_stamp_dev + ((size_t)blockIdx.x * RG_WAVES + w) * 10;
        o[0] = st_end - st_begin; o[1] = st_prolog - st_begin; o[2] = st_push; o[3] = st_cand; o[4] = st_heavy; o[5] = st_issue;
        o[6] = st_drained - st_light; o[7] = st_end - st_drained; o[8] = st_ncand; o[9] = (unsigned long long)qtail;
    }
#endif
}

#if RG_UNIT == 2
// fp64 style (dpd/meso): the instantiations of this translation unit, picked by the launcher of pair_ring.hip
void launch_pair_dpd_ring_dp(const PairArgs &pl, dim3 grid, dim3 block, size_t sm, hipStream_t s, bool wide, int parted, int npart, bool nt1,
                             bool ew1, bool share)
{
#define RG_L2(A, B, C)                                                                                                  \
    do {                                                                                                                \
        if (wide) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 0, true, 0>), grid, block, sm, s, pl);             \
        else if (parted == 2) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 1, true, 2>), grid, block, sm, s, pl);  \
        else if (!parted && npart == 4) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 4, true, 0>), grid, block, sm, s, pl);  \
        else if (!parted && npart == 2) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 2, true, 0>), grid, block, sm, s, pl);  \
        else if (!parted) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 1, true, 0>), grid, block, sm, s, pl);     \
        else if (npart == 4) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 4, true, (C) ? 1 : 0>), grid, block, sm, s, pl);  \
        else if (npart == 2) hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 2, true, (C) ? 1 : 0>), grid, block, sm, s, pl);  \
        else hipLaunchKernelGGL((k_pair_dpd_ring<false, A, B, C, 1, true, (C) ? 1 : 0>), grid, block, sm, s, pl);                  \
    } while (0)
#define RG_T(B, C)                                     \
    do {                                               \
        if (nt1) RG_L2(0, B, C);                       \
        else if (pl.uniform_cut) RG_L2(1, B, C);       \
        else RG_L2(2, B, C);                           \
    } while (0)
    if (share) { if (ew1) RG_T(true, true); else RG_T(false, true); }
    else { if (ew1) RG_T(true, false); else RG_T(false, false); }
#undef RG_T
#undef RG_L2
}
#else
void launch_pair_dpd_ring_dp(const PairArgs &pl, dim3 grid, dim3 block, size_t sm, hipStream_t s, bool wide, int parted, int npart, bool nt1,
                             bool ew1, bool share);
// the instantiation the last launch ran, spelled as rocprofv3 prints it: bench.py attaches profile-derived numbers to its line
// only while this is the kernel they were collected for
// (written into the caller's buffer: several engines - the in-process ranks of the LOCAL transport - launch from several host threads)

// returns false (nothing launched) for a combination the kernel has no form for; the engine reports it as an error of the run
bool launch_pair_dpd_ring(const PairArgs &p, int fast, hipStream_t s, char *variant_out)
{
    char g_last_variant[128];
    int n = p.end - p.beg;
    if (n <= 0) return true;
    PairArgs pl = p;
    pl.lds_veloc = n >= 700000 ? 1 : 0;      // (see issue(): pays from about two rounds of waves on)
    const bool nt1 = p.ntypes == 1;
    size_t ncf = nt1 ? 0 : (size_t)p.ntypes * p.ntypes * (fast ? 8 * 4 : N_COEFF * 8);
    // more than 2^25 atoms (locals + ghosts): the record word cannot hold owner lane, pairing flag and index any more
    const bool wide = (long)p.nall > (1L << 25) || p.debug == 9;      // (debug 9: the wide records on a small system - tests)
    const int ring = RG_RING;
    size_t per_wave = 64 * 16 * 2 + ring * 16 + (nt1 ? 0 : ring) + (wide ? ring : 0) + 64 * 3 * ((fast && RG_FIX32) ? 4 : 8);   // incl. this wave's share of the workgroup accumulators
    size_t sm = ((ncf + 15) & ~(size_t)15) + per_wave * RG_WAVES;
    // small launches: 2 lanes per atom, so that the same atoms fill twice as many waves (a 32^3 box is 2048 waves for 1024
    // SIMDs otherwise, and each wave walks 7 row chunks and ~11 hit batches one after the other)
    // (measured, fused launch: 25^3 21.6 -> 17.3 us, 32^3 25.7 -> 23.5; 40^3 34.4 -> 37.1: one lane per atom from there on;
    // four lanes per atom never beat two)
    int npart = p.npart > 0 ? p.npart : (n <= 163840 ? 2 : 1);
    if (npart != 1 && npart != 2 && npart != 4) npart = 1;
    if (wide) npart = 1;
    // Newton pairing needs every 256-group this launch touches to lie inside [beg, end) - or end at the last local atom
    bool share = p.share != 0 && (p.beg & (64 / npart * RG_WAVES - 1)) == 0 && (p.beg & (RG_GROUP - 1)) == 0;
    // partitioned rows serve the pairing launches of the group size they were built for (1: the walk covers the front section);
    // any other launch over such a table needs every neighbour (2: front and back section, one lane per atom, paired by the index
    // rule when it may pair at all).  (The engine does not partition for the wide records.)
    int parted = 0;
    if (wide && p.nback) return false;      // (partitioned rows under the wide record format: the engine never builds them)
    if (p.nback != nullptr) {
        parted = (share && !wide && p.part_group == 64 / npart * RG_WAVES) ? 1 : 2;
        if (parted == 2) { npart = 1; share = share && (p.beg & (RG_GROUP - 1)) == 0; }
    }
    const int awg = 64 / npart * RG_WAVES;
    dim3 grid(((n + awg - 1) / awg + 7) / 8 * 8), block(64 * RG_WAVES);
    pl.xcd_kb = pl.xcd_sb = pl.xcd_sr = 0;
    {
        // bulk and border workgroups dealt out over the XCDs separately (see the kernel); a hint that is off by a few atoms only moves
        // a workgroup from one share to the other
        const int nwg = (n + awg - 1) / awg;
        const int kb = p.bulk_hint > p.beg ? std::min(nwg, (p.bulk_hint - p.beg) / awg) : 0;
        // (launches of more than one round of workgroups only: the two shares round up separately, and a launch that just fits the
        // card in one round - 32^3 - must not grow a second one)
        if (kb >= 64 && nwg - kb >= 8 && nwg > 5 * 256) {
            pl.xcd_kb = kb; pl.xcd_sb = (kb + 7) / 8; pl.xcd_sr = (nwg - kb + 7) / 8;
            grid = dim3(8 * (pl.xcd_sb + pl.xcd_sr));
        }
    }
    bool ew1 = true;
    if (nt1) ew1 = p.cf1[P_EXPW] == 1.0;
    else ew1 = p.all_expw_one != 0;
    const bool plain = p.rng == 0 && !p.poly && !p.ftab;
#define RG_LAUNCH2(F, A, B, C, P)                                                                                     \
    do {                                                                                                              \
        if (wide) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 0, P, 0>), grid, block, sm, s, pl);                  \
        else if (parted == 2) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 1, P, 2>), grid, block, sm, s, pl);       \
        else if (!parted && npart == 4) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 4, P, 0>), grid, block, sm, s, pl);       \
        else if (!parted && npart == 2) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 2, P, 0>), grid, block, sm, s, pl);       \
        else if (!parted) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 1, P, 0>), grid, block, sm, s, pl);          \
        else if (npart == 4) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 4, P, (C) ? 1 : 0>), grid, block, sm, s, pl);       \
        else if (npart == 2) hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 2, P, (C) ? 1 : 0>), grid, block, sm, s, pl);       \
        else hipLaunchKernelGGL((k_pair_dpd_ring<F, A, B, C, 1, P, (C) ? 1 : 0>), grid, block, sm, s, pl);                       \
    } while (0)
    // (the fp64 style has no variants: always "plain")
#define RG_LAUNCH(F, A, B, C)                                   \
    do {                                                        \
        if (plain) RG_LAUNCH2(F, A, B, C, true);                \
        else RG_LAUNCH2(F, A, B, C, false);                     \
    } while (0)
#define RG_TYPES(F, B, C)                                      \
    do {                                                       \
        if (nt1) RG_LAUNCH(F, 0, B, C);                        \
        else if (p.uniform_cut) RG_LAUNCH(F, 1, B, C);         \
        else RG_LAUNCH(F, 2, B, C);                            \
    } while (0)
#define RG_PICK(F)                                        \
    if (share) {                                          \
        if (ew1) RG_TYPES(F, true, true);                 \
        else RG_TYPES(F, false, true);                    \
    } else {                                              \
        if (ew1) RG_TYPES(F, true, false);                \
        else RG_TYPES(F, false, false);                   \
    }
#ifdef RG_STAMP
    static unsigned long long *h_dev = nullptr;
    static size_t h_waves = 0;
    static long n_launch = 0;
    const size_t nw = (size_t)grid.x * RG_WAVES;
    if (nw > h_waves) {
        if (h_dev) (void)hipFree(h_dev);
        (void)hipMalloc((void **)&h_dev, nw * 10 * sizeof(unsigned long long));
        h_waves = nw;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_dev), &h_dev, sizeof h_dev);
    }
    (void)hipMemsetAsync(h_dev, 0, nw * 10 * sizeof(unsigned long long), s);
#endif
    snprintf(g_last_variant, sizeof g_last_variant, "k_pair_dpd_ring<%s, %d, %s, %s, %d, %s, %d>", fast ? "true" : "false",
             nt1 ? 0 : (p.uniform_cut ? 1 : 2), ew1 ? "true" : "false", share ? "true" : "false", wide ? 0 : npart,
             (plain || !fast) ? "true" : "false", parted);
    if (variant_out) snprintf(variant_out, 128, "%s", g_last_variant);
#ifdef RG_FEW
    // (timing builds, tools/build_variant.sh: only the instantiations of the one-type benchmark decks)
    if (!(fast && nt1 && ew1 && share && plain)) return false;      // (timing build: variant not compiled)
    RG_LAUNCH2(true, 0, true, true, true);
#else
    // (the fp64 instantiations are compiled by their own translation unit, pair_ring_dp.hip: half the build time)
    if (fast) { RG_PICK(true) } else launch_pair_dpd_ring_dp(pl, grid, block, sm, s, wide, parted, npart, nt1, ew1, share);
#endif
#undef RG_PICK
#undef RG_TYPES
#undef RG_LAUNCH
#undef RG_LAUNCH2
#ifdef RG_STAMP
    if (++n_launch % 97 == 50) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(nw * 10);
        (void)hipMemcpy(h.data(), h_dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[10] = {0};
        size_t live = 0;
        for (size_t k = 0; k < nw; k++) {
            if (!h[k * 10]) continue;
            live++;
            for (int q = 0; q < 10; q++) sum[q] += (double)h[k * 10 + q];
        }
        fprintf(stderr, "stamp launch %ld (%s, n %d): waves %zu  cycles/wave total %.0f prologue %.0f push %.0f cand(step+test+nested heavy) %.0f heavy %.0f issue %.0f final-drain %.0f epilogue %.0f | cand/wave %.0f hits/wave %.0f\n",
                n_launch, g_last_variant, n, live, sum[0] / live, sum[1] / live, sum[2] / live, sum[3] / live, sum[4] / live, sum[5] / live, sum[6] / live,
                sum[7] / live, sum[8] / live, sum[9] / live);
    }
#endif
    return true;
}

// lanes per atom the launcher picks for a launch over n atoms (the engine tags rows for the matching pairing group)
int pair_ring_group_for(int n, int npart_opt)
{
    int npart = npart_opt > 0 ? npart_opt : (n <= 163840 ? 2 : 1);
    if (npart != 1 && npart != 2 && npart != 4) npart = 1;
    return 64 / npart * RG_WAVES;
}

int pair_ring_group() { return RG_GROUP; }

#endif

} // namespace meso
