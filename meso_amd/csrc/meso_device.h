// Device-side math for the DPD hot path on gfx950 (wave64).
//
// Restates, for CDNA4, the arithmetic of /root/reference/src/USER-MESO/math_meso.h:
//   TEA block cipher            :444-464      gaussian_TEA<4>        :466-474
//   gaussian_TEA_fast<4>        :476-484      __cospi / __log2u      :380-424
//   __rsqrt / __sqrtd / __rcp   :210-238      __powd                 :332-344
//   bit_space3 / interleave3    :166-183      __mantissa             :436-442
// Every fused multiply-add is an explicit fma(); the translation unit is compiled with
// -ffp-contract=off so that the CPU oracle (oracle/meso_ref.c) can reproduce each per-pair
// quantity bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace meso {

typedef uint32_t u32;

// Kernel arguments live in device memory, cold in every cache when a kernel starts; the compiler loads them where they are first needed,
// in stages that depend on each other through the control flow (the ring kernel: five stages, ~0.6 us each, in front of its first vector
// load).  One scalar load per 64-byte line at the top of the kernel brings the whole block into the scalar cache in ONE round trip.
#ifndef MESO_KA_PREFETCH
#define MESO_KA_PREFETCH 1
#endif
#define MESO_KA_LINE(n) ".if %c2 > " #n "\n\ts_load_dword %0, %1, 64*" #n "\n\t.endif\n\t"
template <int BYTES> __device__ inline void prefetch_kernargs()
{
    if (!MESO_KA_PREFETCH) return;
    static_assert(BYTES <= 40 * 64, "kernel argument block of at most 40 lines");
    // (ONE asm statement: every request is out before the wait, and nothing of it is in flight when the compiler's code goes on -
    // it does not know that the statement loads, so a result register must not be pending behind it)
    const char __attribute__((address_space(4))) *ka = (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    u32 t;
    asm volatile("s_load_dword %0, %1, 0\n\t"
                 MESO_KA_LINE(1) MESO_KA_LINE(2) MESO_KA_LINE(3) MESO_KA_LINE(4) MESO_KA_LINE(5) MESO_KA_LINE(6) MESO_KA_LINE(7)
                 MESO_KA_LINE(8) MESO_KA_LINE(9) MESO_KA_LINE(10) MESO_KA_LINE(11) MESO_KA_LINE(12) MESO_KA_LINE(13) MESO_KA_LINE(14)
                 MESO_KA_LINE(15) MESO_KA_LINE(16) MESO_KA_LINE(17) MESO_KA_LINE(18) MESO_KA_LINE(19) MESO_KA_LINE(20) MESO_KA_LINE(21)
                 MESO_KA_LINE(22) MESO_KA_LINE(23) MESO_KA_LINE(24) MESO_KA_LINE(25) MESO_KA_LINE(26) MESO_KA_LINE(27) MESO_KA_LINE(28)
                 MESO_KA_LINE(29) MESO_KA_LINE(30) MESO_KA_LINE(31) MESO_KA_LINE(32) MESO_KA_LINE(33) MESO_KA_LINE(34) MESO_KA_LINE(35)
                 MESO_KA_LINE(36) MESO_KA_LINE(37) MESO_KA_LINE(38) MESO_KA_LINE(39)
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(t) : "s"(ka), "i"((BYTES + 63) / 64) : "memory");
}
typedef unsigned long long u64;

#define MESO_WAVE 64
#define MESO_EPSILON_SQ 1.0E-20
#define MESO_LN_2 6.9314718055994528623E-1
#define MESO_1_OVER_SQ2 7.0710678118654757274E-1
#define MESO_SQRT_2 1.4142135623730950488
#define MESO_2_TO_MINUS_31 4.6566128730773925781E-10
#define MESO_2_TO_MINUS_32 2.3283064365386962891E-10

// coefficient table layout, pair_dpd_meso.h:15-24
enum { P_CUT = 0, P_CUTSQ = 1, P_CUTINV = 2, P_EXPW = 3, P_A0 = 4, P_GAMMA = 5, P_SIGMA = 6, N_COEFF = 7 };

__host__ __device__ inline u32 bit_space3(u32 x)
{
    x = (x | (x << 12)) & 0X00FC003FU;
    x = (x | (x << 6)) & 0X381C0E07U;
    x = (x | (x << 4)) & 0X190C8643U;
    x = (x | (x << 2)) & 0X49249249U;
    return x;
}
__host__ __device__ inline u32 interleave3(u32 i, u32 j, u32 k)
{
    return bit_space3(i) | (bit_space3(j) << 1) | (bit_space3(k) << 2);
}

__device__ inline u32 mantissa3(float u, float v, float w)
{
    u32 i = __float_as_uint(u) & 0X7FF000U;
    u32 j = __float_as_uint(v) & 0X7FF000U;
    u32 k = __float_as_uint(w) & 0X7FF000U;
    return interleave3(i >> 12, j >> 12, k >> 12);
}

#define MESO_TEA_K0 0xA341316Cu
#define MESO_TEA_K1 0xC8013EA4u
#define MESO_TEA_K2 0xAD90777Du
#define MESO_TEA_K3 0x7E95761Eu
#define MESO_TEA_DT 0x9E3779B9u

template <int N>
__host__ __device__ inline void tea_core(u32 &v0, u32 &v1)
{
    u32 sum = 0;
#pragma unroll
    for (int n = 0; n < N; n++) {
        sum += MESO_TEA_DT;
        v0 += ((v1 << 4) + MESO_TEA_K0) ^ (v1 + sum) ^ ((v1 >> 5) + MESO_TEA_K1);
        v1 += ((v0 << 4) + MESO_TEA_K2) ^ (v0 + sum) ^ ((v0 >> 5) + MESO_TEA_K3);
    }
}

template <int N>
__host__ __device__ inline u32 premix_tea(u32 v0, u32 v1)
{
    tea_core<N>(v0, v1);
    return v0 ^ v1;
}

// per-particle signature, atom_vec_meso.cu:164
__device__ inline u32 signature(u32 step_seed, int tag, float vx, float vy, float vz)
{
    return step_seed ^ premix_tea<16>(__brev((u32)tag), mantissa3(vx, vy, vz));
}

__device__ inline double two_to_n(int n) { return __longlong_as_double(((long long)(1023 + n)) << 52); }

__device__ inline double rsqrt_poly(double x)
{
    double xr = __longlong_as_double(0X5FE660FCB5422422LL - (__double_as_longlong(x) >> 1));
    double x2m = x * -0.5;
    xr *= fma(xr * xr, x2m, 1.5);
    xr *= fma(xr * xr, x2m, 1.5);
    xr *= fma(xr * xr, x2m, 1.5);
    xr *= fma(xr * xr, x2m, 1.5);
    return xr;
}
__device__ inline double sqrtd_poly(double x) { return x * rsqrt_poly(x); }

__device__ inline double rcp_poly(double x)
{
    double xinv = __longlong_as_double(0X7FDE62361B1C4042LL - __double_as_longlong(x));
    xinv -= fma(x, xinv, -1.) * xinv;
    xinv -= fma(x, xinv, -1.) * xinv;
    xinv -= fma(x, xinv, -1.) * xinv;
    xinv -= fma(x, xinv, -1.) * xinv;
    return xinv;
}

__device__ inline double log2d_frac(double x)
{
    bool pred = x > MESO_SQRT_2;
    x *= pred ? 0.5 : MESO_1_OVER_SQ2;
    double z = (x - 1.) * rcp_poly(x + 1.);
    double y = z * z * 33.9705627484771406;
    double s = 4.0928048937567843469E-12;
    s = fma(s, y, 1.4374842194796670219E-10);
    s = fma(s, y, 5.7988453014506741861E-9);
    s = fma(s, y, 2.4074128088151586443E-7);
    s = fma(s, y, 1.0514733588011180538E-5);
    s = fma(s, y, 5.0006798065881969549E-4);
    s = fma(s, y, 2.8312651192953993354E-2);
    s = fma(s, y, 2.8853900817779268114E+0);
    return fma(z, s, (pred ? 1.0 : 0.5));
}

__device__ inline double exp2d_frac(double x)
{
    double s = 6.3026908837748924689E-10;
    s = fma(s, x, 6.5379419072372670333E-9);
    s = fma(s, x, 1.0258347084283025531E-7);
    s = fma(s, x, 1.3207676270599404858E-6);
    s = fma(s, x, 1.5253232908458899497E-5);
    s = fma(s, x, 1.5403509189194102748E-4);
    s = fma(s, x, 1.3333558738165095559E-3);
    s = fma(s, x, 9.6181290971755593396E-3);
    s = fma(s, x, 5.5504108665909870679E-2);
    s = fma(s, x, 2.4022650695904222220E-1);
    s = fma(s, x, 6.9314718055994653980E-1);
    s = fma(s, x, 9.9999999999999999572E-1);
    return s;
}

__device__ inline double powd_poly(double a, double b)
{
    long long bits = __double_as_longlong(a);
    int hi = (int)(bits >> 32);
    u32 lo = (u32)bits;
    double I = (hi >> 20) - 1023;
    long long fb = ((long long)((hi & 0X000FFFFF) | 0X3FF00000) << 32) | lo;
    double F = log2d_frac(__longlong_as_double(fb));
    double II = floor(b * (I + F));
    return two_to_n((int)II) * exp2d_frac(fma(b, F, fma(b, I, -II)));
}

__device__ inline double cospi_poly(double x)
{
    x = 2.0 * x - 1.0;
    double x2 = x * x;
    double s = 3.41817283473266926E-6;
    s = fma(s, x2, -1.60217135750921262E-4);
    s = fma(s, x2, 4.68162024021793872E-3);
    s = fma(s, x2, -7.96925872866600517E-2);
    s = fma(s, x2, 6.45964092644060746E-1);
    s = fma(s, x2, -1.57079632662144460E+0);
    return s * x;
}

__device__ inline double log2u_poly(u32 x)
{
    int I = 31 - __clz((int)x);
    double xx = (double)x * two_to_n(-I);
    double ex = I - 32;
    bool pred = xx > MESO_SQRT_2;
    xx *= pred ? 0.5 : MESO_1_OVER_SQ2;
    double z = (xx - 1.) * rcp_poly(xx + 1.);
    double y = z * z * 33.9705627484771406;
    double s = 2.55854634203511155E-7;
    s = fma(s, y, 1.05013262724846015E-5);
    s = fma(s, y, 5.00072802051539862E-4);
    s = fma(s, y, 2.83126505877817866E-2);
    s = fma(s, y, 2.88539008179006374E+0);
    return fma(z, s, (pred ? 1.0 : 0.5) + ex);
}

// xi_ij in [-4,4], symmetric in (u,v): the larger signature is always v0.
__device__ inline double gaussian_tea(u32 u, u32 v)
{
    bool pred = u > v;
    u32 v0 = pred ? u : v, v1 = pred ? v : u;
    tea_core<4>(v0, v1);
    double f = cospi_poly((v0 & 0X7FFFFFFFu) * MESO_2_TO_MINUS_31) * ((v0 & 0X80000000u) ? 1.0 : -1.0);
    double r = sqrtd_poly(-2.0 * MESO_LN_2 * log2u_poly(v1 > 1u ? v1 : 1u));
    return fmax(-4.0, fmin(r * f, 4.0));
}

// fp32 variant: CUDA's sinpif/log2f/sqrtf become the gfx950 transcendental unit
// (v_sin_f32 takes revolutions: sin(2*pi*x)), v_log_f32, v_sqrt_f32.
__device__ inline float gaussian_tea_fast(u32 u, u32 v)
{
    bool pred = u > v;
    u32 v0 = pred ? u : v, v1 = pred ? v : u;
    tea_core<4>(v0, v1);
    // (float)(int)v0 * 2^-31 in [-1,1), halved for v_sin_f32's revolutions: one multiplication by 2^-32 - powers of two, the
    // same bits; the clamp to +-4 is one v_med3_f32 (its NaN rule differs from fmin/fmax only when v0 = v1 = 0: 2^-64)
    float f = __builtin_amdgcn_sinf((float)(int)v0 * (float)MESO_2_TO_MINUS_32);                  // sin(pi*t)
    float lg = __builtin_amdgcn_logf((float)v1 * (float)MESO_2_TO_MINUS_32);
    float r = __builtin_amdgcn_sqrtf(-2.0f * (float)MESO_LN_2 * lg);
    return __builtin_amdgcn_fmed3f(r * f, -4.0f, 4.0f);
}

// pair_style dpd/mini/meso (pair_dpd_minimal_meso.cu:50-89): mean0var1<8> - the sum of the two signatures mapped to
// [-1,1), four rounds of the degree-4 Chebyshev map 8x^4 - 8x^2 + 1 (arcsine law on [-1,1], variance 1/2), times sqrt 2.
// The reference evaluates the map with __fmaf_rz; gfx950 has no per-instruction rounding, so the two FMAs of a round run
// between two writes of the wave's fp32 rounding mode (MODE[1:0] = 3: toward zero) inside one asm block - the mode never
// leaks, and the stream equals the oracle's fesetround(FE_TOWARDZERO) + fmaf bit for bit.
__device__ inline float logistic_round_rz(float x)
{
    const float x2 = x * x, c8 = 8.0f, m8 = -8.0f;
    float r;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_fma_f32 %0, %2, %1, %3\n\t"
                 "v_fma_f32 %0, %0, %1, 1.0\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(r)
                 : "v"(x2), "v"(c8), "v"(m8));
    return r;
}
__device__ inline float logistic_noise(u32 u, u32 v)
{
    float x = (float)u * (float)MESO_2_TO_MINUS_32 + (float)v * (float)MESO_2_TO_MINUS_32 - 1.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) x = logistic_round_rz(x);
    return x * 1.41421356237309514547f;
}
// uniform_TEA_fast<4> (math_meso.h:501-505): uniform on [-sqrt 3, sqrt 3) (variance 1), the noise of dpd/tableforce/meso;
// called with (min, max) of the two signatures (pair_dpd_tableforce_meso.cu:171)
__device__ inline float uniform_tea_fast(u32 u, u32 v)
{
    u32 v0 = u < v ? u : v, v1 = u < v ? v : u;
    tea_core<4>(v0, v1);
    return (float)(v0 ^ v1) * (float)(1.73205080756887729353 * MESO_2_TO_MINUS_31) - (float)1.73205080756887729353;
}
// the pair noise of the fp32 styles: rng 0 = TEA-keyed Gaussian (dpd/fast/meso), 1 = logistic map (dpd/mini/meso),
// 2 = TEA-keyed uniform (dpd/tableforce/meso)
__device__ inline float pair_noise_fast(int rng, u32 u, u32 v)
{
    return rng == 1 ? logistic_noise(u, v) : rng == 2 ? uniform_tea_fast(u, v) : gaussian_tea_fast(u, v);
}
// dpd/tableforce/meso: the conservative force is read from a table of L points, uniform in r/rc over [0,1], with the linear
// filter of the texture unit the reference samples it with (tex1DLayered, clamp addressing, pair_dpd_tableforce_meso.cu:81-84,
// :181, coordinate transform :291): position x = (r/rc)(L-1), weight frac(x) kept to 8 fractional bits as the CUDA
// programming guide defines linear filtering
__host__ __device__ inline float table_force_f32(float rrinv, const float *tab, int len)
{
    float x = rrinv * (float)(len - 1);
    x = x < 0.f ? 0.f : x;
    int i = (int)x;
    if (i > len - 1) i = len - 1;
    const int i1 = i + 1 < len ? i + 1 : len - 1;
    const float al = floorf((x - (float)i) * 256.0f + 0.5f) * (1.0f / 256.0f);
    return (1.0f - al) * tab[i] + al * tab[i1];
}

// pair_style dpd/polyforce/meso: the conservative force is a polynomial in w = 1 - r/rc, Horner from the highest order
// (polyval / polyval_integral, math_meso.h:53-66); table row = [order][c_order .. c_0], MESO_POLY_PITCH floats per type pair
#define MESO_POLY_MAXLEN 32
#define MESO_POLY_PITCH (MESO_POLY_MAXLEN + 1)
__host__ __device__ inline float polyval_f32(float x, const float *row)
{
    const int order = (int)row[0];
    float r = row[1];
    for (int k = 0; k < order; k++) r = r * x + row[2 + k];
    return r;
}
__host__ __device__ inline float polyval_integral_f32(float x, const float *row)
{
    const int order = (int)row[0];
    float r = row[1] / (float)(order + 1);
    for (int k = 0; k < order; k++) r = r * x + row[2 + k] / (float)(order - k);
    return r * x;
}

// dpd/meso pair force in the reference's mixed precision (fp32 operands, fp64 arithmetic; gpu_dpd<0>
// pair_dpd_meso.cu:120-160), one definition for the force kernels; compiled uncontracted (explicit fma only).
struct PairCoeff64 { double cutinv, expw, a0, gamma, sigma; };
template <bool EW1>
// cutsq_exact >= 0: the caller filtered with a widened fp32 test; the exact test of the reference (rsq < cutsq && rsq >= epsilon, on
// the same fp64 r^2 as rsq_f64) is made here and a pair that fails it gets a zero force
__device__ inline void pair_dpd_f64(const float4 ci, const float4 cj, const float4 vi, const float4 vj, const PairCoeff64 &c,
                                    double dt_inv_sqrt, double &fx, double &fy, double &fz, double cutsq_exact = -1.0)
{
    const double dx = (double)ci.x - (double)cj.x, dy = (double)ci.y - (double)cj.y, dz = (double)ci.z - (double)cj.z;
    const double rsq = dx * dx + dy * dy + dz * dz;
    if (cutsq_exact >= 0.0 && !(rsq < cutsq_exact && rsq >= MESO_EPSILON_SQ)) { fx = fy = fz = 0.0; return; }
    const double rn = gaussian_tea(__float_as_uint(vi.w), __float_as_uint(vj.w));
    const double rinv = rsqrt(rsq);
    const double r = rsq * rinv;
    const double dvx = (double)vi.x - (double)vj.x, dvy = (double)vi.y - (double)vj.y, dvz = (double)vi.z - (double)vj.z;
    const double dot = dx * dvx + dy * dvy + dz * dvz;
    const double wc = 1.0 - r * c.cutinv;
    double wr = wc;
    if (!EW1 && c.expw != 1.0) wr = powd_poly(wc, c.expw);
    double fpair = c.a0 * wc - (c.gamma * wr * wr * dot * rinv) + (c.sigma * wr * rn * dt_inv_sqrt);
    fpair *= rinv;
    fx = dx * fpair; fy = dy * fpair; fz = dz * fpair;
}
__device__ inline double rsq_f64(const float4 a, const float4 b)
{
    const double dx = (double)a.x - (double)b.x, dy = (double)a.y - (double)b.y, dz = (double)a.z - (double)b.z;
    return dx * dx + dy * dy + dz * dz;
}
// fp64 force component -> 64-bit fixed point with 36 fractional bits (1.5e-11 resolution, +-32768 range; the argument
// is clamped): adding 1.5 * 2^16 puts the value into the mantissa of a double whose ulp is 2^-36, and the difference of
// the bit patterns IS the two's-complement fixed-point number.
#define MESO_FIXED36_MAGIC 98304.0
__device__ inline u64 to_fixed36(double x)
{
    x = fmin(fmax(x, -32000.0), 32000.0);
    return (u64)__double_as_longlong(x + MESO_FIXED36_MAGIC) - (u64)__double_as_longlong(MESO_FIXED36_MAGIC);
}
__device__ inline double from_fixed36(u64 a) { return (double)(long long)a * (1.0 / 68719476736.0); }

// Step boundary of one atom: final_integrate of step s, initial_integrate of step s+1 (fix_nve_meso.cu:62-95,157-178)
// and, when step s+1 keeps the neighbour table, gpu_merge_xvt for step s+1 (atom_vec_meso.cu:142-167).  One definition
// for the stand-alone boundary kernel and for the force kernel's epilogue, so both produce the same bits.
// (pre: the atom's state requested ahead of time by the caller - same values, same arithmetic)
struct NvePre { double x, y, z, vx, vy, vz, dtfm; int mask, tag, type; };
// the atom's periodic images (one rank): their number and the first four table entries, requested ahead like NvePre (a face atom has
// one image, an edge atom three: the dependent pair of loads - count, then entries - otherwise ends the kernel's border waves)
struct NveImgPre { int ni; int4 e; };
__device__ inline void nve_prefetch_images(const NveArgs &a, int i, NveImgPre &p)
{
    p.ni = 0; p.e = make_int4(0, 0, 0, 0);
    if (a.img_cnt && !a.img_center && !(a.img_first && i < *a.img_first)) {
        p.ni = a.img_cnt[i];
        p.e = *reinterpret_cast<const int4 *>(a.img + (size_t)i * 8);
    }
}
// (ty > 0: the caller knows the atom's type - the force kernel has it in the merged coordinate record - and, with a.mass_type set, the
// mass comes from the per-type table [every atom's mass IS its type's, launch_unpack_mass] and the group mask is not read when the
// group is "all" [bit 0 of every mask is set, group.cpp]: 16 of the 68 bytes the step boundary reads per atom stay where they are)
__device__ inline void nve_prefetch(const NveArgs &a, int i, NvePre &p, int ty = 0)
{
    p.x = a.x[0][i]; p.y = a.x[1][i]; p.z = a.x[2][i];
    p.vx = a.v[0][i]; p.vy = a.v[1][i]; p.vz = a.v[2][i];
    const bool lean = a.mass_type && ty > 0;
    p.type = lean ? ty : a.type[i];
    // dtf / m: from the per-type table when there is one (k_dtfm_table: the same expression evaluated once per type), else here and now
    p.dtfm = (lean && a.dtfm_type) ? a.dtfm_type[ty] : a.dtf * rcp_poly(lean ? a.mass_type[ty] : a.mass[i]);
    p.mask = (lean && a.groupbit == 1) ? 1 : a.mask[i];
    p.tag = a.tag[i];
}
// (xo, yo, zo: the atom's position after the step boundary)
__device__ inline void nve_boundary_atom(const NveArgs &a, int i, double fx, double fy, double fz, const NvePre *pre, double &xo, double &yo, double &zo,
                                         int ty = 0, const NveImgPre *ipre = nullptr)
{
    double x = pre ? pre->x : a.x[0][i], y = pre ? pre->y : a.x[1][i], z = pre ? pre->z : a.x[2][i];
    double vx = pre ? pre->vx : a.v[0][i], vy = pre ? pre->vy : a.v[1][i], vz = pre ? pre->vz : a.v[2][i];
    const bool lean = !pre && a.mass_type && ty > 0;
    const int type_i = pre ? pre->type : lean ? ty : a.type[i];
    if ((pre ? pre->mask : (lean && a.groupbit == 1) ? 1 : a.mask[i]) & a.groupbit) {
        const double dtfm = pre ? pre->dtfm : (lean && a.dtfm_type) ? a.dtfm_type[ty] : a.dtf * rcp_poly(lean ? a.mass_type[ty] : a.mass[i]);
        vx += dtfm * fx; vy += dtfm * fy; vz += dtfm * fz;       // final_integrate, step s
        vx += dtfm * fx; vy += dtfm * fy; vz += dtfm * fz;       // initial_integrate, step s+1
        x += a.dtv * vx; y += a.dtv * vy; z += a.dtv * vz;
        a.v[0][i] = vx; a.v[1][i] = vy; a.v[2][i] = vz;
        a.x[0][i] = x; a.x[1][i] = y; a.x[2][i] = z;
    }
    xo = x; yo = y; zo = z;
    if (a.merge) {
        float4 c;
        c.x = (float)(x - a.cx); c.y = (float)(y - a.cy); c.z = (float)(z - a.cz);
        c.w = __uint_as_float((u32)(type_i - 1));
        a.coord4_next[i] = c;
        float4 v;
        v.x = (float)vx; v.y = (float)vy; v.z = (float)vz;
        v.w = __uint_as_float(signature(a.seed_next, pre ? pre->tag : a.tag[i], v.x, v.y, v.z));
        a.veloc4_next[i] = v;
        if (a.img_cnt && !a.img_center) {
            // the ghost refresh of step s+1 for my own periodic images (what k_pack_forward computes: same expression, same bits)
            const int ni = ipre ? min(ipre->ni, 8) : (a.img_first && i < *a.img_first) ? 0 : min(a.img_cnt[i], 8);
            for (int m = 0; m < ni; m++) {
                const int e = (ipre && m < 4) ? (m == 0 ? ipre->e.x : m == 1 ? ipre->e.y : m == 2 ? ipre->e.z : ipre->e.w) : a.img[(size_t)i * 8 + m];
                const int d = e >> 26, dest = e & 0x03FFFFFF;
                float4 g;
                g.x = (float)((x + a.img_shift[3 * d]) - a.cx);
                g.y = (float)((y + a.img_shift[3 * d + 1]) - a.cy);
                g.z = (float)((z + a.img_shift[3 * d + 2]) - a.cz);
                g.w = c.w;
                a.coord4_next[dest] = g;
                a.veloc4_next[dest] = v;
            }
        } else if (a.img_cnt) {
            // several ranks: the atom's records in the per-step refresh messages (what k_pack_forward_multi packs: the receiver's
            // centre per direction, velocities img_vofs[d] slots behind the coordinates in the peer's block of the send staging)
            const int ni = min(a.img_cnt[i], 8);
            for (int m = 0; m < ni; m++) {
                const int e = a.img[(size_t)i * 8 + m];
                const int d = e >> 26, dest = e & 0x03FFFFFF;
                float4 g;
                g.x = (float)((x + a.img_shift[3 * d]) - a.img_center[3 * d]);
                g.y = (float)((y + a.img_shift[3 * d + 1]) - a.img_center[3 * d + 1]);
                g.z = (float)((z + a.img_shift[3 * d + 2]) - a.img_center[3 * d + 2]);
                g.w = c.w;
                a.img_c4[dest] = g;
                a.img_v4[dest + a.img_vofs[d]] = v;
            }
        }
    }
}

__device__ inline void nve_boundary_atom(const NveArgs &a, int i, double fx, double fy, double fz, const NvePre *pre = nullptr)
{
    double x, y, z;
    nve_boundary_atom(a, i, fx, fy, fz, pre, x, y, z);
}

// One atom of the reorder gather (gpu_permute_copy / gpu_deinterleave with permutation, atom_vec_meso.h:11-67): atom j of the old
// order becomes atom i of the new one.  mg.coord4 != null: the merged float4 pair of the atom's new place is written as well
// (gpu_merge_xvt folded into the gather: the reorder has x, v, tag and type in registers anyway).  One definition for
// k_permute_atoms and the fused rebuild (rebuild.hip).
struct MergeOut { float4 *coord4, *veloc4; double cx, cy, cz; u32 seed; int *inverse; int *zero; };
__device__ inline void permute_one(const AtomSoA &src, const AtomSoA &dst, int j, int i, int with_f, const MergeOut &mg, double *xout = nullptr)
{
    if (mg.inverse) mg.inverse[j] = i;      // old place -> new place (the overlapped rebuild translates its send list with it)
    if (mg.zero) mg.zero[i] = 0;            // image counters of the new order (filled by the rebuild's k_pack_forward)
    double xx[3], vv[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        xx[d] = src.x[d][j];
        vv[d] = src.v[d][j];
        dst.x[d][i] = xx[d];
        dst.v[d][i] = vv[d];
        if (with_f) dst.f[d][i] = src.f[d][j];      // inside run() the forces are recomputed before anyone reads them
        if (xout) xout[d] = xx[d];
    }
    const int tg = src.tag[j], ty = src.type[j];
    if (mg.coord4) {
        float4 c, v;
        c.x = (float)(xx[0] - mg.cx); c.y = (float)(xx[1] - mg.cy); c.z = (float)(xx[2] - mg.cz);
        c.w = __uint_as_float((u32)(ty - 1));
        v.x = (float)vv[0]; v.y = (float)vv[1]; v.z = (float)vv[2];
        v.w = __uint_as_float(signature(mg.seed, tg, v.x, v.y, v.z));
        mg.coord4[i] = c;
        mg.veloc4[i] = v;
    }
    dst.tag[i] = tg;
    dst.type[i] = ty;
    dst.mask[i] = src.mask[j];
    dst.image[i] = src.image[j];
    dst.mass[i] = src.mass[j];
    // topology lists: only the entries in use travel (most atoms of a solution have none)
    if (src.bpa > 0) {
        const int nb = src.nbond[j];
        dst.nbond[i] = nb;
        for (int b = 0; b < nb; b++) {
            dst.bond_tag[(size_t)i * src.bpa + b] = src.bond_tag[(size_t)j * src.bpa + b];
            dst.bond_type[(size_t)i * src.bpa + b] = src.bond_type[(size_t)j * src.bpa + b];
        }
    }
    if (src.apa > 0) {
        const int na = src.nangle[j];
        dst.nangle[i] = na;
        for (int a = 0; a < 4 * na; a++) dst.angle_tag[(size_t)i * 4 * src.apa + a] = src.angle_tag[(size_t)j * 4 * src.apa + a];
    }
    if (src.msp > 0) {
        const int ns = src.nspecial[j];
        dst.nspecial[i] = ns;
        for (int s = 0; s < ns; s++) dst.special[(size_t)i * src.msp + s] = src.special[(size_t)j * src.msp + s];
    }
}

// border slabs: near_flags bit 2d = near the low face of dim d (sent down), bit 2d+1 = near the high face (sent up)
struct Slabs { double lo[3], hi[3]; };
__device__ inline int near_flags(double cx, double cy, double cz, const double *sl, const double *sh)
{
    int f = 0;
    if (cx <= sl[0]) f |= 1;
    if (cx >= sh[0]) f |= 2;
    if (cy <= sl[1]) f |= 4;
    if (cy >= sh[1]) f |= 8;
    if (cz <= sl[2]) f |= 16;
    if (cz >= sh[2]) f |= 32;
    return f;
}
__device__ inline bool in_dir(int flags, int dir)
{
    int sx = dir % 3 - 1, sy = (dir / 3) % 3 - 1, sz = dir / 9 - 1;
    bool okx = sx == 0 || (sx < 0 ? (flags & 1) : (flags & 2));
    bool oky = sy == 0 || (sy < 0 ? (flags & 4) : (flags & 8));
    bool okz = sz == 0 || (sz < 0 ? (flags & 16) : (flags & 32));
    return okx && oky && okz;
}

// bins: neighbor_meso.cu:410-412 clamp at [nmin,nmax)
__host__ __device__ inline int clampi(int i, int nmin, int nmax)
{
    int a = i < nmax - 1 ? i : nmax - 1;
    return a > nmin ? a : nmin;
}

// Row layout of the cell-ordered table ("chunked-8"): 8 consecutive entries of one atom form one 32-byte word,
// word(i, c) = ((i>>6)*(n_col/8) + c)*64 + (i&63).  A lane writes/reads whole 32-B sectors (no partial-sector
// writes: the 4-byte scattered stores of the transposed layout cost 8x the bytes at HBM, profiles/r01_pmc_*),
// and a wave's access to chunk c of its 64 atoms is one contiguous 2 KiB.
// rank of every lane inside its code, with ONE atomic per run of equal codes in the wave: the atoms arrive nearly sorted, so a
// wave holds ~7 runs of ~9 equal codes, and same-address atomics serialise in L2 (61 us for 1 M single atomics, 64^3)
// tot != null: the head of a run also books the run into the total of its group of (1 << tshift) codes
__device__ inline int run_rank(u32 code, bool valid, int *__restrict__ cnt, int *__restrict__ tot = nullptr, int tshift = 0)
{
    const int lane = __lane_id();
    // (both shuffles outside the short-circuit expression: every lane has to take part in them)
    const u32 prev = __shfl_up(code, 1, 64);
    const int prev_valid = __shfl_up((int)valid, 1, 64);
    const bool head = valid && (lane == 0 || code != prev || !prev_valid);
    const unsigned long long heads = __ballot(head), live = __ballot(valid);
    int rank = 0;
    int start = 0, base = 0;
    if (valid) {
        const unsigned long long below = heads & ((2ull << lane) - 1ull);         // heads at or below my lane (never empty)
        start = 63 - __builtin_clzll(below);
        const unsigned long long after = (heads | ~live) & ~((2ull << start) - 1ull);   // next head or dead lane behind the run's head
        const int end = after ? __builtin_ctzll(after) : 64;
        if (lane == start) {
            base = atomicAdd(cnt + code, end - start);
            if (tot) atomicAdd(tot + (code >> tshift), end - start);
        }
    }
    base = __shfl(base, start, 64);
    if (valid) rank = base + (lane - start);
    return rank;
}

// tot[group] += number of valid lanes with that group, ONE atomic per distinct group in the wave (atoms arrive nearly sorted:
// one or two groups per wave).  Same-address atomics cost ~150 ns each on this chip: one per run of equal codes made the tile
// totals of the fused rebuild cost 20 us at 32^3.
__device__ inline void wave_group_add(u32 group, bool valid, int *__restrict__ tot)
{
    unsigned long long rest = __ballot(valid);
    while (rest) {
        const int first = __builtin_ctzll(rest);
        const u32 g0 = (u32)__builtin_amdgcn_readlane((int)group, first);
        const unsigned long long m = __ballot(valid && group == g0);
        if (__lane_id() == first) atomicAdd(tot + g0, __popcll(m));
        rest &= ~m;
    }
}

// Bit d of the result: the atom has a periodic image in direction d = (sx+1) + 3 (sy+1) + 9 (sz+1).  fl: near_flags of its
// coordinate; (bx, by, bz): its cell.  An image sent up (s = +1) comes from the last cell below the high face, one sent down
// from the first cell above the low face (an atom ON the slab plane of a box whose cells are exactly one ghost cutoff wide can
// sit in the cell next to it: not an image for any kernel here - they all use this function).
__device__ inline u32 image_mask(int fl, int bx, int by, int bz, const int *mbin, u32 dir_mask)
{
    const u32 mx = 2u | ((fl & 1) && bx == 1 ? 1u : 0u) | ((fl & 2) && bx == mbin[0] - 2 ? 4u : 0u);
    const u32 my = 2u | ((fl & 4) && by == 1 ? 1u : 0u) | ((fl & 8) && by == mbin[1] - 2 ? 4u : 0u);
    const u32 mz = 2u | ((fl & 16) && bz == 1 ? 1u : 0u) | ((fl & 32) && bz == mbin[2] - 2 ? 4u : 0u);
    u32 row = 0;                                   // 9 bits: (sx, sy)
#pragma unroll
    for (int k = 0; k < 3; k++) row |= ((my >> k) & 1u) ? mx << (3 * k) : 0u;
    u32 m = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) m |= ((mz >> k) & 1u) ? row << (9 * k) : 0u;
    return m & ~(1u << 13) & dir_mask;
}

// The periodic images of a wave's border atoms booked per tile of FR_COUNT_GTILE ghost cells (k_fr_ghosts PULLS the ghosts of a ghost
// cell from the cell they are images of, so all it needs beforehand is the number of ghosts per tile: its first slot is the sum of the
// totals in front of it).  emask: image_mask of the lane's atom (0: none); (bx, by, bz): its cell.  The ghost cell of an image is the
// geometric image of the atom's own cell.  No returning atomics: nothing waits.  Every lane of the wave calls.
#define FR_COUNT_GTILE 32
__device__ inline void book_images(u32 emask, int bx, int by, int bz, const int *mbin, int *__restrict__ gttot)
{
    u32 any = emask;                  // directions some lane of the wave has an image in
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) any |= (u32)__shfl_xor((int)any, o, 64);
    any = (u32)__builtin_amdgcn_readfirstlane((int)any);
    while (any) {
        const int dir = __builtin_ctz(any);
        any &= any - 1u;
        const bool em = (emask >> dir) & 1u;
        const int sx = dir % 3 - 1, sy = (dir / 3) % 3 - 1, sz = dir / 9 - 1;
        const u32 gc = interleave3((u32)(sx == 0 ? bx : (sx > 0 ? 0 : mbin[0] - 1)), (u32)(sy == 0 ? by : (sy > 0 ? 0 : mbin[1] - 1)),
                                   (u32)(sz == 0 ? bz : (sz > 0 ? 0 : mbin[2] - 1)));
        wave_group_add(em ? gc / FR_COUNT_GTILE : 0u, em, gttot);
    }
}

// one atom of the rebuild's count (FrCountArgs, kernels.h): every lane of the wave calls (valid: the lane holds an atom to count),
// (cx, cy, cz) = the atom's position.  FR_COUNT_TILE codes per tile of the placing kernel (rebuild.hip)
#define FR_COUNT_TILE 64
__device__ inline void fr_count_atom(const FrCountArgs &a, int i, bool valid, double cx, double cy, double cz)
{
    u32 e = 0, key = 0, emask = 0;
    int cell[3] = {0, 0, 0};
    if (valid) {
        double c[3] = {cx, cy, cz};
        if (a.wrap) {
            const int img = a.image[i];
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
                a.x[0][i] = c[0]; a.x[1][i] = c[1]; a.x[2][i] = c[2];
                a.image[i] = im[0] | (im[1] << 10) | (im[2] << 20);
            }
        }
        const int res = 1 << (a.sub_bits / 3);
        u32 b[3], sc[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            b[d] = (u32)clampi((int)((c[d] - a.g.lo[d]) * a.g.bininv[d] + 1), 0, a.g.mbin[d]);
            sc[d] = (u32)clampi((int)((c[d] - a.g.lo[d] - ((double)b[d] - 1) * a.g.binsize[d]) * (res * a.g.bininv[d])), 0, res);
        }
        e = interleave3(b[0], b[1], b[2]);
        key = interleave3(sc[0], sc[1], sc[2]);      // sub-cell Morton key: the order inside the cell (gpu_build_reorder_keypair)
        const bool border = c[0] <= a.sl_lo[0] || c[0] >= a.sl_hi[0] || c[1] <= a.sl_lo[1] || c[1] >= a.sl_hi[1] || c[2] <= a.sl_lo[2] ||
                            c[2] >= a.sl_hi[2];
        if (border) e += (u32)a.M;
        cell[0] = (int)b[0]; cell[1] = (int)b[1]; cell[2] = (int)b[2];
        if (border && a.gttot) emask = image_mask(near_flags(c[0], c[1], c[2], a.sl_lo, a.sl_hi), cell[0], cell[1], cell[2], a.g.mbin, a.dir_mask);
    }
    // rank inside the code, one atomic per run of equal codes; tile totals: one atomic per tile and wave
    const int rank = run_rank(e, valid, a.cnt);
    wave_group_add(e / FR_COUNT_TILE, valid, a.ttot);
    if (a.gttot) book_images(emask, cell[0], cell[1], cell[2], a.g.mbin, a.gttot);
    if (!valid) return;
    // (sub-cell key, old index) travels as one word: the placing kernel orders a cell without touching the coordinates
    const unsigned long long ent = ((unsigned long long)key << 32) | (u32)i;
    if (rank < a.cap) a.bucket[(size_t)e * a.cap + rank] = ent;
    else {
        const int o = atomicAdd(a.novf, 1);
        if (o < a.ovf_cap) { a.ovf[2 * o] = (unsigned long long)e; a.ovf[2 * o + 1] = ent; }
        else atomicMax(a.flags, 300000);
    }
}

__device__ inline size_t row_word8(int i, int c, int n_col) { return ((size_t)(i >> 6) * (n_col >> 3) + c) * 64 + (i & 63); }

// x -> two's-complement 64-bit fixed point with 32 fractional bits: floor(x) in the high word, fract(x) * 2^32 in the low
// word (exact for every fp32 x with |x| < 2^19 down to 2^-32 resolution; a larger force component - a diverging run - is
// clamped to +-2^19 instead of wrapping into a plausible-looking sum: v_med3_f32, one instruction)
__device__ inline u64 to_fixed(float x)
{
    x = __builtin_amdgcn_fmed3f(x, -524287.0f, 524287.0f);
    // x + 1.5 * 2^20 as a double has x * 2^32 (rounded to nearest, two's complement) in its low 52 bits; taking the exponent and
    // the 1.5 off the high word leaves the 64-bit fixed-point number (|x| < 2^19): cvt, add, one integer add
    const double y = (double)x + 1572864.0;
    const u64 b = (u64)__double_as_longlong(y);
    return ((u64)((u32)(b >> 32) - 0x41380000u) << 32) | (u32)b;
}
__device__ inline double from_fixed(u64 a) { return (double)(long long)a * (1.0 / 4294967296.0); }
// fp32 style of the ring kernel: 32-bit fixed point with 16 fractional bits - one multiply and v_cvt_i32_f32 (round toward zero:
// symmetric in the sign, so a pair evaluated from both sides still sums to zero) instead of the four-instruction 64-bit
// conversion, and ds_add_u32 instead of ds_add_u64.  Resolution 1.5e-5 (the fp32 arithmetic of one pair force of size 100 is
// good to 1e-5), range +-32768 for the SUM of an atom's pair forces: partial sums may wrap, the total must fit (a component
// beyond that belongs to a run that has diverged; the single term saturates in the conversion).
#define MESO_FIXED16_SCALE 65536.0f
__device__ inline u32 to_fixed16(float x) { return (u32)(int)(x * MESO_FIXED16_SCALE); }
__device__ inline double from_fixed16(u32 a) { return (double)(int)a * (1.0 / 65536.0); }

struct Shift27 { double s[27][3]; };   // shift added to x for each direction (0 when not crossing a PBC)
struct Center27 { double c[27][3]; };  // merged-coordinate origin of the receiver of each direction

// Arguments of the three-launch rebuild of one rank (rebuild.hip)
struct FusedArgs {
    int novf_later;            // no ghost stage: the overflow count is cleared by the caller's next kernel, not by a memset launch
    const int *skip;           // several ranks: migration code of every atom below skip_n (13 = stays); leavers are holes the count skips,
    int skip_n;                // arrivals sit behind skip_n - the stayers are not compacted before the reorder
    AtomSoA src, dst;          // old order -> new order (ghosts go behind the n locals of dst)
    int n;                     // local atoms
    int with_f;                // the gather carries the forces along
    int wrap;                  // MesoDomain::pbc folded into the count kernel
    double boxlo[3], boxhi[3];
    int per[3];
    BinGeom g;
    Slabs sl;
    int sub_bits, M;
    // locals: counts per extended code [2M+1], buckets [2M][cap], overflow list [(code, entry)], totals per tile / supertile
    int *cnt, cap;
    unsigned long long *bucket, *ovf;      // (sub-cell key << 32) | old index
    int *novf, ovf_cap;
    int *ttot, *ttot_next;     // atoms per tile of 64 codes (double-buffered by rebuild parity: a tile clears the other buffer's entry)
    int *stot;                 // atoms per supertile of 256 tiles, summed from ttot by k_fr_super - only when there are more than
                               // FR_DIRECT_TILES tiles (null otherwise: every tile adds up the tile totals in front of it directly)
    int *estart;               // out [2M+1]
    int *perm;                 // out, nullable: new place -> old place
    int split_gather;          // the placing kernel only orders (writes perm), the payload moves in a streaming pass of its own (k_fr_gather)
    unsigned long long *scratch;   // [n] (a code denser than the LDS stage)
    int lds_cap;               // entries of the LDS stage of one pass
    MergeOut mg;               // merged pairs of the new order (+ the ghosts'), image counters cleared
    // ghosts (gttot null: no ghost stage - several ranks create their ghosts by exchange): ghosts per tile of 64 ghost cells
    int *gttot, *gttot_next, *gstot;
    int img_booked;            // the count booked the periodic images (gttot): the placing / gathering kernels do not
    int merged_ghosts;         // the ghost tiles run in the gather's launch (k_fr_gather_ghosts): set by launch_fused_rebuild's caller
    unsigned dir_mask;         // bit d: direction d is a periodic image direction of this rank
    int *gstart;               // out [M+1]
    int *gcnt_out;             // out, nullable: ghosts per cell [M]; with it only the tiles listed in gorder run
    const int *gorder;         // [ngorder] tiles (of FR_GTILE ghost cells) that hold at least one ghost cell of the bin grid
    int ngorder;
    Shift27 sh;
    Center27 ce;
    int *sendlist;             // out: ghost slot -> source atom (new order)
    unsigned char *senddir;    // out: ghost slot -> direction
    int *img_cnt, *img;        // nullable: image table of the step-boundary epilogue
    int ghost_cap;             // ghosts the arrays can take
    int *dir_start;            // device [28]: [27] = ghost count (the per-step refresh loops to it)
    int *flags;                // device flags ([0] overflow code)
    int *report;               // pinned host memory as the device sees it
    int report_seq;            // != 0: written to report[12] BEHIND the counts (system-scope fence): the host polls it instead of an event
};
void launch_fused_rebuild(const FusedArgs &a, hipStream_t s, bool counted = false);      // counted: the count ran in the force kernel's epilogue
FrCountArgs fused_count_args(const FusedArgs &a);
int fused_tile_codes();
int fused_gtile_codes();
int fused_direct_tiles();
int fused_super_tiles();

__device__ inline int dir_of_entry(const int *__restrict__ dir_start, int k)
{
    // dir_start[28]: exclusive offsets of each direction's segment in the send list
    int d = 0;
#pragma unroll
    for (int q = 1; q < 27; q++) d += (k >= dir_start[q]) ? 1 : 0;
    return d;
}

__host__ __device__ inline double min_image(double dr, double p)   // math_meso.h:148-152
{
    double ph = p * 0.5;
    return dr + (dr > -ph ? (dr < ph ? 0.0 : -p) : p);
}

// Bond forces (and energy) of one atom from its bond list; shared by k_bond (bond.hip) and by the force kernel's step-boundary
// epilogue (pair_ring.hip), so that both give the same bits: compiled uncontracted whatever the caller's setting.
// STYLE 0: harmonic (coefficient table [k][r0], gpu_bond_harmonic bond_harmonic_meso.cu:46-117); STYLE 1: FENE
// ([k][r0][epsilon][sigma]), bond_fene_meso.cu:82-147 == BondFENE::compute src/MOLECULE/bond_fene.cpp:48-124 with the
// warning/abort branches replaced by the clamp the reference's kernel applies.  cf: nbt + 1 entries per coefficient.
template <int STYLE, bool EV>
__device__ inline void bond_forces_of_atom(const float4 *__restrict__ coord4, const float4 c1, int n, const int *__restrict__ idx,
                                           const int *__restrict__ types, const double *cf, int nbt, double px, double py,
                                           double pz, double &fx, double &fy, double &fz, double &e)
{
#pragma clang fp contract(off)
    const double *k = cf, *r0 = cf + nbt + 1, *eps = cf + 2 * (nbt + 1), *sig = cf + 3 * (nbt + 1);
    fx = 0.0; fy = 0.0; fz = 0.0; e = 0.0;
    for (int b = 0; b < n; b++) {
        const int j = idx[b], type = types[b];
        const float4 c2 = coord4[j];
        if (STYLE == 0) {
            double dx = min_image((double)c2.x - (double)c1.x, px);
            double dy = min_image((double)c2.y - (double)c1.y, py);
            double dz = min_image((double)c2.z - (double)c1.z, pz);
            double rsq = dx * dx + dy * dy + dz * dz;
            double rinv = rsqrt(rsq);
            double r = rinv * rsq;
            double fbond = 2.0 * k[type] * (r - r0[type]) * rinv;
            fx += dx * fbond; fy += dy * fbond; fz += dz * fbond;
            if (EV) e += k[type] * (r - r0[type]) * (r - r0[type]);
        } else {
            double dx = min_image((double)c1.x - (double)c2.x, px);
            double dy = min_image((double)c1.y - (double)c2.y, py);
            double dz = min_image((double)c1.z - (double)c2.z, pz);
            double rsq = dx * dx + dy * dy + dz * dz;
            double r0sq = r0[type] * r0[type];
            double rlogarg = fmax(0.1, 1.0 - rsq / r0sq);
            double fbond = -k[type] / rlogarg;
            if (EV) e += -0.5 * k[type] * r0sq * log(rlogarg);
            const double s2 = sig[type] * sig[type];
            if (rsq < 1.25992104989487316477 * s2) {      // 2^(1/3) sigma^2: the WCA part
                double sr2 = s2 / rsq;
                double sr6 = sr2 * sr2 * sr2;
                fbond += 48.0 * eps[type] * sr6 * (sr6 - 0.5) / rsq;
                if (EV) e += 4.0 * eps[type] * sr6 * (sr6 - 1.0) + eps[type];
            }
            fx += dx * fbond; fy += dy * fbond; fz += dz * fbond;
        }
    }
}

} // namespace meso
