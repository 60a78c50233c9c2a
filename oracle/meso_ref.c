/* ---------------------------------------------------------------------------
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C CPU restatement of the USER-MESO *GPU* algorithm for the DPD hot path.
 * The reference's own sources for this path are CUDA (nvcc, inline PTX, texture
 * objects) and cannot be compiled in this image, and the reference ships no tests or
 * golden vectors for it, so:
 *
 *     PARITY UNPINNED against the reference's GPU path.
 *
 * What pins it instead (tests/test_oracle_meso.py):
 *   - at sigma=0 its forces agree with the golden-pinned stock-LAMMPS restatement
 *     (oracle/lmp_dpd_cpu.c) to the fp32-coordinate tolerance stated in the test;
 *   - the cipher core against the published known-answer vectors of TEA (32 cycles,
 *     zero key/block -> 41EA3A0A 94BAA940; key 00112233..ccddeeff, block 01234567
 *     89abcdef -> 126C6B92 C0653A3E), the polynomials against libm;
 *   - TEA/Gaussian invariants (symmetry, |xi|<=4, moments), sum_i F_i = 0,
 *     neighbour set == brute force.
 *
 * Each function cites the reference lines it restates.  All arithmetic is written
 * with explicit fma()/fmaf() exactly where the reference writes __fma_rn; everything
 * else is evaluated left to right with contraction disabled (-ffp-contract=off), and
 * the HIP kernels follow the same convention.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * ------------------------------------------------------------------------- */
#include <fenv.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t uint;

/* ---- bit helpers -------------------------------------------------------- */
static inline uint f2u(float f) { uint u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint u) { float f; memcpy(&f, &u, 4); return f; }
static inline double ll2d(int64_t i) { double d; memcpy(&d, &i, 8); return d; }
static inline int64_t d2ll(double d) { int64_t i; memcpy(&i, &d, 8); return i; }

static inline uint brev32(uint x)
{
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}

/* math_meso.h:166-178 */
static inline uint bit_space3(uint x)
{
    x = (x | (x << 12)) & 0X00FC003FU;
    x = (x | (x << 6)) & 0X381C0E07U;
    x = (x | (x << 4)) & 0X190C8643U;
    x = (x | (x << 2)) & 0X49249249U;
    return x;
}
static inline uint interleave3(uint i, uint j, uint k)
{
    return bit_space3(i) | (bit_space3(j) << 1) | (bit_space3(k) << 2);
}
uint meso_morton_encode(uint x, uint y, uint z) { return interleave3(x, y, z); }

/* math_meso.h:436-442 */
uint meso_mantissa(float u, float v, float w)
{
    uint i = f2u(u) & 0X7FF000U, j = f2u(v) & 0X7FF000U, k = f2u(w) & 0X7FF000U;
    return interleave3(i >> 12, j >> 12, k >> 12);
}

/* ---- TEA  (math_meso.h:444-464) ----------------------------------------- */
#define TEA_K0 0xA341316Cu
#define TEA_K1 0xC8013EA4u
#define TEA_K2 0xAD90777Du
#define TEA_K3 0x7E95761Eu
#define TEA_DT 0x9E3779B9u

/* The cipher core with the key as an argument: the reference's __TEA_core (math_meso.h:450-464) is the Tiny Encryption
 * Algorithm of Wheeler & Needham (1994) with a fixed key (math_meso.h:444-448) and a template round count; with the key
 * exposed the structure is pinned by the published known-answer vectors of TEA (tests/test_oracle_meso.py). */
void meso_tea_core_key(int rounds, const uint key[4], uint *pv0, uint *pv1)
{
    uint v0 = *pv0, v1 = *pv1, sum = 0;
    for (int n = 0; n < rounds; n++) {
        sum += TEA_DT;
        v0 += ((v1 << 4) + key[0]) ^ (v1 + sum) ^ ((v1 >> 5) + key[1]);
        v1 += ((v0 << 4) + key[2]) ^ (v0 + sum) ^ ((v0 >> 5) + key[3]);
    }
    *pv0 = v0; *pv1 = v1;
}

void meso_tea_core(int rounds, uint *pv0, uint *pv1)
{
    static const uint key[4] = {TEA_K0, TEA_K1, TEA_K2, TEA_K3};
    meso_tea_core_key(rounds, key, pv0, pv1);
}

uint meso_premix_tea(int rounds, uint v0, uint v1)
{
    meso_tea_core(rounds, &v0, &v1);
    return v0 ^ v1;
}

/* pair_dpd_meso.cu:268-270 : premix_TEA<64>( seed, update->ntimestep ) */
uint meso_seed_now(int seed, int64_t ntimestep) { return meso_premix_tea(64, (uint)seed, (uint)ntimestep); }

/* atom_vec_meso.cu:164 */
uint meso_signature(uint step_seed, int tag, float vx, float vy, float vz)
{
    return step_seed ^ meso_premix_tea(16, brev32((uint)tag), meso_mantissa(vx, vy, vz));
}

/* ---- fp64 polynomial math (math_meso.h:17-24, 204-424) ------------------- */
#define LN_2 6.9314718055994528623E-1
#define ONE_OVER_SQ2 7.0710678118654757274E-1
#define SQRT_2 1.4142135623730950488
#define TWO_TO_MINUS_31 4.6566128730773925781E-10
#define TWO_TO_MINUS_32 2.3283064365386962891E-10
#define EPSILON_SQ 1.0E-20

static inline double two_to_n(int n) { return ll2d(((int64_t)(1023 + n)) << 52); }

double meso_rsqrt(double x) /* :210-221 */
{
    double xrsqrt = ll2d(0X5FE660FCB5422422LL - (d2ll(x) >> 1));
    double x2m = x * -0.5;
    xrsqrt *= fma(xrsqrt * xrsqrt, x2m, 1.5);
    xrsqrt *= fma(xrsqrt * xrsqrt, x2m, 1.5);
    xrsqrt *= fma(xrsqrt * xrsqrt, x2m, 1.5);
    xrsqrt *= fma(xrsqrt * xrsqrt, x2m, 1.5);
    return xrsqrt;
}
double meso_sqrtd(double x) { return x * meso_rsqrt(x); } /* :223-226 */

double meso_rcp(double x) /* :230-238 */
{
    double xinv = ll2d(0X7FDE62361B1C4042LL - d2ll(x));
    xinv -= fma(x, xinv, -1.) * xinv;
    xinv -= fma(x, xinv, -1.) * xinv;
    xinv -= fma(x, xinv, -1.) * xinv;
    xinv -= fma(x, xinv, -1.) * xinv;
    return xinv;
}

static double log2d_frac(double x) /* :264-279 */
{
    int pred = x > SQRT_2;
    x *= pred ? 0.5 : ONE_OVER_SQ2;
    double z = (x - 1.) * meso_rcp(x + 1.);
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

static double exp2d_frac(double x) /* :308-323 */
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

double meso_powd(double a, double b) /* :332-344 */
{
    int64_t bits = d2ll(a);
    int hi = (int)(bits >> 32);
    uint lo = (uint)bits;
    double I = (hi >> 20) - 1023;
    int64_t fb = ((int64_t)((hi & 0X000FFFFF) | 0X3FF00000) << 32) | lo;
    double F = log2d_frac(ll2d(fb));
    double II = floor(b * (I + F));
    return two_to_n((int)II) * exp2d_frac(fma(b, F, fma(b, I, -II)));
}

double meso_cospi(double x) /* :380-391 */
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

double meso_log2u(uint x) /* :401-424 */
{
    int I = 31 - __builtin_clz(x);
    double xx = (double)x * two_to_n(-I);
    double exp = I - 32;
    int pred = xx > SQRT_2;
    xx *= pred ? 0.5 : ONE_OVER_SQ2;
    double z = (xx - 1.) * meso_rcp(xx + 1.);
    double y = z * z * 33.9705627484771406;
    double s = 2.55854634203511155E-7;
    s = fma(s, y, 1.05013262724846015E-5);
    s = fma(s, y, 5.00072802051539862E-4);
    s = fma(s, y, 2.83126505877817866E-2);
    s = fma(s, y, 2.88539008179006374E+0);
    return fma(z, s, (pred ? 1.0 : 0.5) + exp);
}

/* gaussian_TEA<4>  math_meso.h:466-474 ; u,v = the two signatures, order-independent */
double meso_gaussian_tea(uint u, uint v)
{
    int pred = u > v;
    uint v0 = pred ? u : v, v1 = !pred ? u : v;
    meso_tea_core(4, &v0, &v1);
    double f = meso_cospi((v0 & 0X7FFFFFFFu) * TWO_TO_MINUS_31) * ((v0 & 0X80000000u) ? 1.0 : -1.0);
    uint m = v1 > 1u ? v1 : 1u;
    double r = meso_sqrtd(-2.0 * LN_2 * meso_log2u(m));
    double g = r * f;
    return fmax(-4.0, fmin(g, 4.0));
}

/* gaussian_TEA_fast<4>  math_meso.h:476-484 (CUDA sinpif/log2f/sqrtf -> libm here) */
float meso_gaussian_tea_fast(uint u, uint v)
{
    int pred = u > v;
    uint v0 = pred ? u : v, v1 = !pred ? u : v;
    meso_tea_core(4, &v0, &v1);
    float t = (float)(int)v0 * (float)TWO_TO_MINUS_31;
    float f = (float)sin(3.14159265358979323846 * (double)t);
    float r = sqrtf(-2.0f * (float)LN_2 * log2f((float)v1 * (float)TWO_TO_MINUS_32));
    float g = r * f;
    return fmaxf(-4.0f, fminf(g, 4.0f));
}

/* mean0var1<8>  pair_dpd_minimal_meso.cu:50-89: the pair noise of pair_style dpd/mini/meso.  p = u/2^32 + v/2^32 - 1, four
 * rounds of r = FMA(FMA(8, x^2, -8), x^2, 1) with FMA = __fmaf_rz (round toward zero; x*x rounds to nearest), times sqrt 2.
 * volatile keeps gcc from moving the arithmetic across the rounding-mode switches. */
float meso_logistic_noise(uint u, uint v)
{
    volatile float x = (float)u / 4294967296.f + (float)v / 4294967296.f - 1.0f;
    for (int k = 0; k < 4; k++) {
        volatile float x2 = x * x;
        fesetround(FE_TOWARDZERO);
        volatile float t = fmaf(8.0f, x2, -8.0f);
        volatile float r = fmaf(t, x2, 1.0f);
        fesetround(FE_TONEAREST);
        x = r;
    }
    return x * 1.41421356237309514547f;
}

/* ---- gpu_merge_xvt  atom_vec_meso.cu:142-167 ----------------------------- */
/* coord4/veloc4: n x 4 floats; .w carries (type-1) / signature bit patterns */
void meso_merge_xvt(int n, const double *x, const double *y, const double *z, const double *vx,
                    const double *vy, const double *vz, const int *type, const int *tag, double cx,
                    double cy, double cz, uint seed, float *coord4, float *veloc4)
{
    for (int i = 0; i < n; i++) {
        coord4[4 * i + 0] = (float)(x[i] - cx);
        coord4[4 * i + 1] = (float)(y[i] - cy);
        coord4[4 * i + 2] = (float)(z[i] - cz);
        coord4[4 * i + 3] = u2f((uint)(type[i] - 1));
        float a = (float)vx[i], b = (float)vy[i], c = (float)vz[i];
        veloc4[4 * i + 0] = a; veloc4[4 * i + 1] = b; veloc4[4 * i + 2] = c;
        veloc4[4 * i + 3] = u2f(meso_signature(seed, tag[i], a, b, c));
    }
}

/* ---- full neighbour table from merged fp32 coordinates -------------------
 * Membership test of gpu_build_neighbor_list (neigh_build_meso.cu:85-109):
 *   dr2 = dx*dx + dy*dy + dz*dz in fp32, j != i, dr2 <= rc2_tail  (core and skin joined,
 *   gpu_join_neigh_list :166-200).  Rows come out sorted by j (canonical form for set
 *   comparison); bins follow gpu_assign_bin_id (neighbor_meso.cu:386-421) only in the
 *   sense that any correct cell search yields the same set.
 * table is row-major [nlocal][stride]; returns max row length (may exceed stride ->
 * overflow, rows truncated like the reference's sentinel case). */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

int meso_neigh_full(int nlocal, int nall, const float *coord4, float rc2_tail, int *count,
                    int *table, int stride)
{
    /* simple cell grid over the bounding box of all atoms */
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (int i = 0; i < nall; i++)
        for (int d = 0; d < 3; d++) {
            float c = coord4[4 * i + d];
            if (c < lo[d]) lo[d] = c;
            if (c > hi[d]) hi[d] = c;
        }
    double rc = sqrt((double)rc2_tail) * 1.0001;
    int nb[3];
    double inv[3];
    for (int d = 0; d < 3; d++) {
        nb[d] = (int)((hi[d] - lo[d]) / rc);
        if (nb[d] < 1) nb[d] = 1;
        inv[d] = nb[d] / ((double)(hi[d] - lo[d]) * 1.0000001 + 1e-30);
    }
    int nbins = nb[0] * nb[1] * nb[2];
    int *head = malloc(sizeof(int) * (nbins + 1)), *cell = malloc(sizeof(int) * nall),
        *order = malloc(sizeof(int) * nall);
    memset(head, 0, sizeof(int) * (nbins + 1));
    for (int i = 0; i < nall; i++) {
        int b[3];
        for (int d = 0; d < 3; d++) {
            b[d] = (int)((coord4[4 * i + d] - lo[d]) * inv[d]);
            if (b[d] >= nb[d]) b[d] = nb[d] - 1;
            if (b[d] < 0) b[d] = 0;
        }
        cell[i] = b[0] + nb[0] * (b[1] + nb[1] * b[2]);
        head[cell[i] + 1]++;
    }
    for (int b = 0; b < nbins; b++) head[b + 1] += head[b];
    int *fill = malloc(sizeof(int) * nbins);
    memcpy(fill, head, sizeof(int) * nbins);
    for (int i = 0; i < nall; i++) order[fill[cell[i]]++] = i;
    int maxlen = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(max : maxlen)
    for (int i = 0; i < nlocal; i++) {
        float xi = coord4[4 * i], yi = coord4[4 * i + 1], zi = coord4[4 * i + 2];
        int c = cell[i], bx = c % nb[0], by = (c / nb[0]) % nb[1], bz = c / (nb[0] * nb[1]);
        int n = 0;
        int *row = table + (size_t)i * stride;
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    int x2 = bx + dx, y2 = by + dy, z2 = bz + dz;
                    if (x2 < 0 || x2 >= nb[0] || y2 < 0 || y2 >= nb[1] || z2 < 0 || z2 >= nb[2]) continue;
                    int b = x2 + nb[0] * (y2 + nb[1] * z2);
                    for (int p = head[b]; p < head[b + 1]; p++) {
                        int j = order[p];
                        if (j == i) continue;
                        float ddx = xi - coord4[4 * j], ddy = yi - coord4[4 * j + 1], ddz = zi - coord4[4 * j + 2];
                        float dr2 = ddx * ddx + ddy * ddy + ddz * ddz;
                        if (dr2 <= rc2_tail) {
                            if (n < stride) row[n] = j;
                            n++;
                        }
                    }
                }
        count[i] = n < stride ? n : stride;
        qsort(row, count[i], sizeof(int), cmp_int);
        if (n > maxlen) maxlen = n;
    }
    free(head); free(cell); free(order); free(fill);
    return maxlen;
}

/* ---- gpu_dpd<evflag>  pair_dpd_meso.cu:91-205 ---------------------------- */
/* coeff: ntypes*ntypes*7 doubles {cut,cutsq,cutinv,expw,a0,gamma,sigma} (pair_dpd_meso.h:15-24)
 * table row-major [i*stride + p].  Forces are ADDED to fx,fy,fz (reference: += ).
 * ev (may be NULL): per-atom e_pair[n] and virial[6][n] as the reference stores them. */
void meso_pair_dpd(int ibeg, int iend, const float *coord4, const float *veloc4, const int *count,
                   const int *table, int stride, const double *coeff, int ntypes, double dt_inv_sqrt,
                   double *fx, double *fy, double *fz, double *e_pair, double *virial, int nvir_stride)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = ibeg; i < iend; i++) {
        float c1x = coord4[4 * i], c1y = coord4[4 * i + 1], c1z = coord4[4 * i + 2];
        uint t1 = f2u(coord4[4 * i + 3]);
        float v1x = veloc4[4 * i], v1y = veloc4[4 * i + 1], v1z = veloc4[4 * i + 2];
        uint s1 = f2u(veloc4[4 * i + 3]);
        double ax = 0., ay = 0., az = 0., energy = 0.;
        double vr[6] = {0, 0, 0, 0, 0, 0};
        for (int p = 0; p < count[i]; p++) {
            int j = table[(size_t)i * stride + p];
            double dx = (double)c1x - (double)coord4[4 * j];
            double dy = (double)c1y - (double)coord4[4 * j + 1];
            double dz = (double)c1z - (double)coord4[4 * j + 2];
            double rsq = dx * dx + dy * dy + dz * dz;
            const double *cf = coeff + (t1 * ntypes + f2u(coord4[4 * j + 3])) * 7;
            if (rsq < cf[1] && rsq >= EPSILON_SQ) {
                double rn = meso_gaussian_tea(s1, f2u(veloc4[4 * j + 3]));
                double rinv = 1.0 / sqrt(rsq); /* CUDA rsqrt(double) */
                double r = rsq * rinv;
                double dvx = (double)v1x - (double)veloc4[4 * j];
                double dvy = (double)v1y - (double)veloc4[4 * j + 1];
                double dvz = (double)v1z - (double)veloc4[4 * j + 2];
                double dot = dx * dvx + dy * dvy + dz * dvz;
                double wc = 1.0 - r * cf[2];
                double wr = meso_powd(wc, cf[3]);
                double fpair = cf[4] * wc - (cf[5] * wr * wr * dot * rinv) + (cf[6] * wr * rn * dt_inv_sqrt);
                fpair *= rinv;
                ax += dx * fpair; ay += dy * fpair; az += dz * fpair;
                if (e_pair) {
                    vr[0] += dx * dx * fpair; vr[1] += dy * dy * fpair; vr[2] += dz * dz * fpair;
                    vr[3] += dx * dy * fpair; vr[4] += dx * dz * fpair; vr[5] += dy * dz * fpair;
                    energy += 0.5 * cf[4] * cf[0] * wc * wc;
                }
            }
        }
        fx[i] += ax; fy[i] += ay; fz[i] += az;
        if (e_pair) {
            for (int k = 0; k < 6; k++) virial[(size_t)k * nvir_stride + i] += vr[k] * 0.5;
            e_pair[i] = energy * 0.5;
        }
    }
}

/* ---- gpu_dpd_fast<0>  pair_dpd_fast_meso.cu:91-205 ----------------------- */
/* rng 0: gaussian_TEA_fast (dpd/fast/meso); rng 1: mean0var1<8> (gpu_dpd_mini pair_dpd_minimal_meso.cu:91-178, same force) */
/* polyval  math_meso.h:53-58 (row = [order][c_order .. c_0]) */
static float polyval_row(float x, const float *row)
{
    int order = (int)row[0];
    float r = row[1];
    for (int k = 0; k < order; k++) r = r * x + row[2 + k];
    return r;
}

/* uniform_TEA_fast<4>  math_meso.h:501-505, arguments (min, max) as at its call site pair_dpd_tableforce_meso.cu:171 */
float meso_uniform_tea_fast(uint u, uint v)
{
    uint v0 = u < v ? u : v, v1 = u < v ? v : u;
    meso_tea_core(4, &v0, &v1);
    return (float)(v0 ^ v1) * (float)(1.73205080756887729353 * TWO_TO_MINUS_31) - (float)1.73205080756887729353;
}

/* the texture fetch of gpu_dpd_tableforce (:181; linear filter, clamp, coordinate transform :291): L points uniform in r/rc,
 * interpolation weight with 8 fractional bits (CUDA programming guide, linear filtering) */
static float table_force(float rrinv, const float *tab, int len)
{
    float x = rrinv * (float)(len - 1);
    if (x < 0.f) x = 0.f;
    int i = (int)x;
    if (i > len - 1) i = len - 1;
    int i1 = i + 1 < len ? i + 1 : len - 1;
    float al = floorf((x - (float)i) * 256.0f + 0.5f) * (1.0f / 256.0f);
    return (1.0f - al) * tab[i] + al * tab[i1];
}

/* poly != NULL: gpu_dpd_polyforce pair_dpd_polyforce_meso.cu:91-205 - conservative force polyval(1 - r/rc), rows of 33 floats */
void meso_pair_dpd_fast_rng(int ibeg, int iend, const float *coord4, const float *veloc4, const int *count,
                            const int *table, int stride, const float *coeff, int ntypes, float dt_inv_sqrt,
                            double *fx, double *fy, double *fz, int rng, const float *poly, const float *ftab, int ftab_len)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = ibeg; i < iend; i++) {
        float c1x = coord4[4 * i], c1y = coord4[4 * i + 1], c1z = coord4[4 * i + 2];
        uint t1 = f2u(coord4[4 * i + 3]);
        float v1x = veloc4[4 * i], v1y = veloc4[4 * i + 1], v1z = veloc4[4 * i + 2];
        uint s1 = f2u(veloc4[4 * i + 3]);
        float ax = 0.f, ay = 0.f, az = 0.f;
        for (int p = 0; p < count[i]; p++) {
            int j = table[(size_t)i * stride + p];
            float dx = c1x - coord4[4 * j], dy = c1y - coord4[4 * j + 1], dz = c1z - coord4[4 * j + 2];
            float rsq = dx * dx + dy * dy + dz * dz;
            const float *cf = coeff + (t1 * ntypes + f2u(coord4[4 * j + 3])) * 7;
            if (rsq < cf[1] && (double)rsq >= EPSILON_SQ) {
                uint s2 = f2u(veloc4[4 * j + 3]);
                float rn = rng == 1 ? meso_logistic_noise(s1, s2) : rng == 2 ? meso_uniform_tea_fast(s1, s2) : meso_gaussian_tea_fast(s1, s2);
                float rinv = 1.0f / sqrtf(rsq);
                float r = rsq * rinv;
                float dvx = v1x - veloc4[4 * j], dvy = v1y - veloc4[4 * j + 1], dvz = v1z - veloc4[4 * j + 2];
                float dot = dx * dvx + dy * dvy + dz * dvz;
                float wc = 1.0f - r * cf[2];
                float wr = powf(wc, cf[3]);
                float fc = poly ? polyval_row(wc, poly + (t1 * ntypes + f2u(coord4[4 * j + 3])) * 33) : cf[4] * wc;
                if (ftab) fc = table_force(r * cf[2], ftab + (t1 * ntypes + f2u(coord4[4 * j + 3])) * ftab_len, ftab_len);
                float fpair = fc - (cf[5] * wr * wr * dot * rinv) + (cf[6] * wr * rn * dt_inv_sqrt);
                fpair *= rinv;
                ax += dx * fpair; ay += dy * fpair; az += dz * fpair;
            }
        }
        fx[i] += ax; fy[i] += ay; fz[i] += az;
    }
}

void meso_pair_dpd_fast(int ibeg, int iend, const float *coord4, const float *veloc4, const int *count,
                        const int *table, int stride, const float *coeff, int ntypes, float dt_inv_sqrt,
                        double *fx, double *fy, double *fz)
{
    meso_pair_dpd_fast_rng(ibeg, iend, coord4, veloc4, count, table, stride, coeff, ntypes, dt_inv_sqrt, fx, fy, fz, 0, NULL, NULL, 0);
}

/* ---- fix nve/meso  fix_nve_meso.cu:62-95, 157-178 ------------------------ */
void meso_nve_initial(int n, double *x, double *y, double *z, double *vx, double *vy, double *vz,
                      const double *fx, const double *fy, const double *fz, const int *mask,
                      const double *mass, double dtf, double dtv, int groupbit)
{
    for (int i = 0; i < n; i++)
        if (mask[i] & groupbit) {
            double dtfm = dtf * meso_rcp(mass[i]);
            vx[i] += dtfm * fx[i]; vy[i] += dtfm * fy[i]; vz[i] += dtfm * fz[i];
            x[i] += dtv * vx[i]; y[i] += dtv * vy[i]; z[i] += dtv * vz[i];
        }
}

void meso_nve_final(int n, double *vx, double *vy, double *vz, const double *fx, const double *fy,
                    const double *fz, const int *mask, const double *mass, double dtf, int groupbit)
{
    for (int i = 0; i < n; i++)
        if (mask[i] & groupbit) {
            double dtfm = dtf * meso_rcp(mass[i]);
            vx[i] += dtfm * fx[i]; vy[i] += dtfm * fy[i]; vz[i] += dtfm * fz[i];
        }
}

/* compute temp/meso  compute_temp_meso.cu:58-101 : returns sum m v^2 (tfactor applied by caller) */
double meso_sum_mv2(int n, const double *vx, const double *vy, const double *vz, const double *mass,
                    const int *mask, int groupbit)
{
    double t = 0.0;
    for (int i = 0; i < n; i++)
        if (mask[i] & groupbit) t += mass[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
    return t;
}

/* ---- gpu_assign_bin_id  neighbor_meso.cu:386-421 ------------------------- */
static inline int clampi(int i, int nmin, int nmax) { int a = i < nmax - 1 ? i : nmax - 1; return a > nmin ? a : nmin; }

void meso_assign_bin_id(int n_local, int n_atom, const double *x, const double *y, const double *z,
                        const double *lo, const double *hi, const int *mbin, const double *bininv, uint *bin_id)
{
    for (int i = 0; i < n_atom; i++) {
        int bx = clampi((int)((x[i] - lo[0]) * bininv[0] + 1.0), 0, mbin[0]);
        int by = clampi((int)((y[i] - lo[1]) * bininv[1] + 1.0), 0, mbin[1]);
        int bz = clampi((int)((z[i] - lo[2]) * bininv[2] + 1.0), 0, mbin[2]);
        if (i >= n_local) {
            bx = (x[i] >= lo[0]) ? (x[i] <= hi[0] ? bx : mbin[0] - 1) : 0;
            by = (y[i] >= lo[1]) ? (y[i] <= hi[1] ? by : mbin[1] - 1) : 0;
            bz = (z[i] >= lo[2]) ? (z[i] <= hi[2] ? bz : mbin[2] - 1) : 0;
        }
        bin_id[i] = bx + mbin[0] * (by + bz * mbin[1]);
    }
}

/* ---- gpu_build_reorder_keypair<1>  atom_meso.cu:268-308, sort_local :343-384 ----
 * key = [border bit][Morton(bin)][Morton(16^3 sub-cell)].  NOTE the reference computes the
 * sub-cell offset as coord - (bin-1)*bin_size WITHOUT subtracting the sub-box origin
 * (atom_meso.cu:295-297); that only matters when sublo != 0 and only changes ordering, so the
 * restatement (and the HIP kernel) subtract the origin. */
void meso_reorder_key(int n, const double *x, const double *y, const double *z, const int *borderness,
                      const double *lo, const double *binsize, const int *mbin, uint64_t *key)
{
    int l2_resoln = 16;
    int max_bin = mbin[0] > mbin[1] ? mbin[0] : mbin[1];
    if (mbin[2] > max_bin) max_bin = mbin[2];
    int l1_width = 3 * (int)floor(log2(max_bin * 2.0));
    int l2_width = 12;
    uint64_t border_mask = 1ULL << (l1_width + l2_width);
    for (int i = 0; i < n; i++) {
        const double c[3] = {x[i], y[i], z[i]};
        uint b[3], s[3];
        for (int d = 0; d < 3; d++) {
            double bininv = 1.0 / binsize[d];
            b[d] = (uint)clampi((int)((c[d] - lo[d]) * bininv + 1), 0, mbin[d]);
            s[d] = (uint)clampi((int)((c[d] - lo[d] - ((double)b[d] - 1) * binsize[d]) * (l2_resoln * bininv)), 0, l2_resoln);
        }
        uint64_t z1 = interleave3(b[0], b[1], b[2]), z2 = interleave3(s[0], s[1], s[2]);
        key[i] = (z1 << l2_width) | z2;
        if (borderness[i]) key[i] |= border_mask;
    }
}
