// HIP kernels of the DPD hot path for gfx950 (CDNA4, wave64).  See kernels.h for the map to the
// reference kernels.  Compiled with -ffp-contract=off; fused ops are written as fma()/fmaf().
#include "kernels.h"
#include "meso_device.h"
#include <type_traits>

namespace meso {

static inline int nblk(long n, int b) { return (int)((n + b - 1) / b); }
static inline int capgrid(long n, int b, int cap = 256 * 8) { int g = nblk(n, b); return g < 1 ? 1 : (g > cap ? cap : g); }

// =========================================================================================
// atom kernels
// =========================================================================================

// gpu_merge_xvt (atom_vec_meso.cu:142-167): fp64 SoA -> float4 pair, recentred, signature in .w
__global__ void __launch_bounds__(256) k_merge_xvt(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, const double *__restrict__ vx,
                                                   const double *__restrict__ vy, const double *__restrict__ vz,
                                                   const int *__restrict__ type, const int *__restrict__ tag,
                                                   float4 *__restrict__ coord4, float4 *__restrict__ veloc4,
                                                   double cx, double cy, double cz, u32 seed, int beg, int end)
{
    for (int i = beg + blockDim.x * blockIdx.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) {
        float4 c;
        c.x = (float)(x[i] - cx);
        c.y = (float)(y[i] - cy);
        c.z = (float)(z[i] - cz);
        c.w = __uint_as_float((u32)(type[i] - 1));
        coord4[i] = c;
        float4 v;
        v.x = (float)vx[i];
        v.y = (float)vy[i];
        v.z = (float)vz[i];
        v.w = __uint_as_float(signature(seed, tag[i], v.x, v.y, v.z));
        veloc4[i] = v;
    }
}

void launch_merge_xvt(const AtomSoA &a, float4 *coord4, float4 *veloc4, double cx, double cy, double cz,
                      uint32_t seed, int beg, int end, hipStream_t s)
{
    if (end <= beg) return;
    hipLaunchKernelGGL(k_merge_xvt, dim3(capgrid(end - beg, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], a.v[0],
                       a.v[1], a.v[2], a.type, a.tag, coord4, veloc4, cx, cy, cz, seed, beg, end);
}

// gpu_fix_NVE_init_intgrate<0> (fix_nve_meso.cu:62-95)
__global__ void __launch_bounds__(256) k_nve_initial(double *__restrict__ x, double *__restrict__ y,
                                                     double *__restrict__ z, double *__restrict__ vx,
                                                     double *__restrict__ vy, double *__restrict__ vz,
                                                     const double *__restrict__ fx, const double *__restrict__ fy,
                                                     const double *__restrict__ fz, const int *__restrict__ mask,
                                                     const double *__restrict__ mass, double dtf, double dtv,
                                                     int groupbit, int n, const int *__restrict__ poison)
{
    if (poison && *poison) return;      // a rebuild of this interval reported an outgrown capacity: the state waits for its redo (Engine::run)
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (mask[i] & groupbit) {
            double dtfm = dtf * rcp_poly(mass[i]);
            double a = vx[i] + dtfm * fx[i], b = vy[i] + dtfm * fy[i], c = vz[i] + dtfm * fz[i];
            vx[i] = a; vy[i] = b; vz[i] = c;
            x[i] += dtv * a; y[i] += dtv * b; z[i] += dtv * c;
        }
    }
}

void launch_nve_initial(const AtomSoA &a, double dtf, double dtv, int groupbit, int n, hipStream_t s, const int *poison)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_nve_initial, dim3(capgrid(n, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], a.v[0], a.v[1],
                       a.v[2], a.f[0], a.f[1], a.f[2], a.mask, a.mass, dtf, dtv, groupbit, n, poison);
}

// gpu_fix_NVE_init_intgrate<0> + gpu_merge_xvt for the same atom in one pass (the first step of a run() that keeps the neighbour table:
// the two kernels in the same order on the same values - bit-identical - one launch and one reading of x, v less)
__global__ void __launch_bounds__(256) k_nve_initial_merge(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                                           double *__restrict__ vx, double *__restrict__ vy, double *__restrict__ vz,
                                                           const double *__restrict__ fx, const double *__restrict__ fy,
                                                           const double *__restrict__ fz, const int *__restrict__ mask,
                                                           const double *__restrict__ mass, const int *__restrict__ type,
                                                           const int *__restrict__ tag, float4 *__restrict__ coord4, float4 *__restrict__ veloc4,
                                                           double dtf, double dtv, int groupbit, double cx, double cy, double cz, u32 seed, int n,
                                                           const int *__restrict__ poison)
{
    if (poison && *poison) return;
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double px = x[i], py = y[i], pz = z[i], a = vx[i], b = vy[i], c = vz[i];
        if (mask[i] & groupbit) {
            double dtfm = dtf * rcp_poly(mass[i]);
            a = a + dtfm * fx[i]; b = b + dtfm * fy[i]; c = c + dtfm * fz[i];
            vx[i] = a; vy[i] = b; vz[i] = c;
            px += dtv * a; py += dtv * b; pz += dtv * c;
            x[i] = px; y[i] = py; z[i] = pz;
        }
        float4 cc;
        cc.x = (float)(px - cx); cc.y = (float)(py - cy); cc.z = (float)(pz - cz);
        cc.w = __uint_as_float((u32)(type[i] - 1));
        coord4[i] = cc;
        float4 v;
        v.x = (float)a; v.y = (float)b; v.z = (float)c;
        v.w = __uint_as_float(signature(seed, tag[i], v.x, v.y, v.z));
        veloc4[i] = v;
    }
}

void launch_nve_initial_merge(const AtomSoA &a, double dtf, double dtv, int groupbit, int n, float4 *coord4, float4 *veloc4, double cx,
                              double cy, double cz, uint32_t seed, hipStream_t s, const int *poison)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_nve_initial_merge, dim3(capgrid(n, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], a.v[0], a.v[1], a.v[2], a.f[0],
                       a.f[1], a.f[2], a.mask, a.mass, a.type, a.tag, coord4, veloc4, dtf, dtv, groupbit, cx, cy, cz, seed, n, poison);
}

// gpu_fix_NVE_final_integrate (fix_nve_meso.cu:157-178)
__global__ void __launch_bounds__(256) k_nve_final(double *__restrict__ vx, double *__restrict__ vy,
                                                   double *__restrict__ vz, const double *__restrict__ fx,
                                                   const double *__restrict__ fy, const double *__restrict__ fz,
                                                   const int *__restrict__ mask, const double *__restrict__ mass,
                                                   double dtf, int groupbit, int n, const int *__restrict__ poison)
{
    if (poison && *poison) return;
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (mask[i] & groupbit) {
            double dtfm = dtf * rcp_poly(mass[i]);
            vx[i] += dtfm * fx[i]; vy[i] += dtfm * fy[i]; vz[i] += dtfm * fz[i];
        }
    }
}

void launch_nve_final(const AtomSoA &a, double dtf, int groupbit, int n, hipStream_t s, const int *poison)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_nve_final, dim3(capgrid(n, 256)), dim3(256), 0, s, a.v[0], a.v[1], a.v[2], a.f[0], a.f[1],
                       a.f[2], a.mask, a.mass, dtf, groupbit, n, poison);
}

// Step boundary fused into one pass over the particles: final_integrate of step s, initial_integrate of step s+1
// and (when step s+1 keeps the neighbour table) gpu_merge_xvt for step s+1.  Same operations in the same order as
// the three separate kernels, so results are bit-identical; it saves two launches and re-reading v, f, x
// (56 -> ~30 us per step on the 64^3 box).
__global__ void __launch_bounds__(256) k_nve_boundary(NveArgs a, const double *__restrict__ fx, const double *__restrict__ fy,
                                                      const double *__restrict__ fz, int n, const int *__restrict__ poison)
{
    if (poison && *poison) return;
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        nve_boundary_atom(a, i, fx[i], fy[i], fz[i]);
}

void launch_nve_boundary(const AtomSoA &a, double dtf, double dtv, int groupbit, int n, int merge, float4 *coord4,
                         float4 *veloc4, double cx, double cy, double cz, uint32_t seed_next, hipStream_t s, const int *poison)
{
    if (n <= 0) return;
    NveArgs nv = make_nve_args(a, dtf, dtv, groupbit, merge, coord4, veloc4, cx, cy, cz, seed_next);
    hipLaunchKernelGGL(k_nve_boundary, dim3(capgrid(n, 256)), dim3(256), 0, s, nv, a.f[0], a.f[1], a.f[2], n, poison);
}

NveArgs make_nve_args(const AtomSoA &a, double dtf, double dtv, int groupbit, int merge, float4 *coord4_next,
                      float4 *veloc4_next, double cx, double cy, double cz, uint32_t seed_next)
{
    NveArgs nv;
    for (int d = 0; d < 3; d++) { nv.x[d] = a.x[d]; nv.v[d] = a.v[d]; }
    nv.mass = a.mass; nv.mask = a.mask; nv.tag = a.tag; nv.type = a.type;
    nv.mass_type = nullptr; nv.dtfm_type = nullptr;
    nv.dtf = dtf; nv.dtv = dtv; nv.groupbit = groupbit; nv.merge = merge;
    nv.coord4_next = coord4_next; nv.veloc4_next = veloc4_next;
    nv.cx = cx; nv.cy = cy; nv.cz = cz; nv.seed_next = seed_next;
    nv.img_cnt = nullptr; nv.img = nullptr; nv.img_first = nullptr; nv.img_shift = nullptr;
    nv.img_c4 = coord4_next; nv.img_v4 = veloc4_next; nv.img_vofs = nullptr; nv.img_center = nullptr;
    return nv;
}

// ---- deterministic two-stage block reductions (replace gpu_reduce_sum_host, math_meso.h:677-692)
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ inline double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

template <int OP>
__device__ inline double block_reduce(double v)
{
    __shared__ double sm[4];
    v = OP == 0 ? wave_sum(v) : wave_max(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        r = sm[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); k++) r = OP == 0 ? r + sm[k] : fmax(r, sm[k]);
    }
    __syncthreads();
    return r;
}

// gpu_eK_scalar (compute_temp_meso.cu:58-75) fused with the first reduction stage
__global__ void __launch_bounds__(256) k_sum_mv2(const double *__restrict__ vx, const double *__restrict__ vy,
                                                 const double *__restrict__ vz, const double *__restrict__ mass,
                                                 const int *__restrict__ mask, int groupbit, int n,
                                                 double *__restrict__ partial)
{
    double t = 0.0;
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (mask[i] & groupbit) t += mass[i] * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
    double r = block_reduce<0>(t);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <int OP>
__global__ void __launch_bounds__(256) k_reduce_final(const double *__restrict__ partial, int n,
                                                      double *__restrict__ out)
{
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) t = OP == 0 ? t + partial[i] : fmax(t, partial[i]);
    double r = block_reduce<OP>(t);
    if (threadIdx.x == 0) *out = r;
}

#define MESO_REDUCE_BLOCKS 512

void launch_sum_mv2(const AtomSoA &a, int groupbit, int n, double *partial, double *result, hipStream_t s)
{
    int g = capgrid(n, 256, MESO_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_sum_mv2, dim3(g), dim3(256), 0, s, a.v[0], a.v[1], a.v[2], a.mass, a.mask, groupbit, n,
                       partial);
    hipLaunchKernelGGL(k_reduce_final<0>, dim3(1), dim3(256), 0, s, partial, g, result);
}

// Neighbor::check_distance equivalent: max squared displacement since the last build
__global__ void __launch_bounds__(256) k_max_disp2(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, const double *__restrict__ xhold,
                                                   int n, int stride, double *__restrict__ partial)
{
    double t = 0.0;
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double dx = x[i] - xhold[i], dy = y[i] - xhold[stride + i], dz = z[i] - xhold[2 * stride + i];
        t = fmax(t, dx * dx + dy * dy + dz * dz);
    }
    double r = block_reduce<1>(t);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

void launch_max_disp2(const AtomSoA &a, const double *xhold, int n, int stride, double *partial, double *result,
                      hipStream_t s)
{
    int g = capgrid(n, 256, MESO_REDUCE_BLOCKS);
    hipLaunchKernelGGL(k_max_disp2, dim3(g), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], xhold, n, stride, partial);
    hipLaunchKernelGGL(k_reduce_final<1>, dim3(1), dim3(256), 0, s, partial, g, result);
}

__global__ void __launch_bounds__(256) k_copy_hold(const double *__restrict__ x, const double *__restrict__ y,
                                                   const double *__restrict__ z, double *__restrict__ xhold, int n,
                                                   int stride)
{
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        xhold[i] = x[i]; xhold[stride + i] = y[i]; xhold[2 * stride + i] = z[i];
    }
}

void launch_copy_hold(const AtomSoA &a, double *xhold, int n, int stride, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_copy_hold, dim3(capgrid(n, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], xhold, n, stride);
}

// MesoDomain::pbc (domain_meso.cu:30-145): wrap into [lo,hi), update image flags (10 bits/dim)
__global__ void __launch_bounds__(256) k_pbc(double *__restrict__ x, double *__restrict__ y, double *__restrict__ z,
                                             int *__restrict__ image, double lox, double loy, double loz, double hix,
                                             double hiy, double hiz, int px, int py, int pz, int n)
{
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int img = image[i];
        int ix = img & 1023, iy = (img >> 10) & 1023, iz = img >> 20;
        if (px) {
            double c = x[i], p = hix - lox;
            if (c < lox) { c += p; ix = (ix - 1) & 1023; }
            if (c >= hix) { c -= p; c = fmax(c, lox); ix = (ix + 1) & 1023; }
            x[i] = c;
        }
        if (py) {
            double c = y[i], p = hiy - loy;
            if (c < loy) { c += p; iy = (iy - 1) & 1023; }
            if (c >= hiy) { c -= p; c = fmax(c, loy); iy = (iy + 1) & 1023; }
            y[i] = c;
        }
        if (pz) {
            double c = z[i], p = hiz - loz;
            if (c < loz) { c += p; iz = (iz - 1) & 1023; }
            if (c >= hiz) { c -= p; c = fmax(c, loz); iz = (iz + 1) & 1023; }
            z[i] = c;
        }
        image[i] = ix | (iy << 10) | (iz << 20);
    }
}

void launch_pbc(const AtomSoA &a, const double *lo, const double *hi, const int *per, int n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pbc, dim3(capgrid(n, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], a.image, lo[0], lo[1],
                       lo[2], hi[0], hi[1], hi[2], per[0], per[1], per[2], n);
}

// bandwidth probe: the float4 copy whose rate is quoted beside the nominal HBM peak in bench.py's roofline (four
// independent 16-byte loads per lane in flight, one workgroup per 16 KiB)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy_f4(const float4 *__restrict__ src4, float4 *__restrict__ dst4, size_t n)
{
    const f32x4 *src = (const f32x4 *)src4;
    f32x4 *dst = (f32x4 *)dst4;
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    if (base + 768 < n) {
        const f32x4 a = __builtin_nontemporal_load(src + base), b = __builtin_nontemporal_load(src + base + 256),
                    c = __builtin_nontemporal_load(src + base + 512), d = __builtin_nontemporal_load(src + base + 768);
        __builtin_nontemporal_store(a, dst + base); __builtin_nontemporal_store(b, dst + base + 256);
        __builtin_nontemporal_store(c, dst + base + 512); __builtin_nontemporal_store(d, dst + base + 768);
    } else {
        for (size_t i = base; i < n; i += 256) dst[i] = src[i];
    }
}
void launch_copy_f4(const float4 *src, float4 *dst, size_t n, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_copy_f4, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, s, src, dst, n);
}

template <typename T>
__global__ void __launch_bounds__(256) k_fill(T *__restrict__ p, T val, int n)
{
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = val;
}
void launch_fill_f64(double *p, double val, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_fill<double>, dim3(capgrid(n, 256)), dim3(256), 0, s, p, val, n);
}
void launch_fill_i32(int *p, int val, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_fill<int>, dim3(capgrid(n, 256)), dim3(256), 0, s, p, val, n);
}

// gpu_unpack_by_type (atom_vec_meso.h:90)
__global__ void __launch_bounds__(256) k_unpack_mass(const int *__restrict__ type, const double *__restrict__ mt,
                                                     double *__restrict__ mass, int beg, int end)
{
    for (int i = beg + blockDim.x * blockIdx.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x)
        mass[i] = mt[type[i]];
}
// dtf / m per type, with the expression of the step boundary (nve_boundary_atom: dtf * rcp_poly(m)) - the same bits, computed once
__global__ void __launch_bounds__(64) k_dtfm_table(const double *__restrict__ mt, int ntypes, double dtf, double *__restrict__ out)
{
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    if (t <= ntypes) out[t] = t > 0 ? dtf * rcp_poly(mt[t]) : 0.0;
}
void launch_dtfm_table(const double *mass_type, int ntypes, double dtf, double *dtfm_type, hipStream_t s)
{
    hipLaunchKernelGGL(k_dtfm_table, dim3((ntypes + 64) / 64), dim3(64), 0, s, mass_type, ntypes, dtf, dtfm_type);
}
void launch_unpack_mass(const int *type, const double *mass_type, int, double *mass, int beg, int end, hipStream_t s)
{
    if (end > beg)
        hipLaunchKernelGGL(k_unpack_mass, dim3(capgrid(end - beg, 256)), dim3(256), 0, s, type, mass_type, mass, beg, end);
}

// =========================================================================================
// reorder
// =========================================================================================
static int reorder_l1(const BinGeom &g)
{
    int max_bin = g.mbin[0] > g.mbin[1] ? g.mbin[0] : g.mbin[1];
    if (g.mbin[2] > max_bin) max_bin = g.mbin[2];
    int l1 = 0;
    while ((1 << (l1 + 1)) <= max_bin * 2) l1++;   // floor(log2(2*max_bin)), as sort_local (atom_meso.cu:354)
    return l1;
}

// Key = [border][Morton(bin): 3*l1 bits][Morton(sub-cell): sb bits].  The reference always spends 12 bits on a
// 16^3 sub-cell grid and sorts 64-bit keys; here the sub-cell resolution shrinks (16^3, 8^3, 4^3 ...) so the
// whole key fits 32 bits and rocPRIM's one-sweep radix sort applies.
int reorder_sub_bits(const BinGeom &g)
{
    int room = 32 - 1 - 3 * reorder_l1(g);
    int sb = room >= 12 ? 12 : (room / 3) * 3;
    return sb < 0 ? 0 : sb;
}

int reorder_key_bits(const BinGeom &g) { return 1 + 3 * reorder_l1(g) + reorder_sub_bits(g); }

// gpu_build_reorder_keypair<1> (atom_meso.cu:268-308) with borderness (comm_meso.cu:188-254) computed in place
__global__ void __launch_bounds__(256) k_reorder_keys(const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, BinGeom g, double slx, double sly,
                                                      double slz, double shx, double shy, double shz, int border_bit,
                                                      int sub_bits, u32 *__restrict__ key, int *__restrict__ val, int n)
{
    int i = blockDim.x * blockIdx.x + threadIdx.x;
    if (i >= n) return;
    const double c[3] = {x[i], y[i], z[i]};
    const int res = 1 << (sub_bits / 3);
    u32 b[3], sc[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        b[d] = (u32)clampi((int)((c[d] - g.lo[d]) * g.bininv[d] + 1), 0, g.mbin[d]);
        sc[d] = (u32)clampi((int)((c[d] - g.lo[d] - ((double)b[d] - 1) * g.binsize[d]) * (res * g.bininv[d])), 0, res);
    }
    u32 k = (interleave3(b[0], b[1], b[2]) << sub_bits) | interleave3(sc[0], sc[1], sc[2]);
    bool border = c[0] <= slx || c[0] >= shx || c[1] <= sly || c[1] >= shy || c[2] <= slz || c[2] >= shz;
    if (border) k |= (1u << border_bit);
    key[i] = k;
    val[i] = i;
}

// ---- reorder without a comparison sort (rocPRIM sends <= 2^20 pairs through ~26 merge-sort launches, 134 us at 64^3):
// the key's upper part - [border][Morton(bin)], the "extended code" - has only 2M values, so the atoms are COUNTED per code
// (the atomic's return value is the atom's rank inside its code), the counts are scanned into estart (which the list
// builder needs anyway), every atom is placed at estart[code] + rank, and one pass per group of 128 codes puts the few
// atoms of each code in (sub-cell key, old index) order in LDS - the order the sort gave, so storage stays deterministic.
// wrap != null: MesoDomain::pbc (k_pbc) folded in - one rank has no migration between the wrap and the reorder, so the
// coordinates are wrapped here, on their way to the key (written back only where they changed)
struct WrapArgs { double *x, *y, *z; int *image; double lo[3], hi[3]; int per[3]; };
__global__ void __launch_bounds__(256) k_reorder_keys_count(const double *__restrict__ x, const double *__restrict__ y,
                                                            const double *__restrict__ z, BinGeom g, double slx, double sly,
                                                            double slz, double shx, double shy, double shz, int border_bit,
                                                            int sub_bits, u32 *__restrict__ key, int *__restrict__ rank,
                                                            int *__restrict__ cnt, int n, WrapArgs wr)
{
    int i = blockDim.x * blockIdx.x + threadIdx.x;
    const bool valid = i < n;
    u32 k = 0;
    if (valid) {
        double c[3] = {x[i], y[i], z[i]};
        if (wr.image) {
            const int img = wr.image[i];
            int im[3] = {img & 1023, (img >> 10) & 1023, img >> 20};
            bool moved = false;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                if (!wr.per[d]) continue;
                const double p = wr.hi[d] - wr.lo[d];
                if (c[d] < wr.lo[d]) { c[d] += p; im[d] = (im[d] - 1) & 1023; moved = true; }
                if (c[d] >= wr.hi[d]) { c[d] -= p; c[d] = fmax(c[d], wr.lo[d]); im[d] = (im[d] + 1) & 1023; moved = true; }
            }
            if (moved) {
                wr.x[i] = c[0]; wr.y[i] = c[1]; wr.z[i] = c[2];
                wr.image[i] = im[0] | (im[1] << 10) | (im[2] << 20);
            }
        }
        const int res = 1 << (sub_bits / 3);
        u32 b[3], sc[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            b[d] = (u32)clampi((int)((c[d] - g.lo[d]) * g.bininv[d] + 1), 0, g.mbin[d]);
            sc[d] = (u32)clampi((int)((c[d] - g.lo[d] - ((double)b[d] - 1) * g.binsize[d]) * (res * g.bininv[d])), 0, res);
        }
        k = (interleave3(b[0], b[1], b[2]) << sub_bits) | interleave3(sc[0], sc[1], sc[2]);
        bool border = c[0] <= slx || c[0] >= shx || c[1] <= sly || c[1] >= shy || c[2] <= slz || c[2] >= shz;
        if (border) k |= (1u << border_bit);
        key[i] = k;
    }
    const int r = run_rank(k >> sub_bits, valid, cnt);
    if (valid) rank[i] = r;
}
__global__ void __launch_bounds__(256) k_reorder_place(const u32 *__restrict__ key, const int *__restrict__ rank,
                                                       const int *__restrict__ estart, int sub_bits, int n,
                                                       int *__restrict__ placed, const int *__restrict__ n_dev)
{
    // n_dev: n only sized the grid (an estimate); the count is on the device and the loop covers whatever it turns out to be
    if (n_dev) n = *n_dev;
    for (int i = blockDim.x * blockIdx.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) placed[estart[key[i] >> sub_bits] + rank[i]] = i;
}
#define REORDER_CODES 128
__global__ void __launch_bounds__(REORDER_CODES) k_reorder_order(const int *__restrict__ estart, int ncodes,
                                                                 const u32 *__restrict__ key, const int *__restrict__ placed,
                                                                 int cap, int *__restrict__ val_sorted,
                                                                 u32 *__restrict__ key_sorted, int *__restrict__ inverse,
                                                                 int *__restrict__ cnt)
{
    extern __shared__ unsigned long long pairs[];      // (key << 32) | old index: one compare orders by key, then index
    const int c0 = blockIdx.x * REORDER_CODES, c = c0 + (int)threadIdx.x;
    const int c1 = min(c0 + REORDER_CODES, ncodes);
    const int s0 = estart[c0], s1 = estart[c1];
    const int ns = s1 - s0;
    if (cnt && c <= ncodes && (c < c1 || c == ncodes)) cnt[c] = 0;      // the counts are clean again for the next rebuild
    if (ns <= 0) return;
    const bool staged = ns <= cap;
    if (staged) {
        for (int p = threadIdx.x; p < ns; p += REORDER_CODES) {
            const int idx = placed[s0 + p];
            pairs[p] = ((unsigned long long)key[idx] << 32) | (u32)idx;
        }
    }
    __syncthreads();
    if (c < ncodes) {
        const int b = estart[c] - s0, e = estart[c + 1] - s0;
        if (staged) {
            for (int a = b + 1; a < e; a++) {
                const unsigned long long v = pairs[a];
                int q = a - 1;
                while (q >= b && pairs[q] > v) { pairs[q + 1] = pairs[q]; q--; }
                pairs[q + 1] = v;
            }
        } else {
            // a neighbourhood too dense for the LDS stage: selection into the output, straight from global memory
            for (int a = b; a < e; a++) {
                unsigned long long best = ~0ull;
                // the a-th smallest pair of the segment: smallest pair greater than the previous pick
                const unsigned long long prev = a == b ? 0ull : (((unsigned long long)key_sorted[s0 + a - 1] << 32) | (u32)val_sorted[s0 + a - 1]);
                for (int q = b; q < e; q++) {
                    const int idx = placed[s0 + q];
                    const unsigned long long v = ((unsigned long long)key[idx] << 32) | (u32)idx;
                    if ((a == b || v > prev) && v < best) best = v;
                }
                val_sorted[s0 + a] = (int)(u32)best;
                key_sorted[s0 + a] = (u32)(best >> 32);
                if (inverse) inverse[(u32)best] = s0 + a;
            }
        }
    }
    __syncthreads();
    if (staged) {
        for (int p = threadIdx.x; p < ns; p += REORDER_CODES) {
            const unsigned long long v = pairs[p];
            val_sorted[s0 + p] = (int)(u32)v;
            key_sorted[s0 + p] = (u32)(v >> 32);
            if (inverse) inverse[(u32)v] = s0 + p;
        }
    }
}
void launch_reorder_count(const AtomSoA &a, const BinGeom &g, const double *slab_lo, const double *slab_hi, uint32_t *key,
                          int *rank, int *cnt, int n, const double *wrap_lo, const double *wrap_hi, const int *wrap_per,
                          hipStream_t s)
{
    if (n <= 0) return;
    int bits = reorder_key_bits(g);
    WrapArgs wr = {nullptr, nullptr, nullptr, nullptr, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (wrap_lo) {
        wr.x = a.x[0]; wr.y = a.x[1]; wr.z = a.x[2]; wr.image = a.image;
        for (int d = 0; d < 3; d++) { wr.lo[d] = wrap_lo[d]; wr.hi[d] = wrap_hi[d]; wr.per[d] = wrap_per[d]; }
    }
    hipLaunchKernelGGL(k_reorder_keys_count, dim3(nblk(n, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], g, slab_lo[0],
                       slab_lo[1], slab_lo[2], slab_hi[0], slab_hi[1], slab_hi[2], bits - 1, reorder_sub_bits(g), key, rank,
                       cnt, n, wr);
}
void launch_reorder_place(const uint32_t *key, const int *rank, const int *estart, const BinGeom &g, int ncodes, int n, int cap,
                          int *placed, int *val_sorted, uint32_t *key_sorted, int *cnt, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_reorder_place, dim3(nblk(n, 256)), dim3(256), 0, s, key, rank, estart, reorder_sub_bits(g), n, placed,
                       (const int *)nullptr);
    const size_t dyn = (size_t)cap * 8;
    if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_reorder_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipLaunchKernelGGL(k_reorder_order, dim3((ncodes + REORDER_CODES) / REORDER_CODES), dim3(REORDER_CODES), dyn, s, estart,
                       ncodes, key, placed, cap, val_sorted, key_sorted, (int *)nullptr, cnt);
}

// ghosts: same scheme on the plain Morton code (rank from k_ghost_count, brick.hip); the ordering pass sorts the ghosts of a
// code by ghost index and writes gslot (ghost -> slot) as the inverse
void launch_ghost_order(const uint32_t *code, const int *rank, const int *gstart, int M, int nghost, int cap, int *placed,
                        int *slotval, uint32_t *code_sorted, int *gslot, int *cnt, const int *nghost_dev, hipStream_t s)
{
    if (nghost > 0)
        hipLaunchKernelGGL(k_reorder_place, dim3(nblk(nghost, 256)), dim3(256), 0, s, code, rank, gstart, 0, nghost, placed, nghost_dev);
    const size_t dyn = (size_t)cap * 8;
    if (dyn > 48 * 1024) (void)hipFuncSetAttribute((const void *)k_reorder_order, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    hipLaunchKernelGGL(k_reorder_order, dim3((M + REORDER_CODES) / REORDER_CODES), dim3(REORDER_CODES), dyn, s, gstart, M, code,
                       placed, cap, slotval, code_sorted, gslot, cnt);
}

void launch_reorder_keys(const AtomSoA &a, const BinGeom &g, const double *slab_lo, const double *slab_hi, const int *,
                         uint32_t *key, int *val, int n, hipStream_t s)
{
    if (n <= 0) return;
    int bits = reorder_key_bits(g);
    hipLaunchKernelGGL(k_reorder_keys, dim3(nblk(n, 256)), dim3(256), 0, s, a.x[0], a.x[1], a.x[2], g, slab_lo[0],
                       slab_lo[1], slab_lo[2], slab_hi[0], slab_hi[1], slab_hi[2], bits - 1, reorder_sub_bits(g), key, val,
                       n);
}

// gpu_permute_copy / gpu_deinterleave with permutation (atom_vec_meso.h:11-67): device-resident gather
// mg.coord4 != null: the merged float4 pair of the atom's new place is written as well (gpu_merge_xvt folded into the gather:
// the reorder has x, v, tag and type in registers anyway; 17 us of re-reading them at 64^3)
__global__ void __launch_bounds__(256) k_permute_atoms(AtomSoA src, AtomSoA dst, const int *__restrict__ from, int n, int with_f,
                                                       MergeOut mg)
{
    int i = blockDim.x * blockIdx.x + threadIdx.x;
    if (i >= n) return;
    permute_one(src, dst, from[i], i, with_f, mg);
}
void launch_permute_atoms(const AtomSoA &src, const AtomSoA &dst, const int *perm_from, int n, int with_f, hipStream_t s)
{
    MergeOut mg = {nullptr, nullptr, 0.0, 0.0, 0.0, 0u, nullptr, nullptr};
    if (n > 0) hipLaunchKernelGGL(k_permute_atoms, dim3(nblk(n, 256)), dim3(256), 0, s, src, dst, perm_from, n, with_f, mg);
}
void launch_permute_merge(const AtomSoA &src, const AtomSoA &dst, const int *perm_from, int n, int with_f, float4 *coord4,
                          float4 *veloc4, double cx, double cy, double cz, uint32_t seed, int *inverse, int *zero, hipStream_t s)
{
    MergeOut mg = {coord4, veloc4, cx, cy, cz, seed, inverse, zero};
    if (n > 0) hipLaunchKernelGGL(k_permute_atoms, dim3(nblk(n, 256)), dim3(256), 0, s, src, dst, perm_from, n, with_f, mg);
}

// gpu_permute_from2to (atom_meso.cu:310-314)
__global__ void __launch_bounds__(256) k_invert_perm(const int *__restrict__ A, int *__restrict__ T, int n)
{
    int i = blockDim.x * blockIdx.x + threadIdx.x;
    if (i < n) T[A[i]] = i;
}
void launch_invert_perm(const int *perm_from, int *perm_to, int n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_invert_perm, dim3(nblk(n, 256)), dim3(256), 0, s, perm_from, perm_to, n);
}

// =========================================================================================
// halo: border lists (deterministic two-pass compaction, no global atomics) + pack kernels
// =========================================================================================
#define BORDER_CHUNK 256

__global__ void __launch_bounds__(BORDER_CHUNK) k_border_count(const double *__restrict__ x,
                                                               const double *__restrict__ y,
                                                               const double *__restrict__ z, int beg, int end,
                                                               Slabs sl, int *__restrict__ chunk_count, int nchunk, int *__restrict__ clear)
{
    // per-wave ballots for all 27 directions, ONE barrier, then 27 threads add the wave totals
    __shared__ int wave_tot[27][BORDER_CHUNK / 64];
    if (clear && blockIdx.x == 0 && threadIdx.x == 0) *clear = 0;      // (a word the previous kernels are done with: no memset launch)
    int i = beg + blockIdx.x * BORDER_CHUNK + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int flags = 0;
    if (i < end) flags = near_flags(x[i], y[i], z[i], sl.lo, sl.hi);
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        const u64 m = __ballot(dir != 13 && flags && in_dir(flags, dir));
        if (lane == 0) wave_tot[dir][w] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        int c = 0;
        for (int k = 0; k < BORDER_CHUNK / 64; k++) c += wave_tot[threadIdx.x][k];
        chunk_count[threadIdx.x * nchunk + blockIdx.x] = c;
    }
}

// same two-pass compaction keyed by a per-atom direction code (migration: code 13 = the atom stays)
__global__ void __launch_bounds__(BORDER_CHUNK) k_code_count(const int *__restrict__ code, int beg, int end,
                                                             int *__restrict__ chunk_count, int nchunk)
{
    __shared__ int wave_tot[27][BORDER_CHUNK / 64];
    int i = beg + blockIdx.x * BORDER_CHUNK + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int c0 = i < end ? code[i] : -1;
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        const u64 m = __ballot(c0 == dir);
        if (lane == 0) wave_tot[dir][w] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        int c = 0;
        for (int k = 0; k < BORDER_CHUNK / 64; k++) c += wave_tot[threadIdx.x][k];
        chunk_count[threadIdx.x * nchunk + blockIdx.x] = c;
    }
}

__global__ void __launch_bounds__(BORDER_CHUNK) k_code_fill(const int *__restrict__ code, int beg, int end,
                                                            const int *__restrict__ chunk_offset, int nchunk,
                                                            int *__restrict__ list)
{
    __shared__ int wave_tot[27][BORDER_CHUNK / 64];
    int i = beg + blockIdx.x * BORDER_CHUNK + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int c0 = i < end ? code[i] : -1;
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        const u64 m = __ballot(c0 == dir);
        if (lane == 0) wave_tot[dir][w] = __popcll(m);
    }
    __syncthreads();
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        const bool hit = c0 == dir;
        const u64 m = __ballot(hit);
        if (!m) continue;
        int base = chunk_offset[dir * nchunk + blockIdx.x];
        for (int k = 0; k < w; k++) base += wave_tot[dir][k];
        if (hit) list[base + __popcll(m & ((1ULL << lane) - 1ULL))] = i;
    }
}

void launch_border_count_code(const int *code, int beg, int end, int *chunk_count, int nchunk, hipStream_t s)
{
    if (nchunk > 0) hipLaunchKernelGGL(k_code_count, dim3(nchunk), dim3(BORDER_CHUNK), 0, s, code, beg, end, chunk_count, nchunk);
}
void launch_border_fill_code(const int *code, int beg, int end, const int *chunk_offset, int nchunk, int *list, hipStream_t s)
{
    if (nchunk > 0) hipLaunchKernelGGL(k_code_fill, dim3(nchunk), dim3(BORDER_CHUNK), 0, s, code, beg, end, chunk_offset, nchunk, list);
}

void launch_border_count(const AtomSoA &a, int beg, int end, const double *slab_lo, const double *slab_hi, const int *,
                         int *chunk_count, int nchunk, hipStream_t s, int *clear)
{
    if (nchunk <= 0) {
        if (clear) (void)hipMemsetAsync(clear, 0, sizeof(int), s);
        return;
    }
    Slabs sl;
    for (int d = 0; d < 3; d++) { sl.lo[d] = slab_lo[d]; sl.hi[d] = slab_hi[d]; }
    hipLaunchKernelGGL(k_border_count, dim3(nchunk), dim3(BORDER_CHUNK), 0, s, a.x[0], a.x[1], a.x[2], beg, end, sl,
                       chunk_count, nchunk, clear);
}

__global__ void __launch_bounds__(BORDER_CHUNK) k_border_fill(const double *__restrict__ x,
                                                              const double *__restrict__ y,
                                                              const double *__restrict__ z, int beg, int end, Slabs sl,
                                                              const int *__restrict__ chunk_offset, int nchunk,
                                                              int *__restrict__ sendlist)
{
    __shared__ int wave_tot[27][BORDER_CHUNK / 64];
    int i = beg + blockIdx.x * BORDER_CHUNK + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int flags = 0;
    if (i < end) flags = near_flags(x[i], y[i], z[i], sl.lo, sl.hi);
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        const u64 m = __ballot(dir != 13 && flags && in_dir(flags, dir));
        if (lane == 0) wave_tot[dir][w] = __popcll(m);
    }
    __syncthreads();                        // one barrier for all directions
#pragma unroll 1
    for (int dir = 0; dir < 27; dir++) {
        if (dir == 13) continue;
        const bool hit = flags && in_dir(flags, dir);
        const u64 m = __ballot(hit);
        if (!m) continue;
        int base = chunk_offset[dir * nchunk + blockIdx.x];
        for (int k = 0; k < w; k++) base += wave_tot[dir][k];
        if (hit) sendlist[base + __popcll(m & ((1ULL << lane) - 1ULL))] = i;
    }
}

void launch_border_fill(const AtomSoA &a, int beg, int end, const double *slab_lo, const double *slab_hi, const int *,
                        const int *chunk_offset, int nchunk, int *sendlist, hipStream_t s)
{
    if (nchunk <= 0) return;
    Slabs sl;
    for (int d = 0; d < 3; d++) { sl.lo[d] = slab_lo[d]; sl.hi[d] = slab_hi[d]; }
    hipLaunchKernelGGL(k_border_fill, dim3(nchunk), dim3(BORDER_CHUNK), 0, s, a.x[0], a.x[1], a.x[2], beg, end, sl,
                       chunk_offset, nchunk, sendlist);
}

// dir_start[d] = chunk_offset[d*nchunk] (d = 0..27): one launch instead of 28 tiny device-to-device copies
__global__ void k_dir_starts(const int *__restrict__ chunk_offset, int nchunk, int *__restrict__ dir_start)
{
    int d = threadIdx.x;
    if (d < 28) dir_start[d] = chunk_offset[(size_t)d * nchunk];
}
void launch_dir_starts(const int *chunk_offset, int nchunk, int *dir_start, hipStream_t s)
{
    hipLaunchKernelGGL(k_dir_starts, dim3(1), dim3(64), 0, s, chunk_offset, nchunk, dir_start);
}
// ... and the checks of a rebuild whose counts stay on the device: the send list must fit the launch bound the host chose,
// and the border section must start at or behind the first atom the border scan looked at; flags[0] reports a violation
__global__ void k_dir_starts_check(const int *__restrict__ chunk_offset, int nchunk, int *__restrict__ dir_start, const int *__restrict__ n_bulk,
                                   int scan_beg, int bound, int *__restrict__ flags)
{
    int d = threadIdx.x;
    if (d < 28) dir_start[d] = chunk_offset[(size_t)d * nchunk];
    if (d == 0) {
        const int tot = chunk_offset[(size_t)27 * nchunk];
        if (tot > bound) flags[0] = 200000;
        else if (*n_bulk < scan_beg) flags[0] = 200001;
    }
}
void launch_dir_starts_check(const int *chunk_offset, int nchunk, int *dir_start, const int *n_bulk, int scan_beg, int bound, int *flags,
                             hipStream_t s)
{
    hipLaunchKernelGGL(k_dir_starts_check, dim3(1), dim3(64), 0, s, chunk_offset, nchunk, dir_start, n_bulk, scan_beg, bound, flags);
}

// One launch for the tail of the border pass when its counts fit one workgroup (27 x nchunk <= 64 Ki): exclusive scan of the
// chunk counts (tiles of 4096, carry in a register of thread 0), the 28 direction starts, the checks of k_dir_starts_check,
// and the report for the host - flags[0], n_bulk and the direction starts - written straight into pinned host memory
// (`report`, 64 ints: [8] flag, [9] n_bulk, [16..43] dir_start): no scan launches, no copy launches.
__global__ void __launch_bounds__(1024) k_border_scan(const int *__restrict__ cnt, int *__restrict__ off, int n, int nchunk, int *__restrict__ dir_start,
                                                      const int *__restrict__ n_bulk, int scan_beg, int bound, int *__restrict__ flags,
                                                      int *__restrict__ report)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + tid * 4;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? cnt[i0 + k] : 0;
        const int t = v[0] + v[1] + v[2] + v[3];
        int incl = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < w; k++) pre += wsum[k];
        int run = pre + incl - t;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i0 + k < n) off[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = run;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // (the scanned offsets were written by this workgroup: read them back through global memory after the barrier)
    if (tid < 28) {
        const int ds = off[(size_t)tid * nchunk];
        dir_start[tid] = ds;
        report[16 + tid] = ds;
    }
    if (tid == 0) {
        const int tot = off[(size_t)27 * nchunk];
        int f = flags[0];
        if (tot > bound) f = 200000;
        else if (*n_bulk < scan_beg) f = 200001;
        if (f) flags[0] = f;
        report[8] = f;
        report[9] = *n_bulk;
        report[10] = flags[5];       // fullest brick neighbourhood of the previous list build
        report[11] = flags[6];       // ... a 2-brick neighbourhood neared its stage
    }
}
// send list built on the pre-reorder order -> indices of the new order; also the last word of the host report (n_bulk)
__global__ void __launch_bounds__(256) k_translate_list(int *__restrict__ list, const int *__restrict__ inverse, int bound,
                                                        const int *__restrict__ n_dev, const int *__restrict__ n_bulk, int *__restrict__ report)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && report) report[9] = *n_bulk;
    const int n = *n_dev;          // (bound only sized the grid)
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) list[k] = inverse[list[k]];
}
void launch_translate_list(int *list, const int *inverse, int bound, const int *n_dev, const int *n_bulk, int *report, hipStream_t s)
{
    hipLaunchKernelGGL(k_translate_list, dim3(nblk(std::max(bound, 1), 256)), dim3(256), 0, s, list, inverse, bound, n_dev, n_bulk, report);
}

bool launch_border_scan(const int *cnt, int *off, int nchunk, int *dir_start, const int *n_bulk, int scan_beg, int bound, int *flags,
                        int *report, hipStream_t s)
{
    const int n = 27 * nchunk + 1;
    if (n > 65536) return false;
    hipLaunchKernelGGL(k_border_scan, dim3(1), dim3(1024), 0, s, cnt, off, n, nchunk, dir_start, n_bulk, scan_beg, bound, flags, report);
    return true;
}


// pack_border_vel (atom_vec_dpd_atomic_meso.cu:61-135), device resident: x(+shift), tag, type, mask
__global__ void __launch_bounds__(256) k_pack_border(AtomSoA a, const int *__restrict__ sendlist, int nsend,
                                                     const int *__restrict__ dir_start, Shift27 sh,
                                                     double *__restrict__ dx, double *__restrict__ dy,
                                                     double *__restrict__ dz, int *__restrict__ dtag,
                                                     int *__restrict__ dtype, int *__restrict__ dmask)
{
    __shared__ int ds[28];
    if (threadIdx.x < 28) ds[threadIdx.x] = dir_start[threadIdx.x];
    __syncthreads();
    // the list length is ds[27]; nsend only sized the grid (an estimate while the counts are on their way to the host)
    for (int k = blockDim.x * blockIdx.x + threadIdx.x; k < ds[27]; k += gridDim.x * blockDim.x) {
        int j = sendlist[k];
        int d = dir_of_entry(ds, k);
        dx[k] = a.x[0][j] + sh.s[d][0];
        dy[k] = a.x[1][j] + sh.s[d][1];
        dz[k] = a.x[2][j] + sh.s[d][2];
        dtag[k] = a.tag[j];
        dtype[k] = a.type[j];
        dmask[k] = a.mask[j];
    }
}

// pack_comm_vel (atom_vec_dpd_atomic_meso.cu:165-228) fused with gpu_merge_xvt for the ghost range:
// what travels per step is the merged float4 pair, already in the receiver's frame.
__global__ void __launch_bounds__(256) k_pack_forward(AtomSoA a, const int *__restrict__ sendlist, int nsend,
                                                      const int *__restrict__ dir_start, Shift27 sh, Center27 ce,
                                                      u32 seed, float4 *__restrict__ dcoord,
                                                      float4 *__restrict__ dveloc, const int *__restrict__ dest_slot,
                                                      int *__restrict__ img_cnt, int *__restrict__ img, int img_base,
                                                      const unsigned char *__restrict__ dirs)
{
    __shared__ int ds[28];
    if (threadIdx.x < 28) ds[threadIdx.x] = dir_start[threadIdx.x];
    __syncthreads();
    for (int k = blockDim.x * blockIdx.x + threadIdx.x; k < ds[27]; k += gridDim.x * blockDim.x) {      // (nsend only sized the grid)
        int j = sendlist[k];
        int d = dirs ? (int)dirs[k] : dir_of_entry(ds, k);      // (fused rebuild: ghosts in slot order, direction per entry)
        float4 c;
        c.x = (float)((a.x[0][j] + sh.s[d][0]) - ce.c[d][0]);
        c.y = (float)((a.x[1][j] + sh.s[d][1]) - ce.c[d][1]);
        c.z = (float)((a.x[2][j] + sh.s[d][2]) - ce.c[d][2]);
        c.w = __uint_as_float((u32)(a.type[j] - 1));
        const int out = dest_slot ? dest_slot[k] : k;
        dcoord[out] = c;
        if (img_cnt) {     // rebuild: remember where the images of atom j live (the step-boundary epilogue refreshes them)
            const int slot = atomicAdd(&img_cnt[j], 1);
            if (slot < 8) img[(size_t)j * 8 + slot] = (img_base + out) | (d << 26);
        }
        float4 v;
        v.x = (float)a.v[0][j];
        v.y = (float)a.v[1][j];
        v.z = (float)a.v[2][j];
        v.w = __uint_as_float(signature(seed, a.tag[j], v.x, v.y, v.z));
        dveloc[out] = v;
    }
}

void launch_pack_border(const AtomSoA &a, const int *sendlist, int nsend, const int *dir_start, const double *shift27,
                        double *dx, double *dy, double *dz, int *dtag, int *dtype, int *dmask, hipStream_t s)
{
    if (nsend <= 0) return;
    Shift27 sh;
    for (int d = 0; d < 27; d++)
        for (int k = 0; k < 3; k++) sh.s[d][k] = shift27[3 * d + k];
    hipLaunchKernelGGL(k_pack_border, dim3(nblk(nsend, 256)), dim3(256), 0, s, a, sendlist, nsend, dir_start, sh, dx, dy,
                       dz, dtag, dtype, dmask);
}

void launch_pack_forward(const AtomSoA &a, const int *sendlist, int nsend, const int *dir_start, const double *shift27,
                         const double *center27, uint32_t seed, float4 *dcoord, float4 *dveloc, const int *dest_slot,
                         int *img_cnt, int *img, int img_base, const unsigned char *dirs, hipStream_t s)
{
    if (nsend <= 0) return;
    Shift27 sh;
    Center27 ce;
    for (int d = 0; d < 27; d++)
        for (int k = 0; k < 3; k++) { sh.s[d][k] = shift27[3 * d + k]; ce.c[d][k] = center27[3 * d + k]; }
    hipLaunchKernelGGL(k_pack_forward, dim3(nblk(nsend, 256)), dim3(256), 0, s, a, sendlist, nsend, dir_start, sh, ce,
                       seed, dcoord, dveloc, dest_slot, img_cnt, img, img_base, dirs);
}

// =========================================================================================
// pair force, v1: one lane per i-particle over the transposed table
// =========================================================================================
template <bool FAST, bool EV>
__global__ void __launch_bounds__(256) k_pair_dpd(PairArgs a)
{
    extern __shared__ double smem[];
    double *cf64 = smem;
    float *cf32 = (float *)smem;
    const int ncf = a.ntypes * a.ntypes * N_COEFF;
    for (int p = threadIdx.x; p < ncf; p += blockDim.x) {
        if (FAST) cf32[p] = a.coeff32[p];
        else cf64[p] = a.coeff64[p];
    }
    __syncthreads();
    // XCD-aware order (blocks b and b+8 share an L2): each XCD walks a contiguous range of atoms
    const int nbk = gridDim.x;
    const int blk = (nbk & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3));
    int i = a.beg + blk * blockDim.x + threadIdx.x;
    if (i >= a.end) return;

    const float4 c1 = a.coord4[i];
    const float4 v1 = a.veloc4[i];
    const u32 t1 = __float_as_uint(c1.w), s1 = __float_as_uint(v1.w);
    const int n = a.count[i];
    const int *col = a.table + ((size_t)(i >> 6) * a.n_col) * 64 + (i & 63);
    // row entry p: transposed 64-atom tiles, or chunked-8 words (cell-ordered builder)
    // (partitioned rows - RowPartArgs, kernels.h: entries n .. n + nb - 1 of this walk are the back section, a row of table_back)
    const int nb = a.nback ? a.nback[i] : 0;
    auto entry = [&](int p) -> int {
        if (p >= n) { p -= n; return a.table_back[row_word8(i, p >> 3, a.nb_col) * 8 + (p & 7)]; }
        return a.chunked ? a.table[row_word8(i, p >> 3, a.n_col) * 8 + (p & 7)] : col[(size_t)p * 64];
    };

    if (FAST) {
        float fx = 0.f, fy = 0.f, fz = 0.f, energy = 0.f;
        float vr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float dtis = (float)a.dt_inv_sqrt;
        for (int p = 0; p < n + nb; p++) {
            int j = entry(p);
            float4 c2 = a.coord4[j];
            float dx = c1.x - c2.x, dy = c1.y - c2.y, dz = c1.z - c2.z;
            float rsq = dx * dx + dy * dy + dz * dz;
            const float *cf = cf32 + (t1 * a.ntypes + __float_as_uint(c2.w)) * N_COEFF;
            if (rsq < cf[P_CUTSQ] && rsq >= (float)MESO_EPSILON_SQ) {
                float4 v2 = a.veloc4[j];
                float rn = pair_noise_fast(a.rng, s1, __float_as_uint(v2.w));
                float rinv = __builtin_amdgcn_rsqf(rsq);
                float r = rsq * rinv;
                float dvx = v1.x - v2.x, dvy = v1.y - v2.y, dvz = v1.z - v2.z;
                float dot = dx * dvx + dy * dvy + dz * dvz;
                float wc = 1.0f - r * cf[P_CUTINV];
                float ew = cf[P_EXPW];
                float wr = (ew == 1.0f) ? wc : __powf(wc, ew);
                const float *prow = a.poly ? a.poly + (t1 * a.ntypes + __float_as_uint(c2.w)) * MESO_POLY_PITCH : nullptr;
                float fcons = prow ? polyval_f32(wc, prow) : cf[P_A0] * wc;
                if (a.ftab) fcons = table_force_f32(r * cf[P_CUTINV], a.ftab + (t1 * a.ntypes + __float_as_uint(c2.w)) * a.ftab_len, a.ftab_len);
                float fpair = fcons - (cf[P_GAMMA] * wr * wr * dot * rinv) + (cf[P_SIGMA] * wr * rn * dtis);
                fpair *= rinv;
                fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
                if (EV) {
                    vr[0] += dx * dx * fpair; vr[1] += dy * dy * fpair; vr[2] += dz * dz * fpair;
                    vr[3] += dx * dy * fpair; vr[4] += dx * dz * fpair; vr[5] += dy * dz * fpair;
                    // (polyforce :176; the tableforce kernel books no pair energy, pair_dpd_tableforce_meso.cu:191-198)
                    energy += a.ftab ? 0.f : prow ? polyval_integral_f32(wc, prow) : 0.5f * cf[P_A0] * cf[P_CUT] * wc * wc;
                }
            }
        }
        if (a.accumulate) { a.f[0][i] += fx; a.f[1][i] += fy; a.f[2][i] += fz; }
        else { a.f[0][i] = fx; a.f[1][i] = fy; a.f[2][i] = fz; }
        if (EV) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (a.accumulate) a.virial[k][i] += vr[k] * 0.5f;
                else a.virial[k][i] = vr[k] * 0.5f;
            }
            a.e_pair[i] = energy * 0.5f;
        }
    } else {
        double fx = 0., fy = 0., fz = 0., energy = 0.;
        double vr[6] = {0., 0., 0., 0., 0., 0.};
        for (int p = 0; p < n + nb; p++) {
            int j = entry(p);
            float4 c2 = a.coord4[j];
            double dx = (double)c1.x - (double)c2.x;
            double dy = (double)c1.y - (double)c2.y;
            double dz = (double)c1.z - (double)c2.z;
            double rsq = dx * dx + dy * dy + dz * dz;
            const double *cf = cf64 + (t1 * a.ntypes + __float_as_uint(c2.w)) * N_COEFF;
            if (rsq < cf[P_CUTSQ] && rsq >= MESO_EPSILON_SQ) {
                float4 v2 = a.veloc4[j];
                double rn = gaussian_tea(s1, __float_as_uint(v2.w));
                double rinv = rsqrt(rsq);
                double r = rsq * rinv;
                double dvx = (double)v1.x - (double)v2.x;
                double dvy = (double)v1.y - (double)v2.y;
                double dvz = (double)v1.z - (double)v2.z;
                double dot = dx * dvx + dy * dvy + dz * dvz;
                double wc = 1.0 - r * cf[P_CUTINV];
                double ew = cf[P_EXPW];
                double wr = (ew == 1.0) ? wc : powd_poly(wc, ew);
                double fpair = cf[P_A0] * wc - (cf[P_GAMMA] * wr * wr * dot * rinv) +
                               (cf[P_SIGMA] * wr * rn * a.dt_inv_sqrt);
                fpair *= rinv;
                fx += dx * fpair; fy += dy * fpair; fz += dz * fpair;
                if (EV) {
                    vr[0] += dx * dx * fpair; vr[1] += dy * dy * fpair; vr[2] += dz * dz * fpair;
                    vr[3] += dx * dy * fpair; vr[4] += dx * dz * fpair; vr[5] += dy * dz * fpair;
                    energy += 0.5 * cf[P_A0] * cf[P_CUT] * wc * wc;
                }
            }
        }
        if (a.accumulate) { a.f[0][i] += fx; a.f[1][i] += fy; a.f[2][i] += fz; }
        else { a.f[0][i] = fx; a.f[1][i] = fy; a.f[2][i] = fz; }
        if (EV) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (a.accumulate) a.virial[k][i] += vr[k] * 0.5;
                else a.virial[k][i] = vr[k] * 0.5;
            }
            a.e_pair[i] = energy * 0.5;
        }
    }
}


// =========================================================================================
// cell-ordered list builder + pair force v3 (memory-level parallelism)
// =========================================================================================
// Locals are stored in cell order by the reorder sort ([border][Morton(bin)][sub-cell]) and ghosts are sorted by
// Morton(bin) behind them, so the atoms of any bin are at most three contiguous runs of the merged arrays
// (bulk, border, ghost) and a candidate's global index IS its position: no per-candidate indirection, and the
// 27-bin walk of a lane reads contiguous float4 runs that its wave-mates (same or adjacent bins) share in L1.
__device__ inline float dist2(float4 a, float4 b)
{
    float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}
__device__ inline u32 compact3b(u32 x)
{
    x &= 0x09249249;
    x = (x ^ (x >> 2)) & 0x030c30c3;
    x = (x ^ (x >> 4)) & 0x0300f00f;
    x = (x ^ (x >> 8)) & 0xff0000ff;
    x = (x ^ (x >> 16)) & 0x000003ff;
    return x;
}

// binrange[m] = {bulk start,end, border start,end | ghost start,end (absolute), 0, 0}: the three runs of bin m
__global__ void __launch_bounds__(256) k_bin_ranges(const int *__restrict__ estart, const int *__restrict__ gstart, int M,
                                                    int nlocal, int4 *__restrict__ binrange)
{
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    binrange[2 * (size_t)m] = make_int4(estart[m], estart[m + 1], estart[(size_t)M + m], estart[(size_t)M + m + 1]);
    binrange[2 * (size_t)m + 1] = make_int4(nlocal + gstart[m], nlocal + gstart[m + 1], 0, 0);
}

void launch_bin_ranges(const int *estart, const int *gstart, int M, int nlocal, int4 *binrange, hipStream_t s)
{
    hipLaunchKernelGGL(k_bin_ranges, dim3(nblk(M, 256)), dim3(256), 0, s, estart, gstart, M, nlocal, binrange);
}

// per-lane 8-entry staging in LDS: stage[q][lane]; a full chunk leaves as two 16-byte stores
#define CELL_PUSH(kk)                                                                         \
    {                                                                                         \
        stage[(n & 7) * 256 + threadIdx.x] = (kk);                                            \
        if ((n & 7) == 7 && n < n_col) {                                                      \
            int4 lo_ = make_int4(stage[threadIdx.x], stage[256 + threadIdx.x], stage[512 + threadIdx.x], stage[768 + threadIdx.x]); \
            int4 hi_ = make_int4(stage[1024 + threadIdx.x], stage[1280 + threadIdx.x], stage[1536 + threadIdx.x], (kk)); \
            int4 *w_ = rows + 2 * row_word8(i, n >> 3, n_col);                                \
            w_[0] = lo_; w_[1] = hi_;                                                         \
        }                                                                                     \
        n++;                                                                                  \
    }
#define CELL_TEST(kk, cc)                                                        \
    {                                                                            \
        float d_ = dist2(ci, cc);                                                \
        if ((kk) != i && d_ <= rc2 && !(nsp && cell_excluded(ex, i, nsp, (kk)))) CELL_PUSH(kk) \
    }

// gpu_filter_exclusion (neigh_build_meso.cu:497-544): drop special partners by tag
__device__ inline bool cell_excluded(const ExclArgs &ex, int i, int nsp, int k)
{
    const int t = ex.tagc[k];
    bool hit = false;
    for (int s = 0; s < nsp; s++) hit |= ex.special[(size_t)i * ex.msp + s] == t;
    return hit;
}

__device__ inline void cell_run(const float4 *__restrict__ coord4, int kb, int ke, const float4 ci, int i, float rc2,
                                int n_col, int4 *__restrict__ rows, int *stage, int &n, const ExclArgs &ex, int nsp)
{
    int k = kb;
    for (; k + 4 <= ke; k += 4) {     // four independent loads in flight per lane
        float4 c0 = coord4[k], c1 = coord4[k + 1], c2 = coord4[k + 2], c3 = coord4[k + 3];
        CELL_TEST(k, c0) CELL_TEST(k + 1, c1) CELL_TEST(k + 2, c2) CELL_TEST(k + 3, c3)
    }
    for (; k < ke; k++) {
        float4 c0 = coord4[k];
        CELL_TEST(k, c0)
    }
}

__global__ void __launch_bounds__(256) k_cell_build(const float4 *__restrict__ coord4, const u32 *__restrict__ key,
                                                    int key_shift, const int4 *__restrict__ binrange, int M, int mbx,
                                                    int mby, int mbz, float rc2, int nlocal, int n_col,
                                                    int *__restrict__ count, int *__restrict__ table,
                                                    int *__restrict__ overflow, ExclArgs ex)
{
    __shared__ int stage[8 * 256];
    const int nbk = gridDim.x;
    const int blk = (nbk & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (nbk >> 3) + (blockIdx.x >> 3));
    const int i = blk * blockDim.x + threadIdx.x;
    if (i >= nlocal) return;
    const float4 ci = coord4[i];
    const u32 m = (key[i] >> key_shift) & (u32)(M - 1);
    const int bx = (int)compact3b(m), by = (int)compact3b(m >> 1), bz = (int)compact3b(m >> 2);
    int4 *rows = (int4 *)table;
    int n = 0;
    const int nsp = ex.tagc ? ex.nspecial[i] : 0;
    // one (dy,dz) row of three x-adjacent bins at a time; the six range words of the NEXT row are requested
    // before the current row's candidates are walked, so the dependent bin->range->atoms chain overlaps
    int4 ra[3], rb[3];
    auto fetch = [&](int r, int4 *qa, int4 *qb) {
        const int y2 = by + r % 3 - 1, z2 = bz + r / 3 - 1;
        const bool rowok = r < 9 && y2 >= 0 && y2 < mby && z2 >= 0 && z2 < mbz;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int x2 = bx + t - 1;
            qa[t] = make_int4(0, 0, 0, 0);
            qb[t] = make_int4(0, 0, 0, 0);
            if (rowok && x2 >= 0 && x2 < mbx) {
                const size_t m2 = interleave3((u32)x2, (u32)y2, (u32)z2);
                qa[t] = binrange[2 * m2];
                qb[t] = binrange[2 * m2 + 1];
            }
        }
    };
    fetch(0, ra, rb);
#pragma unroll 1
    for (int r = 0; r < 9; r++) {
        int4 na[3], nb[3];
        fetch(r + 1, na, nb);
#pragma unroll
        for (int t = 0; t < 3; t++) {
            cell_run(coord4, ra[t].x, ra[t].y, ci, i, rc2, n_col, rows, stage, n, ex, nsp);
            if (ra[t].w > ra[t].z) cell_run(coord4, ra[t].z, ra[t].w, ci, i, rc2, n_col, rows, stage, n, ex, nsp);
            if (rb[t].y > rb[t].x) cell_run(coord4, rb[t].x, rb[t].y, ci, i, rc2, n_col, rows, stage, n, ex, nsp);
        }
#pragma unroll
        for (int t = 0; t < 3; t++) { ra[t] = na[t]; rb[t] = nb[t]; }
    }
    // tail chunk (unused slots point at the atom itself: rsq = 0 is rejected by the force kernel)
    if ((n & 7) && n < n_col) {
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = q < (n & 7) ? stage[q * 256 + threadIdx.x] : i;
        int4 *w_ = rows + 2 * row_word8(i, n >> 3, n_col);
        w_[0] = make_int4(v[0], v[1], v[2], v[3]);
        w_[1] = make_int4(v[4], v[5], v[6], v[7]);
    }
    if (n > n_col) { atomicMax(overflow, n); n = n_col; }
    count[i] = n;
}

void launch_cell_build(const float4 *coord4, const uint32_t *sorted_key, int key_shift, const int4 *binrange, int M,
                       const int *mbin, float rc2, int nlocal, int n_col, int *count, int *table, int *overflow,
                       const ExclArgs *excl, hipStream_t s)
{
    if (nlocal <= 0) return;
    int g = (nblk(nlocal, 256) + 7) / 8 * 8;
    ExclArgs ex = {nullptr, nullptr, nullptr, 0};
    if (excl) ex = *excl;
    hipLaunchKernelGGL(k_cell_build, dim3(g), dim3(256), 0, s, coord4, sorted_key, key_shift, binrange, M, mbin[0], mbin[1],
                       mbin[2], rc2, nlocal, n_col, count, table, overflow, ex);
}

void launch_pair_dpd(const PairArgs &p, int fast, int evflag, hipStream_t s)
{
    int n = p.end - p.beg;
    if (n <= 0) return;
    size_t sm = (size_t)p.ntypes * p.ntypes * N_COEFF * (fast ? sizeof(float) : sizeof(double));
    dim3 grid((nblk(n, 256) + 7) / 8 * 8), block(256);
    if (fast && evflag) hipLaunchKernelGGL((k_pair_dpd<true, true>), grid, block, sm, s, p);
    else if (fast) hipLaunchKernelGGL((k_pair_dpd<true, false>), grid, block, sm, s, p);
    else if (evflag) hipLaunchKernelGGL((k_pair_dpd<false, true>), grid, block, sm, s, p);
    else hipLaunchKernelGGL((k_pair_dpd<false, false>), grid, block, sm, s, p);
}

// =========================================================================================
// known-answer test kernels
// =========================================================================================
__global__ void k_test_tea(const u32 *u, const u32 *v, int n, int rounds, u32 *o0, u32 *o1)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 a = u[i], b = v[i];
    switch (rounds) {
    case 4: tea_core<4>(a, b); break;
    case 8: tea_core<8>(a, b); break;
    case 16: tea_core<16>(a, b); break;
    case 64: tea_core<64>(a, b); break;
    default: for (int r = 0; r < rounds; r++) { u32 sum = MESO_TEA_DT * (u32)(r + 1);
            a += ((b << 4) + MESO_TEA_K0) ^ (b + sum) ^ ((b >> 5) + MESO_TEA_K1);
            b += ((a << 4) + MESO_TEA_K2) ^ (a + sum) ^ ((a >> 5) + MESO_TEA_K3); }
    }
    o0[i] = a; o1[i] = b;
}
void launch_test_tea(const uint32_t *u, const uint32_t *v, int n, int rounds, uint32_t *out0, uint32_t *out1,
                     hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_test_tea, dim3(nblk(n, 256)), dim3(256), 0, s, u, v, n, rounds, out0, out1);
}

__global__ void k_test_gaussian(const u32 *u, const u32 *v, int n, double *odp, float *osp)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    odp[i] = gaussian_tea(u[i], v[i]);
    osp[i] = gaussian_tea_fast(u[i], v[i]);
}
void launch_test_gaussian(const uint32_t *u, const uint32_t *v, int n, double *out_dp, float *out_sp, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_test_gaussian, dim3(nblk(n, 256)), dim3(256), 0, s, u, v, n, out_dp, out_sp);
}

__global__ void k_test_logistic(const u32 *u, const u32 *v, int n, float *o)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = logistic_noise(u[i], v[i]);
}
void launch_test_logistic(const uint32_t *u, const uint32_t *v, int n, float *out, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_test_logistic, dim3(nblk(n, 256)), dim3(256), 0, s, u, v, n, out);
}

} // namespace meso
