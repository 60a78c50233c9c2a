// Bonded topology on the device (configs[4]: amphiphilic chains).  Replaces, device-resident:
//   gpu_set_map        atom_meso.cu:74-82        tag -> index map (atomicMin: owned copies win over ghosts)
//   gpu_map_bond       neighbor_meso.cu:86-104   partner tags -> indices after every rebuild
//   gpu_bond_harmonic  bond_harmonic_meso.cu:46-117   F = 2k(r-r0) r^ per stored bond (both atoms store it,
//                      newton off), minimum image, fp32 merged coordinates -> fp64 math
//   gpu_bond_fene      bond_fene_meso.cu:55-148  FENE + WCA per stored bond (rlogarg clamped at 0.1 like the reference)
//   gpu_map_angle      neighbor_meso.cu:161-182  /  gpu_angle_harmonic  angle_harmonic_meso.cu:46-172
//   gpu_filter_exclusion neigh_build_meso.cu:497-544  special (1-2/1-3/1-4) partners never enter the pair rows;
//                      here the tag compare is folded into the list builder instead of a second pass.
// Topology is stored per atom (row-major: bond_tag[i*bpa+b], special[i*msp+s]) so it moves with the atom through
// the reorder gather and the migration messages.
#include "kernels.h"
#include "meso_device.h"

namespace meso {

static inline int nblk(long n, int b) { return (int)((n + b - 1) / b); }

// cell-order tags: locals as stored, ghosts scattered to their Morton(bin) slots
__global__ void __launch_bounds__(256) k_tag_cell(const int *__restrict__ tag, const int *__restrict__ gslot, int nlocal,
                                                  int nghost, const int *__restrict__ nghost_dev, int *__restrict__ tagc)
{
    // nghost_dev: nghost only sized the grid (an estimate); the count is on the device and the loop covers whatever it turns out to be
    if (nghost_dev) nghost = *nghost_dev;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nlocal + nghost; i += gridDim.x * blockDim.x) {
        if (i < nlocal) tagc[i] = tag[i];
        else tagc[nlocal + (gslot ? gslot[i - nlocal] : i - nlocal)] = tag[i];
    }
}

__global__ void __launch_bounds__(256) k_set_map(const int *__restrict__ tagc, int nall, int nlocal, const int *__restrict__ nghost_dev, int maxtag,
                                                 const u32 *__restrict__ bits, int *__restrict__ map)
{
    if (nghost_dev) nall = nlocal + *nghost_dev;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nall; i += gridDim.x * blockDim.x) {
        int t = tagc[i];
        // (bits: only tags that a bond or an angle names are ever looked up - one scattered atomic per chain bead, not per atom)
        if (t >= 0 && t <= maxtag && (!bits || ((bits[t >> 5] >> (t & 31)) & 1u))) atomicMin(map + t, i);
    }
}

__global__ void __launch_bounds__(256) k_map_bonds(const int *__restrict__ nbond, const int *__restrict__ bond_tag, int bpa,
                                                   const int *__restrict__ map, int maxtag, int nlocal,
                                                   int *__restrict__ bond_idx, int *__restrict__ missing)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlocal) return;
    int n = nbond[i];
    for (int b = 0; b < n; b++) {
        int t = bond_tag[(size_t)i * bpa + b];
        int j = (t >= 0 && t <= maxtag) ? map[t] : 0x7fffffff;
        if (j >= 0x7f000000) { atomicAdd(missing, 1); j = i; }      // (map is preset byte-wise to 0x7f7f7f7f)
        bond_idx[(size_t)i * bpa + b] = j;
    }
}

// STYLE 0: harmonic (coefficient table [k][r0]); STYLE 1: FENE ([k][r0][epsilon][sigma]), bond_fene_meso.cu:82-147 ==
// BondFENE::compute src/MOLECULE/bond_fene.cpp:48-124 with the warning/abort branches replaced by the clamp the
// reference's kernel applies
template <int STYLE, bool EV>
__global__ void __launch_bounds__(256) k_bond(const float4 *__restrict__ coord4, const int *__restrict__ nbond,
                                              const int *__restrict__ bond_idx, const int *__restrict__ bond_type, int bpa,
                                              const double *__restrict__ cf, int nbt, double px, double py, double pz,
                                              int nlocal, double *__restrict__ fx_, double *__restrict__ fy_,
                                              double *__restrict__ fz_, double *__restrict__ e_bond, int store)
{
    extern __shared__ double sh[];
    const int ncf = (STYLE == 1 ? 4 : 2) * (nbt + 1);
    for (int t = threadIdx.x; t < ncf; t += blockDim.x) sh[t] = cf[t];
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nlocal; i += gridDim.x * blockDim.x) {
        const int n = nbond[i];
        if (n == 0) {
            if (EV) e_bond[i] = 0.0;
            if (store && fx_) { fx_[i] = 0.0; fy_[i] = 0.0; fz_[i] = 0.0; }      // store mode: this kernel opens the step's forces
            continue;
        }
        const float4 c1 = coord4[i];
        double fx, fy, fz, e;
        bond_forces_of_atom<STYLE, EV>(coord4, c1, n, bond_idx + (size_t)i * bpa, bond_type + (size_t)i * bpa, sh, nbt, px, py, pz, fx, fy, fz, e);
        if (fx_) {   // null: energy-only pass (compute_ebond)
            if (store) { fx_[i] = fx; fy_[i] = fy; fz_[i] = fz; }
            else { fx_[i] += fx; fy_[i] += fy; fz_[i] += fz; }
        }
        if (EV) e_bond[i] = e * 0.5;
    }
}

__global__ void __launch_bounds__(256) k_map_angles(const int *__restrict__ tag, const int *__restrict__ nangle,
                                                    const int *__restrict__ angle_tag, int apa, const int *__restrict__ map,
                                                    int maxtag, int nlocal, int *__restrict__ angle_idx, int *__restrict__ missing)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlocal) return;
    const int n = nangle[i], me = tag[i];
    for (int a = 0; a < n; a++) {
        for (int c = 0; c < 3; c++) {
            const int t = angle_tag[((size_t)i * apa + a) * 4 + c];
            int j = t == me ? i : ((t >= 0 && t <= maxtag) ? map[t] : 0x7fffffff);
            if (j >= 0x7f000000) { atomicAdd(missing, 1); j = i; }
            angle_idx[((size_t)i * apa + a) * 3 + c] = j;
        }
    }
}

// gpu_angle_harmonic (angle_harmonic_meso.cu:77-172) == AngleHarmonic::compute (src/MOLECULE/angle_harmonic.cpp:50-142) seen
// from one atom: E = k (theta - theta0)^2; each of the three atoms evaluates the angle and keeps its own force.  Energy:
// every atom books a third of each of its angles (the reference kernel overwrites instead of accumulating - ':149
// eangle = tk * dtheta' - and books half; the sum here equals the stock style's eangle).
template <bool EV>
__global__ void __launch_bounds__(256) k_angle_harmonic(const float4 *__restrict__ coord4, const int *__restrict__ nangle,
                                                        const int *__restrict__ angle_idx, const int *__restrict__ angle_tag,
                                                        int apa, const double *__restrict__ cf, int nat, double px, double py,
                                                        double pz, int nlocal, double *__restrict__ fx_,
                                                        double *__restrict__ fy_, double *__restrict__ fz_,
                                                        double *__restrict__ e_angle)
{
    extern __shared__ double sh[];
    for (int t = threadIdx.x; t < 2 * (nat + 1); t += blockDim.x) sh[t] = cf[t];
    __syncthreads();
    const double *k = sh, *theta0 = sh + nat + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nlocal; i += gridDim.x * blockDim.x) {
        const int n = nangle[i];
        if (n == 0) { if (EV) e_angle[i] = 0.0; continue; }
        double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
        for (int a = 0; a < n; a++) {
            const int *ix = angle_idx + ((size_t)i * apa + a) * 3;
            const int i1 = ix[0], i2 = ix[1], i3 = ix[2], type = angle_tag[((size_t)i * apa + a) * 4 + 3];
            const float4 c1 = coord4[i1], c2 = coord4[i2], c3 = coord4[i3];
            const double dx1 = min_image((double)c1.x - (double)c2.x, px), dy1 = min_image((double)c1.y - (double)c2.y, py),
                         dz1 = min_image((double)c1.z - (double)c2.z, pz);
            const double rsq1 = dx1 * dx1 + dy1 * dy1 + dz1 * dz1, rinv1 = rsqrt(rsq1);
            const double dx2 = min_image((double)c3.x - (double)c2.x, px), dy2 = min_image((double)c3.y - (double)c2.y, py),
                         dz2 = min_image((double)c3.z - (double)c2.z, pz);
            const double rsq2 = dx2 * dx2 + dy2 * dy2 + dz2 * dz2, rinv2 = rsqrt(rsq2);
            double c = (dx1 * dx2 + dy1 * dy2 + dz1 * dz2) * rinv1 * rinv2;
            c = fmin(1.0, fmax(-1.0, c));
            const double sn = rsqrt(fmax(1.0 - c * c, 0.001));       // SMALL = 0.001 (angle_harmonic_meso.cu:27)
            const double dtheta = acos(c) - theta0[type];
            const double tk = k[type] * dtheta;
            const double aa = -2.0 * tk * sn;
            const double a11 = aa * c * rinv1 * rinv1, a12 = -aa * rinv1 * rinv2, a22 = aa * c * rinv2 * rinv2;
            double dfx = 0.0, dfy = 0.0, dfz = 0.0;
            if (i != i3) { dfx += a11 * dx1 + a12 * dx2; dfy += a11 * dy1 + a12 * dy2; dfz += a11 * dz1 + a12 * dz2; }
            if (i != i1) { dfx += a22 * dx2 + a12 * dx1; dfy += a22 * dy2 + a12 * dy1; dfz += a22 * dz2 + a12 * dz1; }
            const double sg = i != i2 ? 1.0 : -1.0;
            fx += sg * dfx; fy += sg * dfy; fz += sg * dfz;
            if (EV) e += tk * dtheta;
        }
        if (fx_) { fx_[i] += fx; fy_[i] += fy; fz_[i] += fz; }
        if (EV) e_angle[i] = e * (1.0 / 3.0);
    }
}

void launch_map_angles(const int *tag, const int *nangle, const int *angle_tag, int apa, const int *map, int maxtag, int nlocal,
                       int *angle_idx, int *missing, hipStream_t s)
{
    if (nlocal > 0)
        hipLaunchKernelGGL(k_map_angles, dim3(nblk(nlocal, 256)), dim3(256), 0, s, tag, nangle, angle_tag, apa, map, maxtag, nlocal,
                           angle_idx, missing);
}
void launch_angle_harmonic(const float4 *coord4, const int *nangle, const int *angle_idx, const int *angle_tag, int apa,
                           const double *cf, int nat, const double *prd, int nlocal, double *fx, double *fy, double *fz,
                           double *e_angle, hipStream_t s)
{
    if (nlocal <= 0) return;
    int g = nblk(nlocal, 256);
    if (g > 2048) g = 2048;
    size_t sm = 2 * (size_t)(nat + 1) * sizeof(double);
    if (e_angle)
        hipLaunchKernelGGL(k_angle_harmonic<true>, dim3(g), dim3(256), sm, s, coord4, nangle, angle_idx, angle_tag, apa, cf, nat,
                           prd[0], prd[1], prd[2], nlocal, fx, fy, fz, e_angle);
    else
        hipLaunchKernelGGL(k_angle_harmonic<false>, dim3(g), dim3(256), sm, s, coord4, nangle, angle_idx, angle_tag, apa, cf, nat,
                           prd[0], prd[1], prd[2], nlocal, fx, fy, fz, e_angle);
}

void launch_tag_cell(const int *tag, const int *gslot, int nlocal, int nghost, const int *nghost_dev, int *tagc, hipStream_t s)
{
    if (nlocal + nghost > 0)
        hipLaunchKernelGGL(k_tag_cell, dim3(nblk(nlocal + nghost, 256)), dim3(256), 0, s, tag, gslot, nlocal, nghost, nghost_dev, tagc);
}
void launch_set_map(const int *tagc, int nlocal, int nghost, const int *nghost_dev, int maxtag, const uint32_t *tagbits, int *map, hipStream_t s)
{
    if (nlocal + nghost > 0)
        hipLaunchKernelGGL(k_set_map, dim3(nblk(nlocal + nghost, 256)), dim3(256), 0, s, tagc, nlocal + nghost, nlocal, nghost_dev, maxtag, tagbits, map);
}
void launch_map_bonds(const int *nbond, const int *bond_tag, int bpa, const int *map, int maxtag, int nlocal, int *bond_idx,
                      int *missing, hipStream_t s)
{
    if (nlocal > 0)
        hipLaunchKernelGGL(k_map_bonds, dim3(nblk(nlocal, 256)), dim3(256), 0, s, nbond, bond_tag, bpa, map, maxtag, nlocal,
                           bond_idx, missing);
}
void launch_bond(int style, const float4 *coord4, const int *nbond, const int *bond_idx, const int *bond_type, int bpa,
                 const double *cf, int nbt, const double *prd, int nlocal, double *fx, double *fy, double *fz, double *e_bond,
                 int store, hipStream_t s)
{
    if (nlocal <= 0) return;
    int g = nblk(nlocal, 256);
    if (g > 2048) g = 2048;
    size_t sm = (style == 1 ? 4 : 2) * (size_t)(nbt + 1) * sizeof(double);
#define BOND_LAUNCH(S, E)                                                                                                  \
    hipLaunchKernelGGL((k_bond<S, E>), dim3(g), dim3(256), sm, s, coord4, nbond, bond_idx, bond_type, bpa, cf, nbt, prd[0],   \
                       prd[1], prd[2], nlocal, fx, fy, fz, e_bond, store)
    if (style == 1) { if (e_bond) BOND_LAUNCH(1, true); else BOND_LAUNCH(1, false); }
    else { if (e_bond) BOND_LAUNCH(0, true); else BOND_LAUNCH(0, false); }
#undef BOND_LAUNCH
}

} // namespace meso
