// Bonded topology on the device (configs[4]: amphiphilic chains).  Replaces, device-resident:
//   gpu_set_map        atom_meso.cu:74-82        tag -> index map (atomicMin: owned copies win over ghosts)
//   gpu_map_bond       neighbor_meso.cu:86-104   partner tags -> indices after every rebuild
//   gpu_bond_harmonic  bond_harmonic_meso.cu:46-117   F = 2k(r-r0) r^ per stored bond (both atoms store it,
//                      newton off), minimum image, fp32 merged coordinates -> fp64 math
//   gpu_bond_fene      bond_fene_meso.cu:55-148  FENE + WCA per stored bond (rlogarg clamped at 0.1 like the reference)
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
                                                  int nghost, int *__restrict__ tagc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nlocal + nghost) return;
    if (i < nlocal) tagc[i] = tag[i];
    else tagc[nlocal + (gslot ? gslot[i - nlocal] : i - nlocal)] = tag[i];
}

__global__ void __launch_bounds__(256) k_set_map(const int *__restrict__ tagc, int nall, int maxtag, int *__restrict__ map)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nall) return;
    int t = tagc[i];
    if (t >= 0 && t <= maxtag) atomicMin(map + t, i);
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
        if (j == 0x7fffffff) { atomicAdd(missing, 1); j = i; }
        bond_idx[(size_t)i * bpa + b] = j;
    }
}

__host__ __device__ inline double min_image(double dr, double p)   // math_meso.h:148-152
{
    double ph = p * 0.5;
    return dr + (dr > -ph ? (dr < ph ? 0.0 : -p) : p);
}

// STYLE 0: harmonic (coefficient table [k][r0]); STYLE 1: FENE ([k][r0][epsilon][sigma]), bond_fene_meso.cu:82-147 ==
// BondFENE::compute src/MOLECULE/bond_fene.cpp:48-124 with the warning/abort branches replaced by the clamp the
// reference's kernel applies
template <int STYLE, bool EV>
__global__ void __launch_bounds__(256) k_bond(const float4 *__restrict__ coord4, const int *__restrict__ nbond,
                                              const int *__restrict__ bond_idx, const int *__restrict__ bond_type, int bpa,
                                              const double *__restrict__ cf, int nbt, double px, double py, double pz,
                                              int nlocal, double *__restrict__ fx_, double *__restrict__ fy_,
                                              double *__restrict__ fz_, double *__restrict__ e_bond)
{
    extern __shared__ double sh[];
    const int ncf = (STYLE == 1 ? 4 : 2) * (nbt + 1);
    for (int t = threadIdx.x; t < ncf; t += blockDim.x) sh[t] = cf[t];
    __syncthreads();
    const double *k = sh, *r0 = sh + nbt + 1, *eps = sh + 2 * (nbt + 1), *sig = sh + 3 * (nbt + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nlocal; i += gridDim.x * blockDim.x) {
        const int n = nbond[i];
        if (n == 0) { if (EV) e_bond[i] = 0.0; continue; }
        const float4 c1 = coord4[i];
        double fx = 0.0, fy = 0.0, fz = 0.0, e = 0.0;
        for (int b = 0; b < n; b++) {
            const int j = bond_idx[(size_t)i * bpa + b], type = bond_type[(size_t)i * bpa + b];
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
        if (fx_) { fx_[i] += fx; fy_[i] += fy; fz_[i] += fz; }   // null: energy-only pass (compute_ebond)
        if (EV) e_bond[i] = e * 0.5;
    }
}

void launch_tag_cell(const int *tag, const int *gslot, int nlocal, int nghost, int *tagc, hipStream_t s)
{
    if (nlocal + nghost > 0)
        hipLaunchKernelGGL(k_tag_cell, dim3(nblk(nlocal + nghost, 256)), dim3(256), 0, s, tag, gslot, nlocal, nghost, tagc);
}
void launch_set_map(const int *tagc, int nall, int maxtag, int *map, hipStream_t s)
{
    if (nall > 0) hipLaunchKernelGGL(k_set_map, dim3(nblk(nall, 256)), dim3(256), 0, s, tagc, nall, maxtag, map);
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
                 hipStream_t s)
{
    if (nlocal <= 0) return;
    int g = nblk(nlocal, 256);
    if (g > 2048) g = 2048;
    size_t sm = (style == 1 ? 4 : 2) * (size_t)(nbt + 1) * sizeof(double);
#define BOND_LAUNCH(S, E)                                                                                                  \
    hipLaunchKernelGGL((k_bond<S, E>), dim3(g), dim3(256), sm, s, coord4, nbond, bond_idx, bond_type, bpa, cf, nbt, prd[0],   \
                       prd[1], prd[2], nlocal, fx, fy, fz, e_bond)
    if (style == 1) { if (e_bond) BOND_LAUNCH(1, true); else BOND_LAUNCH(1, false); }
    else { if (e_bond) BOND_LAUNCH(0, true); else BOND_LAUNCH(0, false); }
#undef BOND_LAUNCH
}

} // namespace meso
