// unit check of run_rank (meso_device.h) with dead lanes anywhere in the wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../meso_amd/csrc/meso_device.h"
using namespace meso;
__global__ void k(const u32 *code, const int *valid, int *cnt, int *tot, int *rank, int *dbg)
{
    const int i = threadIdx.x;
    const int r = run_rank(code[i], valid[i] != 0, cnt, tot, 6);
    rank[i] = r;
}
int main()
{
    const int n = 64;
    std::vector<u32> code(n, 0);
    std::vector<int> valid(n, 0), rank(n), cnt(256, 0), tot(4, 0);
    for (int i = 1; i <= 10; i++) valid[i] = 1;           // the failing pattern: lanes 1..10, code 0, everything else dead
    for (int i = 20; i < 30; i++) { valid[i] = 1; code[i] = 70 + (i & 1); }
    for (int i = 40; i < 64; i++) { valid[i] = (i % 3) != 0; code[i] = 5; }
    u32 *dc; int *dv, *dcnt, *dtot, *dr, *dd;
    hipMalloc(&dc, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&dcnt, 256 * 4); hipMalloc(&dtot, 16); hipMalloc(&dr, n * 4); hipMalloc(&dd, 64 * 4);
    hipMemcpy(dc, code.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, valid.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dcnt, 0, 256 * 4); hipMemset(dtot, 0, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dc, dv, dcnt, dtot, dr, dd);
    hipMemcpy(rank.data(), dr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(cnt.data(), dcnt, 256 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(tot.data(), dtot, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    std::vector<int> seen(256, 0);
    for (int i = 0; i < n; i++) if (valid[i]) { if (rank[i] < 0 || rank[i] >= cnt[code[i]]) bad++; seen[code[i]]++; }
    for (int c = 0; c < 256; c++) if (seen[c] != cnt[c]) { printf("code %d: %d atoms, counter %d\n", c, seen[c], cnt[c]); bad++; }
    for (int i = 0; i < n; i++) printf("%d:%d/%u->%d ", i, valid[i], code[i], rank[i]);
    printf("\ntot %d %d  bad %d\n", tot[0], tot[1], bad);
    return bad ? 1 : 0;
}
