// Dispatch latency of a chain of small DEPENDENT kernels on one stream: plain launches against one hipGraph launch of the same
// chain (the rebuild of a small box is such a chain: ~16 kernels of a few microseconds each).
//   hipcc --offload-arch=gfx950 -O3 -o graph_chain graph_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_small(int *p, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1;
}

int main()
{
    const int chain = 16, reps = 200;
    for (int n : {1024, 65536, 262144}) {
        int *d;
        CK(hipMalloc(&d, n * sizeof(int)));
        CK(hipMemset(d, 0, n * sizeof(int)));
        hipStream_t s;
        CK(hipStreamCreate(&s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        dim3 grid((n + 255) / 256), block(256);
        // plain launches
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; r++)
                for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_small, grid, block, 0, s, d, n);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
        }
        float ms_plain;
        CK(hipEventElapsedTime(&ms_plain, e0, e1));
        // the same chain as a graph
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < chain; k++) hipLaunchKernelGGL(k_small, grid, block, 0, s, d, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms_graph = 0;
        for (int w = 0; w < 2; w++) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_graph, e0, e1));
        }
        printf("n = %7d: chain of %d dependent kernels  plain %.2f us per kernel   graph %.2f us per kernel\n", n, chain,
               ms_plain * 1e3 / (reps * chain), ms_graph * 1e3 / (reps * chain));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        CK(hipFree(d));
    }
    return 0;
}
