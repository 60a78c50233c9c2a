// Micro-benchmark: throughput of wave-scope LDS reductions on gfx950 (ds_add_f32 / ds_add_f64 / ds_add_u32 /
// ds_add_u64 / plain read-add-write), 64 lanes adding to 64 per-wave accumulators with a 3-4-way owner repeat
// pattern like the compacted pair kernel's batches.   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;

template <int MODE>
__global__ void __launch_bounds__(256) k(const int *__restrict__ owner, int iters, float *out)
{
    __shared__ double acc64[4][3][64];
    float *acc32 = (float *)&acc64[0][0][0];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = 0; c < 3; c++) acc64[w][c][lane] = 0.0;
    __syncthreads();
    float *a32 = acc32 + w * 3 * 64 * 2;
    double *a64 = &acc64[w][0][0];
    u32 *u32p = (u32 *)a32;
    u64 *u64p = (u64 *)a64;
    float x = 1.0f + lane * 1e-3f;
    for (int it = 0; it < iters; it++) {
        const int o = owner[(it & 63) * 64 + lane];
        x = x * 1.0001f + 0.5f;
        if (MODE == 0) {
            [[clang::atomic(no_remote_memory, no_fine_grained_memory, ignore_denormal_mode)]] {
                __hip_atomic_fetch_add(&a32[o], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&a32[64 + o], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_add(&a32[128 + o], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        } else if (MODE == 1) {
            __hip_atomic_fetch_add(&a64[o], (double)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&a64[64 + o], (double)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&a64[128 + o], (double)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (MODE == 2) {
            const u32 v = (u32)(int)(x * 262144.0f);
            __hip_atomic_fetch_add(&u32p[o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&u32p[64 + o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&u32p[128 + o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (MODE == 3) {
            const u64 v = (u64)(long long)(x * 262144.0f);
            __hip_atomic_fetch_add(&u64p[o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&u64p[64 + o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(&u64p[128 + o], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (MODE == 4) {     // non-atomic store of the result (12 B per lane), the owner sums later
            volatile float *r = a32 + lane * 3;
            r[0] = x; r[1] = x; r[2] = x;
        } else {                    // no LDS at all: the loop overhead
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = x + (float)acc64[w][0][lane] + acc32[threadIdx.x];
}

template <int MODE> double run(const int *d_owner, float *d_out, int iters, const char *name)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_owner, iters, d_out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_owner, iters, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double wave_ops = (double)grid * 4 * iters;          // wave-level "3-component adds"
    // per CU: 8 blocks * 4 waves * iters groups
    double cyc_per_group = ms * 1e-3 * 2.4e9 / (wave_ops / 256.0);
    printf("%-28s %8.3f ms   %6.1f CU-cycles per 64-lane x 3-component group (4 SIMDs share one LDS)\n", name, ms, cyc_per_group);
    return ms;
}

int main()
{
    std::vector<int> owner(64 * 64);
    unsigned s = 12345;
    for (int it = 0; it < 64; it++) {
        // a batch = ~2.3 ballot groups: ascending owner lanes, each lane present with p = 0.45
        int n = 0, o = 0;
        while (n < 64) {
            s = s * 1664525u + 1013904223u;
            if ((s >> 8) % 100 < 45) owner[it * 64 + n++] = o;
            o = (o + 1) & 63;
        }
    }
    int *d_owner; float *d_out;
    (void)hipMalloc(&d_owner, owner.size() * 4);
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4);
    (void)hipMemcpy(d_owner, owner.data(), owner.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    run<5>(d_owner, d_out, iters, "loop only");
    run<0>(d_owner, d_out, iters, "ds_add_f32 x3");
    run<1>(d_owner, d_out, iters, "ds_add_f64 x3");
    run<2>(d_owner, d_out, iters, "ds_add_u32 x3");
    run<3>(d_owner, d_out, iters, "ds_add_u64 x3");
    run<4>(d_owner, d_out, iters, "ds_write x3 (12 B/lane)");
    return 0;
}
