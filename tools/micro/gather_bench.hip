// Random-gather throughput of one CU's vector memory path and of LDS on gfx950 (design input for the force kernel's data
// organisation).  Every lane gathers W bytes from a random element of a table of F elements (F sized
// so that the table sits in L1, in L2, or beyond), 8 independent gathers in flight per lane, 256 CUs x 5 workgroups x 4 waves.
// Prints CU-cycles per wave-level gather instruction (at the measured clock) and lane-gathers per second.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip && ./gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ inline u32 fold(T v);
template <> __device__ inline u32 fold<u32>(u32 v) { return v; }
template <> __device__ inline u32 fold<u32x2>(u32x2 v) { return v.x ^ v.y; }
template <> __device__ inline u32 fold<u32x4>(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// MODE 0: uniformly random element per lane; 1: lanes pairwise adjacent (2 lanes share a 2-element block);
// 2: quads adjacent; 3: fully coalesced (lane-consecutive) from a random base
template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_gather(const T *__restrict__ tab, u32 mask, int iters, u32 *out, u32 win_stride)
{
    const u32 lane = threadIdx.x & 63;
    const T *t = tab + (size_t)blockIdx.x * win_stride;     // each workgroup its own window (win_stride 0: all share one)
    u32 s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    u32 acc = 0;
    for (int it = 0; it < iters; it++) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            s = s * 1664525u + 1013904223u;
            u32 r = s >> 8;
            u32 j;
            if (MODE == 0) j = r & mask;
            else if (MODE == 1) j = ((__shfl(r, lane & ~1u, 64) & mask) & ~1u) | (lane & 1u);
            else if (MODE == 2) j = ((__shfl(r, lane & ~3u, 64) & mask) & ~3u) | (lane & 3u);
            else if (MODE == 3) j = ((__shfl(r, 0, 64) & mask) & ~63u) | lane;
            else if (MODE == 4) j = ((__shfl(r, lane & ~7u, 64) & mask) & ~31u) | ((r >> 20) & 31u);      // 8 lanes inside one 32-element window
            else if (MODE == 5) j = ((__shfl(r, lane & ~7u, 64) & mask) & ~15u) | ((r >> 20) & 15u);      // 8 lanes inside one 16-element window
            else if (MODE == 6) j = ((__shfl(r, lane & ~7u, 64) & mask) & ~7u) | ((r >> 20) & 7u);        // 8 lanes inside one 8-element window (one 128-B line of b128)
            else j = r & mask;
            // MODE 7 / 8: random, 7 / 12 of every 16 lanes switched off (exec mask) - does a gather cost per instruction or per active lane?
            if (MODE == 7 || MODE == 8) {
                T z; for (int k = 0; k < (int)(sizeof(T) / 4); k++) ((u32 *)&z)[k] = 0;
                v[q] = ((r >> 18) & 15u) >= (MODE == 7 ? 7u : 12u) ? t[j] : z;
            } else
            v[q] = t[j];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc ^= fold<T>(v[q]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename T>
__global__ void __launch_bounds__(256) k_lds_gather(u32 mask, int iters, u32 *out)
{
    extern __shared__ char sm[];
    T *t = (T *)sm;
    for (u32 i = threadIdx.x; i <= mask; i += 256) { T z; for (int k = 0; k < (int)(sizeof(T) / 4); k++) ((u32 *)&z)[k] = i + k; t[i] = z; }
    __syncthreads();
    u32 s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    u32 acc = 0;
    for (int it = 0; it < iters; it++) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            s = s * 1664525u + 1013904223u;
            v[q] = t[(s >> 8) & mask];
        }
#pragma unroll
        for (int q = 0; q < 8; q++) acc ^= fold<T>(v[q]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static double clock_ghz = 2.4;

template <typename T, int MODE>
static void run(const char *name, u32 elems, u32 win_stride, const T *tab, u32 *out)
{
    const int nb = 256 * 5, iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<T, MODE>), dim3(nb), dim3(256), 0, 0, tab, elems - 1, 20, out, win_stride);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_gather<T, MODE>), dim3(nb), dim3(256), 0, 0, tab, elems - 1, iters, out, win_stride);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)nb * 4 * iters * 8;                 // wave-level gather instructions
    double cyc = ms * 1e-3 * clock_ghz * 1e9 * 256 / winstr;    // CU cycles per instruction
    printf("%-44s %6.1f cycles/wave-instr  %7.1f G lane-gathers/s  %7.1f GB/s useful\n", name, cyc, winstr * 64 / (ms * 1e-3) / 1e9,
           winstr * 64 * sizeof(T) / (ms * 1e-3) / 1e9);
}

template <typename T>
static void run_lds(const char *name, u32 elems, u32 *out)
{
    const int nb = 256 * 2, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    size_t sm = (size_t)elems * sizeof(T);
    hipLaunchKernelGGL((k_lds_gather<T>), dim3(nb), dim3(256), sm, 0, elems - 1, 20, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_lds_gather<T>), dim3(nb), dim3(256), sm, 0, elems - 1, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)nb * 4 * iters * 8;
    double cyc = ms * 1e-3 * clock_ghz * 1e9 * 256 / winstr;
    printf("%-44s %6.1f cycles/wave-instr  %7.1f G lane-gathers/s\n", name, cyc, winstr * 64 / (ms * 1e-3) / 1e9);
}

int main()
{
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    if (khz > 0) clock_ghz = khz * 1e-6;
    printf("clock %.2f GHz\n", clock_ghz);
    const size_t bytes = (size_t)512 << 20;
    void *tab = nullptr;
    u32 *out = nullptr;
    hipMalloc(&tab, bytes);
    hipMalloc((void **)&out, 64);
    hipMemset(tab, 1, bytes);
    // window sizes in ELEMENTS; per-workgroup windows (stride = window) unless noted
    struct { const char *what; u32 bytes_win; } F[] = {{"4 KiB window (L1)", 4096}, {"16 KiB window (L1)", 16384}, {"64 KiB window (L2)", 65536},
                                                        {"256 KiB window (L2)", 262144}};
    for (auto &f : F) {
        char nm[128];
        snprintf(nm, sizeof nm, "global b32  random, %s", f.what);  run<u32, 0>(nm, f.bytes_win / 4, f.bytes_win / 4, (const u32 *)tab, out);
        snprintf(nm, sizeof nm, "global b64  random, %s", f.what);  run<u32x2, 0>(nm, f.bytes_win / 8, f.bytes_win / 8, (const u32x2 *)tab, out);
        snprintf(nm, sizeof nm, "global b128 random, %s", f.what);  run<u32x4, 0>(nm, f.bytes_win / 16, f.bytes_win / 16, (const u32x4 *)tab, out);
    }
    run<u32x4, 1>("global b128 lane pairs adjacent, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 2>("global b128 lane quads adjacent, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 3>("global b128 coalesced, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 4>("global b128 8 lanes in a 32-elem window, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 5>("global b128 8 lanes in a 16-elem window, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 6>("global b128 8 lanes in an 8-elem window, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 4>("global b128 8 lanes in a 32-elem window, 64 KiB", 4096, 4096, (const u32x4 *)tab, out);
    run<u32x4, 7>("global b128 random, 7/16 lanes off, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 8>("global b128 random, 12/16 lanes off, 16 KiB", 1024, 1024, (const u32x4 *)tab, out);
    run<u32x4, 7>("global b128 random, 7/16 lanes off, 64 KiB", 4096, 4096, (const u32x4 *)tab, out);
    run<u32, 7>("global b32 random, 7/16 lanes off, 16 KiB", 4096, 4096, (const u32 *)tab, out);
    run<u32x2, 2>("global b64  lane quads adjacent, 16 KiB", 2048, 2048, (const u32x2 *)tab, out);
    run<u32x2, 3>("global b64  coalesced, 16 KiB", 2048, 2048, (const u32x2 *)tab, out);
    run<u32, 3>("global b32  coalesced, 16 KiB", 4096, 4096, (const u32 *)tab, out);
    run<u32x4, 0>("global b128 random, 64 MiB shared table", 4u << 20, 0, (const u32x4 *)tab, out);
    run<u32x2, 0>("global b64  random, 32 MiB shared table", 4u << 20, 0, (const u32x2 *)tab, out);
    run_lds<u32>("LDS ds_read_b32  random, 32 KiB", 8192, out);
    run_lds<u32x2>("LDS ds_read_b64  random, 32 KiB", 4096, out);
    run_lds<u32x4>("LDS ds_read_b128 random, 32 KiB", 2048, out);
    hipFree(tab); hipFree(out);
    return 0;
}
