// Micro-benchmark: issue cost of VALU instruction kinds on gfx950 (cycles per wave-instruction at 8 waves per SIMD, long
// independent chains): v_fma_f32, v_pk_fma_f32, v_cmp (VCC / SGPR destination), v_mbcnt pair, v_readlane, v_lshl_add_u32,
// v_mad_u64_u32, v_cndmask.   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x

#define CHAIN4U(NAME) REP16(asm volatile(NAME " %0, %0, %1\n " NAME " %1, %1, %2\n " NAME " %2, %2, %3\n " NAME " %3, %3, %0" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));)
#define CHAIN4F(NAME) REP16(asm volatile(NAME " %0, %0, %1\n " NAME " %1, %1, %2\n " NAME " %2, %2, %3\n " NAME " %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
#define CHAIN4U3(NAME) REP16(asm volatile(NAME " %0, %0, %1, %2\n " NAME " %1, %1, %2, %3\n " NAME " %2, %2, %3, %0\n " NAME " %3, %3, %0, %1" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));)
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1;
    double dd0 = a0, dd1 = a1;
    unsigned long long s = 0;
    unsigned w0 = threadIdx.x, w1 = w0 * 3 + 1, w2 = w0 * 5 + 2, w3 = w0 * 7 + 3;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %0\n v_fma_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 1) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
            a0 = p0[0]; a1 = p0[1]; a2 = p1[0]; a3 = p1[1]; a4 = p2[0]; a5 = p2[1]; a6 = p3[0]; a7 = p3[1];
        } else if (MODE == 2) { REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        else if (MODE == 3) { REP16(asm volatile("v_cmp_lt_f32 %0, %1, %2\n v_cmp_lt_f32 %0, %2, %3\n v_cmp_lt_f32 %0, %3, %4\n v_cmp_lt_f32 %0, %4, %1" : "=s"(s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        else if (MODE == 4) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %2, 0\n v_mbcnt_hi_u32_b32 %0, %3, %0\n v_mbcnt_lo_u32_b32 %1, %3, 0\n v_mbcnt_hi_u32_b32 %1, %2, %1" : "+v"(u0), "+v"(u1) : "s"((unsigned)it), "s"((unsigned)iters));) }
        else if (MODE == 5) { unsigned r; REP16(asm volatile("v_readlane_b32 %0, %1, 3\n v_readlane_b32 %0, %2, 5\n v_readlane_b32 %0, %1, 7\n v_readlane_b32 %0, %2, 9" : "=s"(r) : "v"(u0), "v"(u1));) u0 += r; }
        else if (MODE == 6) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0\n v_lshl_add_u32 %0, %0, 1, %1\n v_lshl_add_u32 %1, %1, 1, %0" : "+v"(u0), "+v"(u1));) }
        else if (MODE == 7) { unsigned long long q = u0; REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %2, %1, %0" : "+v"(q) : "v"(u0), "v"(u1) : "vcc");) u0 = (unsigned)q; }
        else if (MODE == 8) { REP16(asm volatile("v_cmp_lt_f32 %0, %1, %2\n s_and_b64 %0, %0, exec\n v_cmp_lt_f32 %0, %3, %4\n s_and_b64 %0, %0, exec" : "=s"(s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        else if (MODE == 10) { CHAIN4U("v_add_u32") }
        else if (MODE == 11) { CHAIN4U("v_xor_b32") }
        else if (MODE == 12) { CHAIN4U("v_lshlrev_b32") }
        else if (MODE == 13) { CHAIN4U3("v_or3_b32") }
        else if (MODE == 14) { CHAIN4U3("v_add3_u32") }
        else if (MODE == 15) { CHAIN4F("v_sub_f32") }
        else if (MODE == 16) { CHAIN4F("v_mul_f32") }
        else if (MODE == 17) { REP16(asm volatile("v_cvt_i32_f32 %0, %4\n v_cvt_i32_f32 %1, %5\n v_cvt_i32_f32 %2, %6\n v_cvt_i32_f32 %3, %7" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        else if (MODE == 18) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) :: "vcc");) }
        else if (MODE == 19) { CHAIN4U("v_and_b32") }
        else if (MODE == 20) { REP16(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %1, %2, %1\n v_bcnt_u32_b32 %2, %3, %2\n v_bcnt_u32_b32 %3, %0, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3));) }
        else if (MODE == 21) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %1, %1, %0, %0" : "+v"(dd0), "+v"(dd1));) }
        else if (MODE == 22) { CHAIN4U("v_mul_u32_u24") }
        else if (MODE == 23) { CHAIN4U("v_mul_lo_u32") }
        else if (MODE == 24) { REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        else if (MODE == 25) { CHAIN4U("v_lshrrev_b32") }
        else if (MODE == 26) { CHAIN4U("v_sub_u32") }
        else if (MODE == 9) {   // the builder's scan step: cmp -> sgpr, and, saveexec, mbcnt x2, lshl_add, ds_write, restore exec, bcnt, add
            unsigned long long e; unsigned n;
            REP16(asm volatile("v_cmp_ge_f32 %0, %4, %5\n s_and_saveexec_b64 %1, %0\n v_mbcnt_lo_u32_b32 %2, %7, 0\n v_mbcnt_hi_u32_b32 %2, %7, %2\n v_lshl_add_u32 %2, %2, 1, %6\n s_or_b64 exec, exec, %1\n s_bcnt1_i32_b64 %3, %0\n" : "=s"(s), "=s"(e), "+v"(u0), "=s"(n) : "v"(a0), "v"(a1), "v"(u1), "s"((unsigned)it));) u0 += n;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)u0 + (float)u1 + (float)(unsigned)s + (float)(w0 + w1 + w2 + w3) + (float)(dd0 + dd1);
}
template <int MODE> void run(float *d, const char *name, int per_iter)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, grid = 256 * 8;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 8 waves * iters * per_iter
    const double insts = 8.0 * iters * per_iter;
    printf("%-44s %.3f ms  %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_fma_f32", 64); run<1>(d, "v_pk_fma_f32", 64); run<2>(d, "v_cmp_lt_f32 -> vcc", 64); run<3>(d, "v_cmp_lt_f32 -> sgpr pair", 64);
    run<4>(d, "v_mbcnt lo/hi", 64); run<5>(d, "v_readlane_b32", 64); run<6>(d, "v_lshl_add_u32", 64); run<7>(d, "v_mad_u64_u32", 64);
    run<10>(d, "v_add_u32", 64); run<26>(d, "v_sub_u32", 64); run<11>(d, "v_xor_b32", 64); run<19>(d, "v_and_b32", 64); run<12>(d, "v_lshlrev_b32", 64); run<25>(d, "v_lshrrev_b32", 64);
    run<13>(d, "v_or3_b32", 64); run<14>(d, "v_add3_u32", 64); run<15>(d, "v_sub_f32", 64); run<16>(d, "v_mul_f32", 64); run<17>(d, "v_cvt_i32_f32", 64);
    run<18>(d, "v_cndmask_b32 (vcc)", 64); run<20>(d, "v_bcnt_u32_b32", 64); run<21>(d, "v_fma_f64", 64); run<22>(d, "v_mul_u32_u24", 64); run<23>(d, "v_mul_lo_u32", 64); run<24>(d, "v_rsq_f32", 64); run<9>(d, "scan-step skeleton (4 VALU + 3 SALU)", 16);
    return 0;
}
