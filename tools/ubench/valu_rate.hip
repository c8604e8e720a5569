// Issue rate of a few vector instructions on the device: every wave runs a long unrolled chain of one
// instruction on eight independent registers; all SIMDs are filled (8 waves each).  Prints cycles per
// wave-instruction per SIMD (4 = full rate for a 64-lane wave on a 16-lane SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
#define BODY(INS) REP8(REP8(asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
                                     INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
                                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));))
#define KERNEL(NAME, INS)                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t k, int it) \
    {                                                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < it; i++) { BODY(INS) }                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    }
#define BODY3(INS) REP8(REP8(asm volatile(INS " %0, %0, %8, %1\n" INS " %1, %1, %8, %2\n" INS " %2, %2, %8, %3\n" INS " %3, %3, %8, %4\n" \
                                      INS " %4, %4, %8, %5\n" INS " %5, %5, %8, %6\n" INS " %6, %6, %8, %7\n" INS " %7, %7, %8, %0\n" \
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));))
#define KERNEL3(NAME, INS)                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t k, int it) \
    {                                                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < it; i++) { BODY3(INS) }                                \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
    }
KERNEL(k_add, "v_add_u32")
KERNEL(k_sub, "v_sub_u32")
KERNEL(k_and, "v_and_b32")
KERNEL(k_or, "v_or_b32")
KERNEL(k_xor, "v_xor_b32")
KERNEL(k_lshr, "v_lshrrev_b32")
KERNEL(k_max, "v_max_u32")
KERNEL(k_min, "v_min_i32")
KERNEL(k_addf, "v_add_f32")
KERNEL(k_maxf, "v_max_f32")
KERNEL(k_bcnt, "v_bcnt_u32_b32")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32")
KERNEL(k_cndmask, "v_cndmask_b32")
KERNEL3(k_add3, "v_add3_u32")
KERNEL3(k_lshl_add, "v_lshl_add_u32")
KERNEL3(k_lshl_or, "v_lshl_or_b32")
KERNEL3(k_and_or, "v_and_or_b32")
KERNEL3(k_or3, "v_or3_b32")
KERNEL3(k_bfe, "v_bfe_u32")
KERNEL3(k_bfi, "v_bfi_b32")
KERNEL3(k_perm, "v_perm_b32")
KERNEL3(k_alignbit, "v_alignbit_b32")
KERNEL3(k_dot4, "v_dot4_u32_u8")
KERNEL3(k_mad24, "v_mad_u32_u24")
KERNEL3(k_fma, "v_fma_f32")
KERNEL3(k_sad, "v_sad_u32")
KERNEL3(k_med3, "v_med3_u32")
KERNEL(k_mul_lo, "v_mul_lo_u32")
KERNEL(k_mul_hi, "v_mul_hi_u32")
KERNEL(k_mul24, "v_mul_u32_u24")
KERNEL(k_lshl, "v_lshlrev_b32")
KERNEL(k_mulf32, "v_mul_f32")
#define BODY64(INS) REP8(REP8(asm volatile(INS " %0, %0, %4\n" INS " %1, %1, %4\n" INS " %2, %2, %4\n" INS " %3, %3, %4\n" \
                                       INS " %0, %0, %4\n" INS " %1, %1, %4\n" INS " %2, %2, %4\n" INS " %3, %3, %4\n" \
                                       : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(kd));))
#define KERNEL64(NAME, INS)                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t k, int it) \
    {                                                                              \
        double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, kd = 1.0 + 1e-9 * k; \
        for (int i = 0; i < it; i++) { BODY64(INS) }                               \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3);       \
    }
KERNEL64(k_mulf64, "v_mul_f64")
KERNEL64(k_addf64, "v_add_f64")
__global__ __launch_bounds__(256) void k_cvt64(uint32_t *out, uint32_t k, int it)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0, d1, d2, d3;
    for (int i = 0; i < it; i++) {
        REP8(REP8(asm volatile("v_cvt_f64_u32 %0, %4\nv_cvt_f64_u32 %1, %5\nv_cvt_f64_u32 %2, %6\nv_cvt_f64_u32 %3, %7\n"
                               "v_cvt_f64_u32 %0, %4\nv_cvt_f64_u32 %1, %5\nv_cvt_f64_u32 %2, %6\nv_cvt_f64_u32 %3, %7\n"
                               : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));))
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3);
}
template <typename K>
void run(const char *name, K kern, uint32_t *out, double mhz, int simds)
{
    const int it = 64, blocks = simds * 2;   // 8 waves per SIMD: 2 blocks of 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 3u, it);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 3u, it);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 8.0 * it * 64 * 8;   // wave-instructions per SIMD
    printf("%-16s %8.3f ms  %.2f cycles per wave-instruction at %.0f MHz\n", name, ms, ms * 1e-3 * mhz * 1e6 / per_simd, mhz);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double mhz = p.clockRate / 1000.0;
    const int simds = p.multiProcessorCount * 4;
    uint32_t *out;
    hipMalloc(&out, (size_t)simds * 2 * 256 * 4);
    run("v_add_u32", k_add, out, mhz, simds);
    run("v_sub_u32", k_sub, out, mhz, simds);
    run("v_and_b32", k_and, out, mhz, simds);
    run("v_or_b32", k_or, out, mhz, simds);
    run("v_xor_b32", k_xor, out, mhz, simds);
    run("v_lshrrev_b32", k_lshr, out, mhz, simds);
    run("v_max_u32", k_max, out, mhz, simds);
    run("v_min_i32", k_min, out, mhz, simds);
    run("v_add_f32", k_addf, out, mhz, simds);
    run("v_max_f32", k_maxf, out, mhz, simds);
    run("v_bcnt_u32_b32", k_bcnt, out, mhz, simds);
    run("v_mbcnt_lo", k_mbcnt, out, mhz, simds);
    run("v_cndmask_b32", k_cndmask, out, mhz, simds);
    run("v_add3_u32", k_add3, out, mhz, simds);
    run("v_lshl_add_u32", k_lshl_add, out, mhz, simds);
    run("v_lshl_or_b32", k_lshl_or, out, mhz, simds);
    run("v_and_or_b32", k_and_or, out, mhz, simds);
    run("v_or3_b32", k_or3, out, mhz, simds);
    run("v_bfe_u32", k_bfe, out, mhz, simds);
    run("v_bfi_b32", k_bfi, out, mhz, simds);
    run("v_perm_b32", k_perm, out, mhz, simds);
    run("v_alignbit_b32", k_alignbit, out, mhz, simds);
    run("v_dot4_u32_u8", k_dot4, out, mhz, simds);
    run("v_mad_u32_u24", k_mad24, out, mhz, simds);
    run("v_fma_f32", k_fma, out, mhz, simds);
    run("v_sad_u32", k_sad, out, mhz, simds);
    run("v_med3_u32", k_med3, out, mhz, simds);
    run("v_lshlrev_b32", k_lshl, out, mhz, simds);
    run("v_mul_u32_u24", k_mul24, out, mhz, simds);
    run("v_mul_lo_u32", k_mul_lo, out, mhz, simds);
    run("v_mul_hi_u32", k_mul_hi, out, mhz, simds);
    run("v_mul_f32", k_mulf32, out, mhz, simds);
    run("v_mul_f64", k_mulf64, out, mhz, simds);
    run("v_add_f64", k_addf64, out, mhz, simds);
    run("v_cvt_f64_u32", k_cvt64, out, mhz, simds);
    return 0;
}
