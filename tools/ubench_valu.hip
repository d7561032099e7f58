// tools/ubench_valu.hip — issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the pair
// kernel uses, measured on the box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_valu.hip && /tmp/ub
// Each kernel runs N_IT iterations of 16 independent chains of one instruction (inline asm so that the
// compiler cannot fuse or remove them); 2048 blocks x 256 threads = 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N_IT = 2000;

#define DEF_KERNEL64(NAME, ASM)                                                              \
    __global__ void NAME(double *out, double a, double b)                                    \
    {                                                                                        \
        double r[16];                                                                        \
        for (int i = 0; i < 16; ++i) r[i] = a + i + threadIdx.x;                             \
        for (int it = 0; it < N_IT; ++it) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(b)); \
        }                                                                                    \
        double s = 0;                                                                        \
        for (int i = 0; i < 16; ++i) s += r[i];                                              \
        if (s == 12345.678) out[0] = s;                                                      \
    }

#define DEF_KERNEL32(NAME, ASM)                                                              \
    __global__ void NAME(double *out, double a, double b)                                    \
    {                                                                                        \
        float r[16];                                                                         \
        float fb = (float)b;                                                                 \
        for (int i = 0; i < 16; ++i) r[i] = (float)a + i + threadIdx.x;                      \
        for (int it = 0; it < N_IT; ++it) {                                                  \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(fb)); \
        }                                                                                    \
        float s = 0;                                                                         \
        for (int i = 0; i < 16; ++i) s += r[i];                                              \
        if (s == 12345.678f) out[0] = s;                                                     \
    }

DEF_KERNEL64(k_add_f64, "v_add_f64 %0, %0, %1")
DEF_KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
DEF_KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
DEF_KERNEL64(k_min_f64, "v_min_f64 %0, %0, %1")
DEF_KERNEL64(k_min_f64_abs, "v_min_f64 %0, |%0|, |%1|")
DEF_KERNEL64(k_add_f64_abs, "v_add_f64 %0, |%0|, -%1")
DEF_KERNEL64(k_cmp_f64, "v_cmp_gt_f64 vcc, %0, %1")
DEF_KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
DEF_KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
DEF_KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
DEF_KERNEL32(k_fract_f32, "v_fract_f32 %0, %0")
DEF_KERNEL32(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF_KERNEL32(k_min_f32_abs, "v_min_f32 %0, |%0|, |%1|")
DEF_KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
DEF_KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL32(k_cmp_f32, "v_cmp_gt_f32 vcc, %0, %1")

__global__ void k_cvt_f32_f64(double *out, double a, double b)
{
    double r[16];
    float f[16];
    for (int i = 0; i < 16; ++i) r[i] = a + i + threadIdx.x;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(r[i]));
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += f[i];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
double run(K kern, const char *name, double *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d_out, 1.5, 0.999999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d_out, 1.5, 0.999999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = waves/SIMD * N_IT * 16 ; waves = 2048*4 over 1024 SIMDs = 8 per SIMD
    const double insts_per_simd = 8.0 * N_IT * 16;
    const double cyc = ms * 1e-3 * 2.4e9 / insts_per_simd;
    printf("%-16s %8.3f ms   %.2f cycles/inst/SIMD (at 2.4 GHz)\n", name, ms, cyc);
    return cyc;
}

int main()
{
    double *d;
    hipMalloc(&d, 64);
    run(k_add_f64, "v_add_f64", d);
    run(k_add_f64_abs, "v_add_f64 |a|-b", d);
    run(k_mul_f64, "v_mul_f64", d);
    run(k_fma_f64, "v_fma_f64", d);
    run(k_min_f64, "v_min_f64", d);
    run(k_min_f64_abs, "v_min_f64 |a||b|", d);
    run(k_cmp_f64, "v_cmp_gt_f64", d);
    run(k_cvt_f32_f64, "v_cvt_f32_f64", d);
    run(k_add_f32, "v_add_f32", d);
    run(k_fma_f32, "v_fma_f32", d);
    run(k_min_f32_abs, "v_min_f32 abs", d);
    run(k_sqrt_f32, "v_sqrt_f32", d);
    run(k_fract_f32, "v_fract_f32", d);
    run(k_cvt_i32_f32, "v_cvt_i32_f32", d);
    run(k_cmp_f32, "v_cmp_gt_f32", d);
    run(k_lshl_add, "v_lshl_add_u32", d);
    run(k_mul_lo, "v_mul_lo_u32", d);
    return 0;
}
