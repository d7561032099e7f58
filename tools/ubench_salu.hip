// tools/ubench_salu.hip — what scalar instructions and branches cost NEXT TO vector instructions on one SIMD:
// the pair kernels issue about one SALU/branch instruction per VALU instruction (exec masking per pair slot).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_salu tools/ubench_salu.hip && tools/_bin/ubench_salu
//
// Grid = 256 CUs x k blocks of 256 threads, k pinned through dynamic LDS (k waves per SIMD); HIP-event time of the
// launch; ns per loop trip per SIMD = t / (k * n_it) — compare the rows, the absolute clock does not matter.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define R16(X) X X X X X X X X X X X X X X X X
#define VOP "v_pk_fma_f32 %0, %1, %1, %0\n\t"
#define SOP "s_add_u32 %2, %2, 1\n\t"
#define SOP2 "s_and_b64 %3, %3, exec\n\t"
#define BR "s_cbranch_execz 1f\n\t"
#define MASK "v_cmp_gt_f32 vcc, %4, %5\n\ts_and_saveexec_b64 %3, vcc\n\ts_cbranch_execz 1f\n\t"
#define UNMASK "s_or_b64 exec, exec, %3\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void k_mix(float *sink, float a, int n_it)
{
    extern __shared__ unsigned char lds[];
    f32x2 v = {a + threadIdx.x, a};
    f32x2 w = {0.999f, 1.001f};
    unsigned s = 1;
    unsigned long long m = ~0ull;
    float big = 1e30f, x = a;
    for (int it = 0; it < n_it; ++it) {
        // (every operand the text writes is an in/out operand, and scc is declared clobbered: an input-only SGPR that
        // the text increments can be the register the compiler keeps the loop bound in)
        if (MODE == 0) asm volatile(R16(VOP) : "+v"(v), "+v"(w));
        if (MODE == 1) asm volatile(R16(SOP) : "+v"(v), "+v"(w), "+s"(s)::"scc");
        if (MODE == 2) asm volatile(R16(VOP SOP) : "+v"(v), "+v"(w), "+s"(s)::"scc");
        if (MODE == 3) asm volatile(R16(VOP SOP SOP2) : "+v"(v), "+v"(w), "+s"(s), "+s"(m)::"scc");
        if (MODE == 4) asm volatile(R16(VOP BR) "1:\n\t" : "+v"(v), "+v"(w));
        if (MODE == 5) asm volatile(R16(VOP SOP BR) "1:\n\t" : "+v"(v), "+v"(w), "+s"(s)::"scc");
        // a pair slot as the kernels have it: compare, mask, branch over, 6 vector ops, unmask  (8 VALU, 2 SALU, 1 branch)
        if (MODE == 6)
            asm volatile(R16(MASK VOP VOP VOP VOP VOP VOP "1:\n\t" UNMASK VOP) : "+v"(v), "+v"(w), "+s"(s), "+s"(m), "+v"(big), "+v"(x)::"vcc", "scc");
        // the same 8 vector ops without the masking
        if (MODE == 7) asm volatile(R16("v_cmp_gt_f32 vcc, %4, %5\n\t" VOP VOP VOP VOP VOP VOP VOP) : "+v"(v), "+v"(w), "+s"(s), "+s"(m), "+v"(big), "+v"(x)::"vcc");
    }
    if (v[0] + v[1] + s + (float)m == 12345.0f) sink[0] = v[0] + lds[0];
}

template <int MODE>
void run(const char *name, int k, double valu_per_trip, float *sink)
{
    const int n_it = 20000;
    const size_t lds = (size_t)(163840 / k) & ~size_t(1023);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mix<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mix<MODE>, dim3(256 * k), dim3(256), lds, 0, sink, 1.0f, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<MODE>, dim3(256 * k), dim3(256), lds, 0, sink, 1.0f, n_it);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns_trip = ms * 1e6 / ((double)k * n_it);
    printf("%-52s k=%d  %.1f ns per trip per SIMD", name, k, ns_trip);
    if (valu_per_trip > 0) printf("  = %.2f ns per VALU instruction", ns_trip / valu_per_trip);
    printf("\n");
    fflush(stdout);
}

int main()
{
    float *sink;
    (void)hipMalloc(&sink, 64);
    for (int k : {1, 2, 4, 6, 8}) {
        run<0>("16 v_pk_fma_f32", k, 16, sink);
        run<1>("16 s_add_u32", k, 0, sink);
        run<2>("16 x (v_pk_fma, s_add)", k, 16, sink);
        run<3>("16 x (v_pk_fma, s_add, s_and_b64)", k, 16, sink);
        run<4>("16 x (v_pk_fma, s_cbranch_execz not taken)", k, 16, sink);
        run<5>("16 x (v_pk_fma, s_add, s_cbranch_execz)", k, 16, sink);
        run<6>("16 x masked slot (cmp, saveexec, branch, 7 pk_fma, or)", k, 128, sink);
        run<7>("16 x the same 8 VALU without masking", k, 128, sink);
    }
    return 0;
}
