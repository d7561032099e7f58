// tools/ubench_mfma64.hip — FP64 rates on the box: v_mfma_f64_16x16x4_f64 alone, v_fma_f64 alone, and both together
// (in one instruction stream, and as separate waves of one SIMD), as TFLOP/s over the whole chip.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_mfma64 tools/ubench_mfma64.hip && tools/_bin/ubench_mfma64
//
// Every kernel is a grid of 256 CUs x `k` blocks of 256 threads (k waves per SIMD); HIP-event time of the launch.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: MFMA only; 1: FMA only; 2: both in every wave (4 MFMA + `nf` FMA per trip); 3: waves 0..k/2 MFMA, rest FMA
template <int MODE, int NF>
__global__ __launch_bounds__(256) void rate_kernel(double *sink, double a0, double b0, int n_it, int split)
{
    d4 acc[4];
    double f[16];
    for (int i = 0; i < 4; ++i) acc[i] = d4{a0, a0 + 1, a0 + 2, a0 + 3};
    for (int i = 0; i < 16; ++i) f[i] = a0 + i + threadIdx.x;
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    const bool mf = MODE == 0 || MODE == 2 || (MODE == 3 && (int)(blockIdx.x % split) < split / 2);
    const bool fm = MODE == 1 || MODE == 2 || (MODE == 3 && !mf);
    for (int it = 0; it < n_it; ++it) {
        if (mf) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (fm) {
#pragma unroll
            for (int i = 0; i < NF; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i % 16]) : "v"(b), "v"(a));
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    if (s == 12345.678) sink[0] = s;
}

template <int MODE, int NF>
void run(const char *name, int k, int n_it, double *sink)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * k;
    hipLaunchKernelGGL((rate_kernel<MODE, NF>), dim3(grid), dim3(256), 0, 0, sink, 1.0, 0.5, 10, k);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE, NF>), dim3(grid), dim3(256), 0, 0, sink, 1.0, 0.5, n_it, k);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * 4;
    double n_mf = 0, n_fm = 0;
    if (MODE == 0 || MODE == 2) n_mf = waves;
    if (MODE == 1 || MODE == 2) n_fm = waves;
    if (MODE == 3) n_mf = waves / 2, n_fm = waves / 2;
    const double fl_mf = n_mf * n_it * 4.0 * 2048.0, fl_fm = n_fm * n_it * (double)NF * 128.0;
    printf("%-34s k=%d  %.3f ms  mfma %.1f TF  fma %.1f TF  total %.1f TF\n", name, k, ms, fl_mf / ms * 1e-9,
           fl_fm / ms * 1e-9, (fl_mf + fl_fm) / ms * 1e-9);
}

int main()
{
    double *sink;
    hipMalloc(&sink, 8);
    for (int k : {1, 2, 4}) {
        run<0, 0>("mfma_f64_16x16x4 only", k, 20000, sink);
        run<1, 64>("v_fma_f64 only", k, 20000, sink);
        run<2, 16>("both, one stream, 4 mfma + 16 fma", k, 20000, sink);
        run<2, 32>("both, one stream, 4 mfma + 32 fma", k, 20000, sink);
        run<2, 64>("both, one stream, 4 mfma + 64 fma", k, 20000, sink);
        if (k >= 2) run<3, 64>("both, separate waves", k, 20000, sink);
    }
    return 0;
}
