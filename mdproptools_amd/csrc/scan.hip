// scan.hip — cumulative trapezoid (G4) for gfx950.
//
// Replaces scipy's cumulative trapezoid as called at dynamical/viscosity.py:151 and
// dynamical/conductivity.py:231 of the reference: inc[m] = dx*(y[m]+y[m+1])/2, I[k] = sum_{m<k} inc[m].
// HBM-bound (16 bytes per sample). Three-phase scan: per-block inclusive scan of 2048 increments
// (8 per lane sequentially, then a wave/LDS scan of the lane totals), an ordered scan of the block
// totals, then the offsets are added. The summation order differs from scipy's sequential cumsum,
// so agreement is to rounding (tests: rtol 1e-9 with an absolute floor of 1e-12*max|I|).
#include <algorithm>

#include "ctx.h"

#pragma clang fp contract(off)

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_PER = 8;
constexpr int SC_BLOCK = SC_THREADS * SC_PER;

__device__ __forceinline__ double trap_inc(const double *__restrict__ y, long long m, double dx)
{
    return dx * (y[m + 1] + y[m]) / 2.0;  // scipy: d * (y[1:] + y[:-1]) / 2.0
}

// n_inc = n-1 increments per series. out_series points at the first integral value (after the
// optional leading zero). block_tot [n_series][n_blocks].
__global__ __launch_bounds__(SC_THREADS) void trap_scan_local_kernel(
    const double *__restrict__ y, double *__restrict__ out, double *__restrict__ block_tot,
    long long n, long long out_stride, int lead, double dx, int n_blocks)
{
    __shared__ double s_tot[SC_THREADS];
    const int series = blockIdx.y;
    const double *ys = y + (size_t)series * n;
    double *os = out + (size_t)series * out_stride + lead;
    const long long n_inc = n - 1;
    const long long base = (long long)blockIdx.x * SC_BLOCK + (long long)threadIdx.x * SC_PER;
    double v[SC_PER];
    double run = 0.0;
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const long long m = base + u;
        const double inc = m < n_inc ? trap_inc(ys, m, dx) : 0.0;
        run += inc;
        v[u] = run;
    }
    s_tot[threadIdx.x] = run;
    __syncthreads();
    // Hillis-Steele inclusive scan of the 256 lane totals
    for (int d = 1; d < SC_THREADS; d <<= 1) {
        const double add = (int)threadIdx.x >= d ? s_tot[threadIdx.x - d] : 0.0;
        __syncthreads();
        s_tot[threadIdx.x] += add;
        __syncthreads();
    }
    const double before = threadIdx.x ? s_tot[threadIdx.x - 1] : 0.0;
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const long long m = base + u;
        if (m < n_inc) os[m] = before + v[u];
    }
    if (threadIdx.x == SC_THREADS - 1) block_tot[(size_t)series * n_blocks + blockIdx.x] = s_tot[SC_THREADS - 1];
}

// exclusive scan of the block totals, one lane per series (n_blocks is small: n/2048)
__global__ void trap_scan_blocks_kernel(double *__restrict__ block_tot, int n_blocks, int n_series)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_series) return;
    double run = 0.0;
    double *p = block_tot + (size_t)s * n_blocks;
    for (int b = 0; b < n_blocks; ++b) {
        const double t = p[b];
        p[b] = run;
        run += t;
    }
}

__global__ __launch_bounds__(SC_THREADS) void trap_scan_add_kernel(
    double *__restrict__ out, const double *__restrict__ block_tot, long long n, long long out_stride,
    int lead, int n_blocks)
{
    const int series = blockIdx.y;
    double *os = out + (size_t)series * out_stride + lead;
    const double off = block_tot[(size_t)series * n_blocks + blockIdx.x];
    const long long n_inc = n - 1;
    const long long base = (long long)blockIdx.x * SC_BLOCK + threadIdx.x;
    if (blockIdx.x == 0) {
        if (lead && threadIdx.x == 0) out[(size_t)series * out_stride] = 0.0;
        return;
    }
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const long long m = base + (long long)u * SC_THREADS;
        if (m < n_inc) os[m] += off;
    }
}

}  // namespace

extern "C" {

int mdhip_cumtrapz(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device,
                   double dx, int leading_zero, double *out)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(n >= 0 && n_series >= 0, "negative sizes");
    if (n_series == 0 || n == 0) return MDHIP_OK;
    MD_REQUIRE(y && out, "NULL array");
    MD_REQUIRE(n_series <= 65535, "at most 65535 series per call");
    const int lead = leading_zero ? 1 : 0;
    const int64_t out_stride = n - 1 + lead;
    if (n == 1) {
        if (lead)
            for (int s = 0; s < n_series; ++s) out[s] = 0.0;
        return MDHIP_OK;
    }
    MD_HIP(hipSetDevice(ctx->device));
    int rc;
    const double *d_y = (const double *)mdhip_stage(ctx, WS_XYZ_I, y, (size_t)n_series * n * 8, on_device, &rc);
    if (rc) return rc;
    const int n_blocks = (int)((n - 1 + SC_BLOCK - 1) / SC_BLOCK);
    const size_t out_b = (size_t)n_series * out_stride * 8;
    MD_WS(d_out, double, WS_OUT, out_b);
    MD_WS(d_tot, double, WS_PART, (size_t)n_series * n_blocks * 8);
    KernelTimer timer(ctx);
    ctx->last_kernel = "trap_scan_local_kernel";
    hipLaunchKernelGGL(trap_scan_local_kernel, dim3((unsigned)n_blocks, (unsigned)n_series),
                       dim3(SC_THREADS), 0, ctx->stream, d_y, d_out, d_tot, (long long)n,
                       (long long)out_stride, lead, dx, n_blocks);
    hipLaunchKernelGGL(trap_scan_blocks_kernel, dim3((unsigned)((n_series + 63) / 64)), dim3(64), 0,
                       ctx->stream, d_tot, n_blocks, n_series);
    hipLaunchKernelGGL(trap_scan_add_kernel, dim3((unsigned)n_blocks, (unsigned)n_series),
                       dim3(SC_THREADS), 0, ctx->stream, d_out, d_tot, (long long)n,
                       (long long)out_stride, lead, n_blocks);
    timer.stop();
    MD_HIP(hipGetLastError());
    MD_HIP(hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    MD_HIP(hipStreamSynchronize(ctx->stream));
    timer.collect();
    return MDHIP_OK;
}

}  // extern "C"
