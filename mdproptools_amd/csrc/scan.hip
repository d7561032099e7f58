// scan.hip — cumulative trapezoid (G4) for gfx950.
//
// Replaces scipy's cumulative trapezoid as called at dynamical/viscosity.py:151 and
// dynamical/conductivity.py:231 of the reference: inc[m] = dx*(y[m]+y[m+1])/2, I[k] = sum_{m<k} inc[m].
// HBM-bound (16 bytes per sample). Three-phase scan: per-block inclusive scan of 2048 increments
// (8 per lane sequentially, then a wave-shuffle scan of the lane totals), a scan of the block
// totals, then the offsets are added. The summation order differs from scipy's sequential cumsum,
// so agreement is to rounding (tests: rtol 1e-9 with an absolute floor of 1e-12*max|I|).
#include <algorithm>

#include "ctx.h"

#pragma clang fp contract(off)

namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_PER = 8;
constexpr int SC_BLOCK = SC_THREADS * SC_PER;

// LDS index of sample / result i of a block: one pad double per 8, so that a lane walking ITS 8 consecutive entries
// (stride 9 between lanes) and the block walking consecutive entries (coalesced global side) both spread over the banks
__device__ __forceinline__ int sc_pad(int i) { return i + (i >> 3); }

// n_inc = n-1 increments per series. out_series points at the first integral value (after the
// optional leading zero). block_tot [n_series][n_blocks].
// Round 3: the samples of a block come in through coalesced loads into LDS (the lane that owns increments 8 t .. 8 t + 7
// used to read its 9 samples straight from global memory: 64-byte strides between lanes, eight partial sweeps of the
// same lines), the lane totals are scanned with wave shuffles (six steps, no barrier) and the four wave totals in
// order, and the results leave through LDS again, coalesced: three barriers per block instead of seventeen.
__global__ __launch_bounds__(SC_THREADS) void trap_scan_local_kernel(
    const double *__restrict__ y, double *__restrict__ out, double *__restrict__ block_tot,
    long long n, long long out_stride, int lead, double dx, int n_blocks)
{
    __shared__ double s_v[SC_BLOCK + SC_BLOCK / 8 + 2];
    __shared__ double s_w[SC_THREADS / 64];
    const int series = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *ys = y + (size_t)series * n;
    double *os = out + (size_t)series * out_stride + lead;
    const long long n_inc = n - 1;
    const long long blk0 = (long long)blockIdx.x * SC_BLOCK;
    // samples blk0 .. blk0 + 2048 (one more than increments)
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const int i = u * SC_THREADS + tid;
        s_v[sc_pad(i)] = blk0 + i < n ? ys[blk0 + i] : 0.0;
    }
    if (tid == 0) s_v[sc_pad(SC_BLOCK)] = blk0 + SC_BLOCK < n ? ys[blk0 + SC_BLOCK] : 0.0;
    __syncthreads();
    double v[SC_PER];
    double run = 0.0;
    {
        const int i0 = tid * SC_PER;
        double y0 = s_v[sc_pad(i0)];
#pragma unroll
        for (int u = 0; u < SC_PER; ++u) {
            const double y1 = s_v[sc_pad(i0 + u + 1)];
            const double inc = blk0 + i0 + u < n_inc ? dx * (y1 + y0) / 2.0 : 0.0;  // scipy: d * (y[1:] + y[:-1]) / 2.0
            run += inc;
            v[u] = run;
            y0 = y1;
        }
    }
    // inclusive scan of the lane totals inside the wave, then the wave totals in order
    double incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();  // (also: every lane has read its samples, s_v can take the results)
    double before = incl - run;
    for (int w = 0; w < wv; ++w) before += s_w[w];
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) s_v[sc_pad(tid * SC_PER + u)] = before + v[u];
    if (tid == SC_THREADS - 1) block_tot[(size_t)series * n_blocks + blockIdx.x] = before + run;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const int i = u * SC_THREADS + tid;
        if (blk0 + i < n_inc) os[blk0 + i] = s_v[sc_pad(i)];
    }
}

// exclusive scan of the block totals of one series per block: every lane takes a contiguous share in order, the 256
// share totals are scanned (wave shuffles, then the four wave totals in order). (Round 2: ONE lane per series walked
// the totals through dependent global loads: 64 us for 489 blocks.)
__global__ __launch_bounds__(SC_THREADS) void trap_scan_blocks_kernel(double *__restrict__ block_tot, int n_blocks)
{
    __shared__ double s_w[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double *p = block_tot + (size_t)blockIdx.x * n_blocks;
    const int share = (n_blocks + SC_THREADS - 1) / SC_THREADS;
    const int lo = min(tid * share, n_blocks), hi = min(lo + share, n_blocks);
    double run = 0.0;
    for (int b = lo; b < hi; ++b) run += p[b];
    double incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    double acc = incl - run;
    for (int w = 0; w < wv; ++w) acc += s_w[w];
    for (int b = lo; b < hi; ++b) {
        const double t = p[b];
        p[b] = acc;
        acc += t;
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

// y device [n_series][n] -> d_out device [n_series][n - 1 + lead]; everything on the context's stream
int mdhip_cumtrapz_enqueue(mdhip_ctx *ctx, int64_t n, int n_series, const double *d_y, double dx, int lead, double *d_out)
{
    const int64_t out_stride = n - 1 + lead;
    const int n_blocks = (int)((n - 1 + SC_BLOCK - 1) / SC_BLOCK);
    MD_WS(d_tot, double, WS_PART, (size_t)n_series * n_blocks * 8);
    hipLaunchKernelGGL(trap_scan_local_kernel, dim3((unsigned)n_blocks, (unsigned)n_series),
                       dim3(SC_THREADS), 0, ctx->stream, d_y, d_out, d_tot, (long long)n,
                       (long long)out_stride, lead, dx, n_blocks);
    hipLaunchKernelGGL(trap_scan_blocks_kernel, dim3((unsigned)n_series), dim3(SC_THREADS), 0, ctx->stream, d_tot,
                       n_blocks);
    hipLaunchKernelGGL(trap_scan_add_kernel, dim3((unsigned)n_blocks, (unsigned)n_series),
                       dim3(SC_THREADS), 0, ctx->stream, d_out, d_tot, (long long)n,
                       (long long)out_stride, lead, n_blocks);
    MD_HIP(hipGetLastError());
    return MDHIP_OK;
}

static int cumtrapz_impl(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                         int leading_zero, double *out, int out_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n >= 0 && n_series >= 0, "negative sizes");
    if (n_series == 0 || n == 0) return cs.end();
    MD_REQUIRE(y && out, "NULL array");
    MD_REQUIRE(n_series <= 65535, "at most 65535 series per call");
    const int lead = leading_zero ? 1 : 0;
    const int64_t out_stride = n - 1 + lead;
    MD_HIP(hipSetDevice(ctx->device));
    if (n == 1) {
        if (lead) {
            const int rc0 = mdhip_zero_result(ctx, out, (size_t)n_series * 8, out_on_device);
            if (rc0) return rc0;
        }
        return cs.end();
    }
    int rc;
    const double *d_y = (const double *)mdhip_stage(ctx, WS_XYZ_I, y, (size_t)n_series * n * 8, on_device, &rc);
    if (rc) return rc;
    const size_t out_b = (size_t)n_series * out_stride * 8;
    double *d_out = out;
    if (!out_on_device) {
        d_out = (double *)mdhip_ws(ctx, WS_OUT, out_b);
        if (!d_out) return MDHIP_ENOMEM;
    }
    KernelTimer timer(ctx);
    ctx->last_kernel = "trap_scan_local_kernel";
    rc = mdhip_cumtrapz_enqueue(ctx, n, n_series, d_y, dx, lead, d_out);
    if (rc) return rc;
    timer.stop();
    if (!out_on_device) {
        rc = mdhip_result(cs, out, d_out, out_b, 0);
        if (rc) return rc;
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

extern "C" {

int mdhip_cumtrapz(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device,
                   double dx, int leading_zero, double *out)
{
    return cumtrapz_impl(ctx, n, n_series, y, on_device, dx, leading_zero, out, 0);
}

int mdhip_cumtrapz_dev(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                       int leading_zero, double *out_dev)
{
    return cumtrapz_impl(ctx, n, n_series, y, on_device, dx, leading_zero, out_dev, 1);
}

int mdhip_cumtrapz_async(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                         int leading_zero, double *out)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return cumtrapz_impl(ctx, n, n_series, y, on_device, dx, leading_zero, out, 0);
}

int mdhip_cumtrapz_dev_async(mdhip_ctx *ctx, int64_t n, int n_series, const double *y, int on_device, double dx,
                             int leading_zero, double *out_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return cumtrapz_impl(ctx, n, n_series, y, on_device, dx, leading_zero, out_dev, 1);
}

}  // extern "C"
