// scan.hip — cumulative trapezoid (G4) for gfx950.
//
// Replaces scipy's cumulative trapezoid as called at dynamical/viscosity.py:151 and
// dynamical/conductivity.py:231 of the reference: inc[m] = dx*(y[m]+y[m+1])/2, I[k] = sum_{m<k} inc[m].
// HBM-bound (16 bytes per sample). ONE pass (round 4): every block scans its 2048 increments in registers (8 per lane
// sequentially, a wave-shuffle scan of the lane totals, the four wave totals in order), publishes its total, adds up
// the totals of the blocks before it — in a FIXED order, so the result does not depend on timing — and writes
// local + offset: the samples are read once and the integrals written once (rounds 1-3: three launches, the local
// scans written, re-read and re-written: 2.0x the bytes). The summation order differs from scipy's sequential cumsum,
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

// The one-pass form. Tile t = series * n_blocks + block of 2048 increments; a launch covers the tiles [t0, t0 + grid),
// ONE per block, and the grid is at most what the chip holds at once (mdhip_cumtrapz_enqueue asks the occupancy
// calculator; longer inputs take several launches, a per-series carry goes from one to the next). Every block scans
// its tile in registers and publishes the tile's total; the block of the FIRST tile a series has in the launch then
// waits for the totals of that series' tiles in the launch, scans them — every lane a contiguous share in order, the
// 256 share totals by wave shuffles, the four wave sums in order: a fixed order, the result does not depend on timing —
// and publishes every tile's offset; a block waits for ITS offset (one lane polls one word), adds it and writes.
// Nobody waits before having published and the scanners wait for totals only: no cycle.
// How the words travel between the CUs. A total and an offset are ONE 64-bit word each, empty = SC_EMPTY (a NaN pattern
// no arithmetic produces; NaN results are stored as the canonical quiet NaN), written and polled with relaxed
// device-scope atomics — no flag beside the value, hence no release / acquire fence: on this GPU a device-scope fence
// writes back and invalidates the XCD's whole L2, and two of them per block made a first version of this kernel four
// times SLOWER than the three launches it replaces. Every block empties its two words again once it has its offset
// (the scanner has read every total before it publishes the first offset), so the next launch finds them empty;
// trap_scan_fill_kernel runs once per buffer.
// Measured build against build in one process (tools/ab_libs_scan.py, 3 x 1e6 samples): 25.8 us against 24.7 us for the
// three launches — the same time for half the bytes (PMC: profiles/pmc_secondary.json): what is left is the load phase,
// two device-scope hops of ~1.5 us each (total -> scanner -> offset) and the store phase, one after the other in every
// block of the launch at once; a tile-per-XCD mapping with L2-scope hops would shorten the hops but would make the
// RESULT depend on how blocks are dealt to XCDs, which this library does not rely on.
// The samples of a tile come in through coalesced loads into LDS, the results leave through LDS again, coalesced.
constexpr unsigned long long SC_EMPTY = 0x7ff4dead5ca1ab1eULL;

__device__ __forceinline__ void sc_publish(double *slot, double v)
{
    unsigned long long bits = __double_as_longlong(v);
    if (v != v) bits = 0x7ff8000000000000ULL;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ double sc_await(const double *slot)
{
    unsigned long long bits;
    while ((bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(slot), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT)) == SC_EMPTY)
        __builtin_amdgcn_s_sleep(2);
    return __longlong_as_double((long long)bits);
}

__global__ void trap_scan_fill_kernel(unsigned long long *__restrict__ w, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = SC_EMPTY;
}

__global__ __launch_bounds__(SC_THREADS) void trap_scan_onepass_kernel(
    const double *__restrict__ y, double *__restrict__ out, double *__restrict__ totals, double *__restrict__ offsets,
    double *__restrict__ carry, unsigned t0, unsigned t_end, long long n, long long out_stride, int lead, double dx,
    int n_blocks, double post_scale)
{
    __shared__ double s_v[SC_BLOCK + SC_BLOCK / 8 + 2];
    __shared__ double s_w[SC_THREADS / 64];
    __shared__ double s_off;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long n_inc = n - 1;
    const unsigned t = t0 + blockIdx.x, slot = blockIdx.x;
    if (t >= t_end) return;
    const int series = (int)(t / (unsigned)n_blocks), bid = (int)(t % (unsigned)n_blocks);
    const double *ys = y + (size_t)series * n;
    double *os = out + (size_t)series * out_stride + lead;
    const long long blk0 = (long long)bid * SC_BLOCK;
    // samples blk0 .. blk0 + 2048 (one more than increments)
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const int i = u * SC_THREADS + tid;
        s_v[sc_pad(i)] = blk0 + i < n ? __builtin_nontemporal_load(ys + blk0 + i) : 0.0;
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
    if (tid == SC_THREADS - 1) sc_publish(&totals[slot], before + run);
    __syncthreads();  // (s_w is free)
    if (bid == 0 || t == t0) {
        // this block scans the totals of its series' tiles in this launch: slots lo .. lo + m
        const unsigned hi_t = min(t_end, (unsigned)(series + 1) * (unsigned)n_blocks);
        const unsigned lo = slot;
        const int m = (int)(hi_t - t);
        const int share = (m + SC_THREADS - 1) / SC_THREADS;
        const int a = min(tid * share, m), e = min(a + share, m);
        double tot[SC_PER];  // (m <= the launch's blocks <= 8 per lane: scan_capacity is at most 8 blocks per CU x 256)
        double mine = 0.0;
#pragma unroll
        for (int q = 0; q < SC_PER; ++q) {
            tot[q] = a + q < e ? sc_await(&totals[lo + a + q]) : 0.0;
            mine += tot[q];
        }
        double sc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double up = __shfl_up(sc, d, 64);
            if (lane >= d) sc += up;
        }
        if (lane == 63) s_w[wv] = sc;
        __syncthreads();
        // what came before this launch (the first tile of the series: nothing)
        double acc = bid == 0 ? 0.0 : carry[series];
        acc += sc - mine;
        for (int w = 0; w < wv; ++w) acc += s_w[w];
#pragma unroll
        for (int q = 0; q < SC_PER; ++q) {
            if (a + q < e) sc_publish(&offsets[lo + a + q], acc);
            acc += tot[q];
        }
        if (tid == SC_THREADS - 1) carry[series] = acc;  // (read by the next launch of this call, behind this one)
    }
    if (tid == 0) {
        s_off = sc_await(&offsets[slot]);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&offsets[slot]), SC_EMPTY, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&totals[slot]), SC_EMPTY, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double off = s_off;
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) s_v[sc_pad(tid * SC_PER + u)] = (off + (before + v[u])) * post_scale;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < SC_PER; ++u) {
        const int i = u * SC_THREADS + tid;
        if (blk0 + i < n_inc) __builtin_nontemporal_store(s_v[sc_pad(i)], os + blk0 + i);
    }
    if (lead && bid == 0 && tid == 0) out[(size_t)series * out_stride] = 0.0;
}

}  // namespace

// y device [n_series][n] -> d_out device [n_series][n - 1 + lead]; everything on the context's stream
int mdhip_cumtrapz_enqueue(mdhip_ctx *ctx, int64_t n, int n_series, const double *d_y, double dx, int lead, double *d_out,
                           double post_scale)
{
    const int64_t out_stride = n - 1 + lead;
    const int n_blocks = (int)((n - 1 + SC_BLOCK - 1) / SC_BLOCK);
    const size_t total = (size_t)n_series * n_blocks;
    MD_REQUIRE(total < (1u << 30), "too many scan tiles (%zu)", total);
    // every block of a launch must be resident (see the kernel): at most what the occupancy calculator says the chip holds
    if (ctx->scan_capacity <= 0) {
        int per_cu = 0;
        MD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(trap_scan_onepass_kernel),
                                                            SC_THREADS, 0));
        ctx->scan_capacity = std::max(1, per_cu) * ctx->cu_count;
    }
    MD_REQUIRE(ctx->scan_capacity <= SC_PER * SC_THREADS, "internal: scan launches of more than %d blocks", SC_PER * SC_THREADS);
    const size_t cap = (size_t)ctx->scan_capacity;
    // per launch: totals | offsets, one word per block (empty between launches: the blocks see to that); carry: one
    // double per series, behind them (a buffer that has to grow for more series is emptied again)
    const bool fresh = ctx->ws[WS_SCAN].cap < (2 * cap + (size_t)n_series) * 8;
    MD_WS(d_ws, double, WS_SCAN, (2 * cap + (size_t)std::max(n_series, 64)) * 8);
    double *d_tot = d_ws, *d_off = d_ws + cap, *d_carry = d_ws + 2 * cap;
    if (fresh) {
        hipLaunchKernelGGL(trap_scan_fill_kernel, dim3((unsigned)((2 * cap + 255) / 256)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<unsigned long long *>(d_ws), (unsigned)(2 * cap));
    }
    for (size_t t0 = 0; t0 < total; t0 += cap) {
        const size_t t1 = std::min(total, t0 + cap);
        const unsigned g = (unsigned)(t1 - t0);
        hipLaunchKernelGGL(trap_scan_onepass_kernel, dim3(g), dim3(SC_THREADS), 0, ctx->stream, d_y, d_out, d_tot, d_off,
                           d_carry, (unsigned)t0, (unsigned)t1, (long long)n, (long long)out_stride, lead, dx, n_blocks,
                           post_scale);
    }
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
    ctx->last_kernel = "trap_scan_onepass_kernel";
    rc = mdhip_cumtrapz_enqueue(ctx, n, n_series, d_y, dx, lead, d_out, 1.0);
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
