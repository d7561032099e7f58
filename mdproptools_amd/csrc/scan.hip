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
// ONE per block. Every block scans its tile in registers, publishes the tile's total, then adds up the totals of the
// tiles of ITS series that come BEFORE it in the launch — every lane a contiguous share in order, the 256 share sums by
// wave shuffles, the four wave sums in order: a fixed order, the result does not depend on timing — adds that offset
// and writes. Waits go BACKWARDS only (round 5): a block waits for blocks with a smaller index, never for a
// later one, so the launch needs no co-residency — with blocks dispatched in index order (what the hardware does; HIP
// does not promise it) the lowest unfinished block never waits, and by induction everything drains, whatever else
// shares the GPU (a second context or process, a CU mask). Round 4's form (the first tile of a series scanned the
// totals of the LATER tiles and handed out offsets) was correct only with the whole grid resident. Should the dispatch
// order ever differ, a poll gives up after ~2 s, raises the stall word, every block runs out, and the host fails the
// call (MDHIP_EHIP) instead of hanging the GPU.
// How the words travel between the CUs. A total is ONE 64-bit word, empty = SC_EMPTY (a NaN pattern no arithmetic
// produces; NaN results are stored as the canonical quiet NaN), written and polled with relaxed device-scope atomics —
// no flag beside the value, hence no release / acquire fence: on this GPU a device-scope fence writes back and
// invalidates the XCD's whole L2, and two of them per block made a first version of this kernel four times SLOWER than
// the three launches it replaced. A total is read by every later block of its series, so nobody can empty it inside the
// launch: there are TWO sets of words, launches alternate between them, and the blocks of a launch empty the set the
// launch BEFORE them used (complete by stream order) — block b the words b, b + grid, b + 2 grid, ...; a fill kernel
// runs once per buffer (and after a stall).
// The samples of a tile come in through coalesced loads into LDS, the results leave through LDS again, coalesced.
constexpr unsigned long long SC_EMPTY = 0x7ff4dead5ca1ab1eULL;
constexpr unsigned SC_SPIN_MAX = 1u << 21;  // polls of ~1 us each before a block gives up

__device__ __forceinline__ void sc_publish(double *slot, double v)
{
    unsigned long long bits = __double_as_longlong(v);
    if (v != v) bits = 0x7ff8000000000000ULL;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -> the word once it is there; gives up (-> 0, *stall raised) after SC_SPIN_MAX polls or when another block has
__device__ __forceinline__ double sc_await(const double *slot, unsigned *stall)
{
    unsigned long long bits;
    unsigned spin = 0;
    while ((bits = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(slot), __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT)) == SC_EMPTY) {
        __builtin_amdgcn_s_sleep(2);
        ++spin;
        if ((spin & 1023u) == 0u && __hip_atomic_load(stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return 0.0;
        if (spin >= SC_SPIN_MAX) {
            __hip_atomic_store(stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 0.0;
        }
    }
    return __longlong_as_double((long long)bits);
}

__global__ void trap_scan_fill_kernel(unsigned long long *__restrict__ w, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = SC_EMPTY;
}

__global__ __launch_bounds__(SC_THREADS) void trap_scan_onepass_kernel(
    const double *__restrict__ y, double *__restrict__ out, double *__restrict__ totals, double *__restrict__ other,
    unsigned n_words, const double *__restrict__ carry_in, double *__restrict__ carry_out, unsigned *__restrict__ stall,
    unsigned t0, unsigned t_end, long long n,
    long long out_stride, int lead, double dx, int n_blocks, double post_scale)
{
    __shared__ double s_v[SC_BLOCK + SC_BLOCK / 8 + 2];
    __shared__ double s_w[SC_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long n_inc = n - 1;
    const unsigned t = t0 + blockIdx.x, slot = blockIdx.x;
    // the words the launch before this one used: empty again for the launch after this one
    for (unsigned w = slot * SC_THREADS + tid; w < n_words; w += gridDim.x * SC_THREADS)
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(other) + w, SC_EMPTY, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
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
    double tile_total = 0.0;
    if (tid == SC_THREADS - 1) {
        tile_total = before + run;
        sc_publish(&totals[slot], tile_total);
    }
    __syncthreads();  // (s_w is free)
    // the tiles of this series before this one in the launch: slots lo .. slot - 1
    const unsigned first_t = max(t0, (unsigned)series * (unsigned)n_blocks);
    const unsigned lo = first_t - t0;
    const int m = (int)(t - first_t);  // < the launch's blocks <= SC_PER * SC_THREADS
    const int share = (m + SC_THREADS - 1) / SC_THREADS;
    const int a = min(tid * share, m), e = min(a + share, m);
    double mine = 0.0;
#pragma unroll
    for (int q = 0; q < SC_PER; ++q)
        if (a + q < e) mine += sc_await(&totals[lo + a + q], stall);
    double sc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double up = __shfl_up(sc, d, 64);
        if (lane >= d) sc += up;
    }
    if (lane == 63) s_w[wv] = sc;
    __syncthreads();
    // what came before this launch (a series that starts inside it: nothing), then the four wave sums in order
    // (round 6, ADVICE r05: the carry words alternate by launch parity like the totals — this launch READS what the launch
    // before it wrote and WRITES the other set; with one set the last tile's store below raced with these loads whenever a
    // continued series filled the whole launch)
    double off = first_t > (unsigned)series * (unsigned)n_blocks ? carry_in[series] : 0.0;
    for (int w = 0; w < SC_THREADS / 64; ++w) off += s_w[w];
    // (the launch's last tile: what the next launch of this call starts its series from — behind this one on the stream)
    if (t == t_end - 1 && tid == SC_THREADS - 1) carry_out[series] = off + tile_total;
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

// y device [n_series][n] -> d_out device [n_series][n - 1 + lead]; everything on the context's stream. Must run inside a
// call (ctx->cur): the stall word comes back through the call's pinned staging and is looked at when it completes.
int mdhip_cumtrapz_enqueue(mdhip_ctx *ctx, int64_t n, int n_series, const double *d_y, double dx, int lead, double *d_out,
                           double post_scale)
{
    const int64_t out_stride = n - 1 + lead;
    const int n_blocks = (int)((n - 1 + SC_BLOCK - 1) / SC_BLOCK);
    const size_t total = (size_t)n_series * n_blocks;
    MD_REQUIRE(total < (1u << 30), "too many scan tiles (%zu)", total);
    // tiles per launch: a block adds up at most SC_PER totals per lane (no residency condition: the waits go backwards)
    const size_t cap = (size_t)SC_PER * SC_THREADS;
    // two sets of totals (launches alternate, see the kernel) | stall word (a line of its own) | carry: two sets of one
    // double per series, alternating with the totals (a buffer that has to grow for more series is emptied again)
    const size_t n_carry = (size_t)std::max(n_series, 64);
    const size_t words = 2 * cap + 16 + 2 * n_carry;
    const bool fresh = ctx->ws[WS_SCAN].cap < words * 8 || ctx->scan_capacity <= 0;
    MD_WS(d_ws, double, WS_SCAN, words * 8);
    double *d_set[2] = {d_ws, d_ws + cap};
    unsigned *d_stall = reinterpret_cast<unsigned *>(d_ws + 2 * cap);
    double *d_carry[2] = {d_ws + 2 * cap + 16, d_ws + 2 * cap + 16 + n_carry};
    if (fresh) {
        hipLaunchKernelGGL(trap_scan_fill_kernel, dim3((unsigned)((2 * cap + 255) / 256)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<unsigned long long *>(d_ws), (unsigned)(2 * cap));
        MD_HIP(hipMemsetAsync(d_stall, 0, 128, ctx->stream));
        ctx->scan_capacity = (int)cap;
        ctx->scan_parity = 0;
    }
    for (size_t t0 = 0; t0 < total; t0 += cap) {
        const size_t t1 = std::min(total, t0 + cap);
        const unsigned g = (unsigned)(t1 - t0);
        const int par = ctx->scan_parity;
        ctx->scan_parity ^= 1;
        hipLaunchKernelGGL(trap_scan_onepass_kernel, dim3(g), dim3(SC_THREADS), 0, ctx->stream, d_y, d_out, d_set[par],
                           d_set[par ^ 1], (unsigned)cap, d_carry[par ^ 1], d_carry[par], d_stall, (unsigned)t0, (unsigned)t1,
                           (long long)n,
                           (long long)out_stride, lead, dx, n_blocks, post_scale);
    }
    MD_HIP(hipGetLastError());
    if (ctx->cur) {
        MD_PIN(h_stall, unsigned, 4);
        *h_stall = 0u;
        MD_HIP(hipMemcpyAsync(h_stall, d_stall, 4, hipMemcpyDeviceToHost, ctx->stream));
        ctx->cur->steps.emplace_back([ctx, h_stall]() {
            if (*h_stall == 0u) return (int)MDHIP_OK;
            ctx->scan_capacity = 0;  // the words are in an unknown state: filled again by the next call
            return mdhip_fail(ctx, MDHIP_EHIP, "cumtrapz: a block of the one-pass scan waited ~2 s for an earlier block's "
                                               "total (blocks not dispatched in index order?); results are not valid");
        });
    }
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
