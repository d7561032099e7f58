// xcorr.hip — Green-Kubo correlation functions (G2, G3) for gfx950.
//
// Replaces dynamical/conductivity.py:97-114 (correlate), dynamical/viscosity.py:103-115
// (autocorrelate "wkt" and "brute_force") of the reference:
//     c[k] = sum_{t=0}^{n-1-k} a[t+k] * b[t] / (n-k)
//
// MDHIP_XCORR_FFT: zero-pad to a power of two >= 2n, real-to-complex transforms, A*conj(B), inverse, first n lags,
// unbiased 1/(n-k). The transforms are the library's own (fft_pow2.hip: no run-time kernel compilation, so the
// first call of a process costs what every call costs), with the padding, the spectrum product and the 1/(n-k)
// scaling fused into their first / pointwise / last kernels (mdhip_fft_xcorr). HBM-bound.
//
// MDHIP_XCORR_DIRECT: register-blocked direct lag sums, FP64-FMA bound (n^2/2 fused multiply-adds
// per pair). A block owns a tile of 2048 consecutive lags (8 per lane) and streams time in chunks
// staged through LDS; a lane keeps three groups of 8 window entries in registers so that 8 LDS reads feed 64 FMAs, every
// read issued a whole group before its first use, while b[t] (the same for every lane) arrives through scalar loads,
// also a group ahead, as the SGPR operand of the FMA. The a-window is stored transposed in LDS ([i mod 8][i div 8]) so that the 64 lanes
// of a wave read consecutive doubles (no bank conflicts). Lag tiles are paired (j, nT-1-j) so every
// block has the same amount of work; time is split into slabs whose partial sums are added in a
// fixed order by a second kernel (no float atomics).
#include <algorithm>

#include "ctx.h"

namespace {

// ---------------------------------------------------------------------------------------------
// FFT path
// ---------------------------------------------------------------------------------------------

// Transform length: the reference pads to 2n (conductivity.py:111, viscosity.py:112); any length >= 2n - 1 gives the
// same linear correlation: the next power of two, the lengths fft_pow2.hip transforms. Against the 2n-point transform
// the result moves by rounding only (tests: 1e-10 acf[0]).
long long fft_length(long long n)
{
    long long L = 2;
    while (L < 2 * n) L <<= 1;
    return L;
}

int xcorr_fft(mdhip_ctx *ctx, long long n, int n_pairs, const double *d_a, const double *d_b,
              bool same, long long n_lags, double *d_out, double out_scale)
{
    const long long L = fft_length(n);
    const long long H = L / 2;
    // series pairs per chunk: all of them while the four transform buffers stay below ~1 GiB (and fit grid.y)
    int chunk = (int)std::max<long long>(1, std::min<long long>(std::min(n_pairs, 65535), (1LL << 30) / (L * 8 * 4)));
    const size_t buf_b = (size_t)chunk * H * 16 + 64;
    MD_WS(buf0, double2, WS_AUX0, buf_b);
    MD_WS(buf1, double2, WS_AUX1, buf_b);
    MD_WS(buf2, double2, WS_AUX2, same ? 64 : buf_b);
    MD_WS(buf3, double2, WS_AUX3, same ? 64 : buf_b);
    for (int p0 = 0; p0 < n_pairs; p0 += chunk) {
        const int nb = std::min(chunk, n_pairs - p0);
        const double *pa = d_a + (size_t)p0 * n;
        const int rc = mdhip_fft_xcorr(ctx, pa, same ? pa : d_b + (size_t)p0 * n, n, L, nb, buf0, buf1, buf2, buf3,
                                       n_lags, d_out + (size_t)p0 * n_lags, out_scale);
        if (rc) return rc;
    }
    return MDHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// direct path
// ---------------------------------------------------------------------------------------------

constexpr int DX_THREADS = 256;
constexpr int DX_LPT = 8;                        // consecutive lags per lane
constexpr int DX_KT = DX_THREADS * DX_LPT;       // lags per tile (2048)
constexpr int DX_TT = 2016;                      // time steps per LDS stage (42 trips of 48 steps)
constexpr int DX_AW = DX_TT + DX_KT + 8 + 16;    // a-window length per stage (+ two groups of prefetch)
constexpr int DX_ROW = DX_AW / 8 + 3;            // transposed rows: 8 rows of DX_ROW doubles (514: rows 4112 B apart,
                                                 // the 8 rows a staging store walks fall into different banks)

// partial[slab][q] = sum over the slab's time range of a[t+lag]*b[t], lag = lag0 + q, q < n_lags
// (lag0 > 0: a lag range of the whole function, the unit of the multi-GPU split by lags)
__global__ __launch_bounds__(DX_THREADS) void xcorr_direct_kernel(
    const double *__restrict__ a, const double *__restrict__ b, long long n, long long lag0, long long n_lags,
    int n_tiles, int n_slabs, double *__restrict__ partial)
{
    __shared__ double s_a[8 * DX_ROW];
    const int tid = threadIdx.x;
    const int pair_id = blockIdx.x;  // handles lag tiles pair_id and n_tiles-1-pair_id
    const int slab = blockIdx.y;
    a += (size_t)blockIdx.z * n;  // series pair of this block
    b += (size_t)blockIdx.z * n;
    partial += (size_t)blockIdx.z * n_slabs * n_lags;

    for (int half = 0; half < 2; ++half) {
        const int tile = half == 0 ? pair_id : n_tiles - 1 - pair_id;
        if (half == 1 && tile == pair_id) break;
        const long long Q0 = (long long)tile * DX_KT;  // first lag of the tile, relative to lag0
        if (Q0 >= n_lags) continue;
        const long long K0 = lag0 + Q0;
        // time range of this tile: t in [0, n-K0); split evenly into n_slabs slabs (multiples of 8)
        const long long t_total = n - K0;
        long long per = (t_total + n_slabs - 1) / n_slabs;
        per = (per + 7) & ~7LL;
        const long long t_lo = (long long)slab * per;
        const long long t_hi = t_lo + per < t_total ? t_lo + per : t_total;
        double acc[DX_LPT];
#pragma unroll
        for (int m = 0; m < DX_LPT; ++m) acc[m] = 0.0;

        for (long long T0 = t_lo; T0 < t_hi; T0 += DX_TT) {
            __syncthreads();
            // stage a[T0+K0 .. T0+K0+AW), zero beyond the series (a pair whose a sample does not exist adds 0)
            for (int i = tid; i < DX_AW; i += DX_THREADS) {
                const long long g = T0 + K0 + i;
                s_a[(i & 7) * DX_ROW + (i >> 3)] = g < n ? a[g] : 0.0;
            }
            __syncthreads();
            const long long left = t_hi - T0;  // time steps of this slab from T0 on
            const int tt_count = (int)(left < DX_TT ? left : DX_TT);
            // Lane window a_stage[8*tid + t + m], m = 0..7, held as groups of 8 consecutive entries: group g of a
            // trip needs the entries of groups g and g+1 (X, Y) and b[8g..8g+7]. Everything a group needs is requested
            // one whole group (64 FMAs, ~280 cycles) earlier: at the top of group g the 8 LDS reads of group g+2's
            // entries (Z; one address register, immediate offsets) and the scalar load of group g+1's b are issued,
            // then come the 64 FMAs on registers that are already complete — the only wait is the one at the group
            // top (lgkmcnt(0), for requests a group old). X, Y, Z rotate and the two b buffers alternate: six groups
            // (48 steps, 384 FMAs) per trip, no register moves. The fast trips stop where a prefetch of b would leave
            // the series; the steps behind them (the last stage of a slab only) are done one by one.
            const double *bs = b + T0;
            int tt = 0;
            const long long room = n - T0 - 56;  // a trip at tt reads b up to T0 + tt + 55
            int fast_end = 0;
            if (room >= 0) fast_end = (int)std::min<long long>((tt_count / 48) * 48, (room / 48 + 1) * 48);
            if (fast_end >= 48) {
                double wx[8], wy[8], wz[8], b0[8], b1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    wx[j] = s_a[j * DX_ROW + tid];
                    wy[j] = s_a[j * DX_ROW + tid + 1];
                    b0[j] = bs[j];
                }
#define DX_FMAS(X, Y, BC, U0, U1)                                                                           \
    _Pragma("unroll") for (int u = (U0); u < (U1); ++u)                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int m = 0; m < DX_LPT; ++m)                                                  \
            acc[m] = __builtin_fma((u + m) < 8 ? X[(u + m) & 7] : Y[(u + m) & 7], BC[u], acc[m]);           \
    }
// (the first step's FMAs come BEFORE the requests: they need this group's b, whose scalar load can only be waited for
// with lgkmcnt(0) — placed behind the new requests that wait would drain them too)
#define DX_GROUP(X, Y, Z, BC, BN, G)                                                                        \
    {                                                                                                       \
        DX_FMAS(X, Y, BC, 0, 1)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) BN[j] = bp[8 * ((G) + 1) + j];                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) Z[j] = row[j * DX_ROW + (G) + 2];                     \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        DX_FMAS(X, Y, BC, 1, 8)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
                for (; tt < fast_end; tt += 48) {
                    const double *bp = bs + tt;
                    const double *row = s_a + tid + (tt >> 3);
                    DX_GROUP(wx, wy, wz, b0, b1, 0)
                    DX_GROUP(wy, wz, wx, b1, b0, 1)
                    DX_GROUP(wz, wx, wy, b0, b1, 2)
                    DX_GROUP(wx, wy, wz, b1, b0, 3)
                    DX_GROUP(wy, wz, wx, b0, b1, 4)
                    DX_GROUP(wz, wx, wy, b1, b0, 5)
                }
#undef DX_GROUP
#undef DX_FMAS
            }
            for (; tt < tt_count; ++tt) {
                const double bt = bs[tt];
#pragma unroll
                for (int m = 0; m < DX_LPT; ++m)
                    acc[m] = __builtin_fma(s_a[((tt + m) & 7) * DX_ROW + tid + ((tt + m) >> 3)], bt, acc[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < DX_LPT; ++m) {
            const long long q = Q0 + (long long)tid * DX_LPT + m;
            if (q < n_lags) partial[(size_t)slab * n_lags + q] = acc[m];
        }
    }
}

__global__ void xcorr_finish_kernel(const double *__restrict__ partial, double *__restrict__ out,
                                    long long n, long long lag0, long long n_lags, int n_slabs)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_lags) return;
    partial += (size_t)blockIdx.y * n_slabs * n_lags;
    double s = 0.0;
    for (int r = 0; r < n_slabs; ++r) s += partial[(size_t)r * n_lags + k];
    out[(size_t)blockIdx.y * n_lags + k] = s / (double)(n - (lag0 + k));
}

int xcorr_direct(mdhip_ctx *ctx, long long n, int n_pairs, const double *d_a, const double *d_b,
                 long long lag0, long long n_lags, double *d_out)
{
    const int n_tiles = (int)((n_lags + DX_KT - 1) / DX_KT);
    const int n_blocks = (n_tiles + 1) / 2;
    // Time slabs: every block of a launch has the same amount of work (tiles are paired), so the launch ends with a
    // partly filled last round of blocks: ~6 rounds of the 4 resident blocks per CU keep that below a few percent
    // (C5, one series: 5 slabs = 1.2 rounds 46.5 TFLOP/s, 25 slabs = 6 rounds 54.3; tools/ab_xcorr_direct.py). All
    // series pairs of a call share one launch (grid.z) while the slab sums fit ~1 GiB.
    const long long max_slabs = (n - lag0 + DX_TT - 1) / DX_TT;
    const size_t slab_b = (size_t)n_lags * 8;
    int group = (int)std::max<long long>(1, std::min<long long>(std::min(n_pairs, 65535), (1LL << 30) / (slab_b * 8)));
    int n_slabs = ctx->opt_xcorr_tile > 0
                      ? ctx->opt_xcorr_tile
                      : (int)((6LL * ctx->cu_count * 4 + (long long)n_blocks * group - 1) / ((long long)n_blocks * group));
    if (n_slabs > max_slabs) n_slabs = (int)max_slabs;
    if (n_slabs > 65535) n_slabs = 65535;
    if (n_slabs < 1) n_slabs = 1;
    group = (int)std::max<long long>(1, std::min<long long>(group, (1LL << 30) / (slab_b * n_slabs)));
    MD_WS(d_partial, double, WS_PART, (size_t)group * n_slabs * slab_b);
    for (int p0 = 0; p0 < n_pairs; p0 += group) {
        const int np = std::min(group, n_pairs - p0);
        // a slab can be empty for short tiles: start from zeros
        MD_HIP(hipMemsetAsync(d_partial, 0, (size_t)np * n_slabs * slab_b, ctx->stream));
        hipLaunchKernelGGL(xcorr_direct_kernel, dim3((unsigned)n_blocks, (unsigned)n_slabs, (unsigned)np),
                           dim3(DX_THREADS), 0, ctx->stream, d_a + (size_t)p0 * n, d_b + (size_t)p0 * n, n,
                           lag0, n_lags, n_tiles, n_slabs, d_partial);
        MD_HIP(hipGetLastError());
        hipLaunchKernelGGL(xcorr_finish_kernel, dim3((unsigned)((n_lags + 255) / 256), (unsigned)np), dim3(256), 0,
                           ctx->stream, d_partial, d_out + (size_t)p0 * n_lags, n, lag0, n_lags, n_slabs);
        MD_HIP(hipGetLastError());
    }
    return MDHIP_OK;
}

// out[p][k] *= scale (direct estimator; the FFT path scales in its last pass)
__global__ void scale_kernel(double *__restrict__ x, long long n, double scale)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= scale;
}

// mean over the series of [n_series][m], numpy's order: ((x0 + x1) + x2 ...) / n_series
__global__ void series_mean_kernel(const double *__restrict__ x, int n_series, long long m, double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double s = x[i];
    for (int p = 1; p < n_series; ++p) s += x[(size_t)p * m + i];
    out[i] = s / (double)n_series;
}

// the correlation of device series into a device buffer, on the stream; `scale` multiplies every finished value
int xcorr_enqueue(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *d_a, const double *d_b, int method,
                  int64_t lag_begin, int64_t n_lags, double scale, double *d_out)
{
    const bool same = d_a == d_b;
    if (method == MDHIP_XCORR_FFT && n == 1) method = MDHIP_XCORR_DIRECT;  // one sample: the product itself
    ctx->last_kernel = method == MDHIP_XCORR_FFT ? "fft_pass_kernel<0, 0>" : "xcorr_direct_kernel";
    if (method == MDHIP_XCORR_FFT) return xcorr_fft(ctx, n, n_pairs, d_a, d_b, same, n_lags, d_out, scale);
    const int rc = xcorr_direct(ctx, n, n_pairs, d_a, d_b, lag_begin, n_lags, d_out);
    if (rc) return rc;
    if (scale != 1.0) {
        const long long tot = (long long)n_pairs * n_lags;
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, d_out, tot, scale);
        MD_HIP(hipGetLastError());
    }
    return MDHIP_OK;
}

int xcorr_lags_impl(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                    int method, int64_t lag_begin, int64_t n_lags, double *out, int out_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n >= 0 && n_pairs >= 0 && n_lags >= 0 && lag_begin >= 0 && lag_begin + n_lags <= n,
               "bad sizes (lag_begin + n_lags must be <= n)");
    MD_REQUIRE(method == MDHIP_XCORR_FFT || method == MDHIP_XCORR_DIRECT, "unknown method %d", method);
    MD_REQUIRE(lag_begin == 0 || method == MDHIP_XCORR_DIRECT, "a lag range needs the direct method");
    if (n == 0 || n_pairs == 0 || n_lags == 0) return cs.end();
    MD_REQUIRE(a && b && out, "NULL array");
    MD_REQUIRE(n < (1LL << 29), "series longer than 2^29 samples are not supported");
    MD_HIP(hipSetDevice(ctx->device));
    int rc;
    const size_t in_b = (size_t)n_pairs * n * 8;
    const bool same = a == b;
    const double *d_a = (const double *)mdhip_stage(ctx, WS_XYZ_I, a, in_b, on_device, &rc);
    if (rc) return rc;
    const double *d_b = d_a;
    if (!same) {
        d_b = (const double *)mdhip_stage(ctx, WS_XYZ_J, b, in_b, on_device, &rc);
        if (rc) return rc;
    }
    const size_t out_b = (size_t)n_pairs * n_lags * 8;
    double *d_out = out;
    if (!out_on_device) {
        d_out = (double *)mdhip_ws(ctx, WS_OUT, out_b);
        if (!d_out) return MDHIP_ENOMEM;
    }
    KernelTimer timer(ctx, n_pairs);
    rc = xcorr_enqueue(ctx, n, n_pairs, d_a, d_b, method, lag_begin, n_lags, 1.0, d_out);
    timer.stop();
    if (rc) return rc;
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

}  // namespace

extern "C" {

int mdhip_xcorr(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b,
                int on_device, int method, int64_t n_lags, double *out)
{
    return xcorr_lags_impl(ctx, n, n_pairs, a, b, on_device, method, 0, n_lags, out, 0);
}

int mdhip_xcorr_async(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b,
                      int on_device, int method, int64_t n_lags, double *out)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return xcorr_lags_impl(ctx, n, n_pairs, a, b, on_device, method, 0, n_lags, out, 0);
}

int mdhip_xcorr_lags(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                     int method, int64_t lag_begin, int64_t n_lags, double *out)
{
    return xcorr_lags_impl(ctx, n, n_pairs, a, b, on_device, method, lag_begin, n_lags, out, 0);
}

int mdhip_xcorr_lags_dev(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                         int method, int64_t lag_begin, int64_t n_lags, double *out_dev)
{
    return xcorr_lags_impl(ctx, n, n_pairs, a, b, on_device, method, lag_begin, n_lags, out_dev, 1);
}

int mdhip_xcorr_lags_dev_async(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                               int method, int64_t lag_begin, int64_t n_lags, double *out_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return xcorr_lags_impl(ctx, n, n_pairs, a, b, on_device, method, lag_begin, n_lags, out_dev, 1);
}

// G3 -> unit factor -> G4 (-> mean over the series) without leaving the device: see include/mdhip.h
int mdhip_green_kubo(mdhip_ctx *ctx, int64_t n, int n_series, const double *a, const double *b, int on_device,
                     int method, double acf_scale, double dx, double integral_scale, int leading_zero, double *acf,
                     double *integral, double *integral_mean)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n >= 0 && n_series >= 0, "negative sizes");
    MD_REQUIRE(method == MDHIP_XCORR_FFT || method == MDHIP_XCORR_DIRECT, "unknown method %d", method);
    if (n == 0 || n_series == 0) return cs.end();
    MD_REQUIRE(a && b && integral, "NULL array");
    MD_REQUIRE(n >= 2 && n < (1LL << 29), "series of 2 .. 2^29 samples are supported");
    MD_REQUIRE(n_series <= 65535, "at most 65535 series per call");
    MD_HIP(hipSetDevice(ctx->device));
    int rc;
    const size_t in_b = (size_t)n_series * n * 8;
    const bool same = a == b;
    const double *d_a = (const double *)mdhip_stage(ctx, WS_XYZ_I, a, in_b, on_device, &rc);
    if (rc) return rc;
    const double *d_b = d_a;
    if (!same) {
        d_b = (const double *)mdhip_stage(ctx, WS_XYZ_J, b, in_b, on_device, &rc);
        if (rc) return rc;
    }
    const int lead = leading_zero ? 1 : 0;
    const int64_t m = n - 1 + lead;
    const size_t acf_b = in_b, int_b = (size_t)n_series * m * 8;
    MD_WS(d_acf, double, WS_OUT, acf_b);
    MD_WS(d_int, double, WS_OUT2, int_b);
    KernelTimer timer(ctx, n_series);
    rc = xcorr_enqueue(ctx, n, n_series, d_a, d_b, method, 0, n, acf_scale, d_acf);
    if (rc) return rc;
    // the correlation functions start their way to the host while the integrals are still being formed
    if (acf) {
        rc = mdhip_result(cs, acf, d_acf, acf_b, 0);
        if (rc) return rc;
    }
    // (integral_scale rides in the scan's store: finished value times scale, the rounding of np.multiply on the host)
    rc = mdhip_cumtrapz_enqueue(ctx, n, n_series, d_acf, dx, lead, d_int, integral_scale);
    if (rc) return rc;
    double *d_mean = nullptr;
    if (integral_mean) {
        d_mean = (double *)mdhip_ws(ctx, WS_OUT3, (size_t)m * 8);
        if (!d_mean) return MDHIP_ENOMEM;
        hipLaunchKernelGGL(series_mean_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_int,
                           n_series, (long long)m, d_mean);
        MD_HIP(hipGetLastError());
    }
    timer.stop();
    rc = mdhip_result(cs, integral, d_int, int_b, 0);
    if (rc) return rc;
    if (integral_mean) {
        rc = mdhip_result(cs, integral_mean, d_mean, (size_t)m * 8, 0);
        if (rc) return rc;
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

int mdhip_green_kubo_async(mdhip_ctx *ctx, int64_t n, int n_series, const double *a, const double *b, int on_device,
                           int method, double acf_scale, double dx, double integral_scale, int leading_zero, double *acf,
                           double *integral, double *integral_mean)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_green_kubo(ctx, n, n_series, a, b, on_device, method, acf_scale, dx, integral_scale, leading_zero, acf,
                            integral, integral_mean);
}

}  // extern "C"
