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
// staged through LDS; a lane keeps a 16-deep sliding window of `a` in registers so that 8 LDS reads
// feed 64 FMAs, while b[t] (the same for every lane) arrives through scalar loads as an SGPR operand. The a-window is stored transposed in LDS ([i mod 8][i div 8]) so that the 64 lanes
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
              bool same, long long n_lags, double *d_out)
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
                                       n_lags, d_out + (size_t)p0 * n_lags);
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
constexpr int DX_TT = 2048;                      // time steps per LDS stage (multiple of 8)
constexpr int DX_AW = DX_TT + DX_KT + 8;         // a-window length per stage
constexpr int DX_ROW = DX_AW / 8 + 1;            // transposed rows: 8 rows of DX_ROW doubles

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
            // lane window a_stage[8*tid + tt + j], j = 0..15, as two halves that swap roles every 8 samples (no
            // moves); element i = 8*tid + j + tt sits in row (j & 7), column tid + tt/8 + (j >> 3).
            // b[t] is the same for every lane: scalar loads from global memory, SGPR operand of the FMA.
            double wa[8], wb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wa[j] = s_a[j * DX_ROW + tid];
            const double *bs = b + T0;
            const long long left = t_hi - T0;  // samples of b that belong to this slab from T0 on
            const int tt_full = (int)(left < DX_TT ? (left & ~7LL) : DX_TT);
#define DX_STEP(A, B, TT, GUARD)                                                             \
    {                                                                                        \
        const int col_ = tid + ((TT) >> 3) + 1;                                              \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) B[j] = s_a[j * DX_ROW + col_];         \
        _Pragma("unroll") for (int u = 0; u < 8; ++u)                                        \
        {                                                                                    \
            const double bt_ = (!(GUARD) || (TT) + u < left) ? bs[(TT) + u] : 0.0;           \
            _Pragma("unroll") for (int m = 0; m < DX_LPT; ++m)                               \
                acc[m] = __builtin_fma((u + m) < 8 ? A[(u + m) & 7] : B[(u + m) & 7], bt_, acc[m]); \
        }                                                                                    \
    }
            int tt = 0;
            for (; tt + 8 < tt_full; tt += 16) {
                DX_STEP(wa, wb, tt, false)
                DX_STEP(wb, wa, tt + 8, false)
            }
            if (tt < tt_full) {
                DX_STEP(wa, wb, tt, false)
#pragma unroll
                for (int j = 0; j < 8; ++j) wa[j] = wb[j];
                tt += 8;
            }
            if (tt < DX_TT && tt < left) DX_STEP(wa, wb, tt, true)  // the slab's last, partial group of 8
#undef DX_STEP
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
    double s = 0.0;
    for (int r = 0; r < n_slabs; ++r) s += partial[(size_t)r * n_lags + k];
    out[k] = s / (double)(n - (lag0 + k));
}

int xcorr_direct(mdhip_ctx *ctx, long long n, int n_pairs, const double *d_a, const double *d_b,
                 long long lag0, long long n_lags, double *d_out)
{
    const int n_tiles = (int)((n_lags + DX_KT - 1) / DX_KT);
    const int n_blocks = (n_tiles + 1) / 2;
    int n_slabs = ctx->opt_xcorr_tile > 0 ? ctx->opt_xcorr_tile : (ctx->cu_count * 4 + n_blocks - 1) / n_blocks;
    const long long max_slabs = (n - lag0 + DX_TT - 1) / DX_TT;
    if (n_slabs > max_slabs) n_slabs = (int)max_slabs;
    if (n_slabs < 1) n_slabs = 1;
    MD_WS(d_partial, double, WS_PART, (size_t)n_slabs * n_lags * 8);
    for (int p = 0; p < n_pairs; ++p) {
        // a slab can be empty for short tiles: start from zeros
        MD_HIP(hipMemsetAsync(d_partial, 0, (size_t)n_slabs * n_lags * 8, ctx->stream));
        hipLaunchKernelGGL(xcorr_direct_kernel, dim3((unsigned)n_blocks, (unsigned)n_slabs),
                           dim3(DX_THREADS), 0, ctx->stream, d_a + (size_t)p * n, d_b + (size_t)p * n, n,
                           lag0, n_lags, n_tiles, n_slabs, d_partial);
        MD_HIP(hipGetLastError());
        hipLaunchKernelGGL(xcorr_finish_kernel, dim3((unsigned)((n_lags + 255) / 256)), dim3(256), 0,
                           ctx->stream, d_partial, d_out + (size_t)p * n_lags, n, lag0, n_lags, n_slabs);
        MD_HIP(hipGetLastError());
    }
    return MDHIP_OK;
}

}  // namespace

extern "C" {

int mdhip_xcorr(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b,
                int on_device, int method, int64_t n_lags, double *out)
{
    return mdhip_xcorr_lags(ctx, n, n_pairs, a, b, on_device, method, 0, n_lags, out);
}

int mdhip_xcorr_lags(mdhip_ctx *ctx, int64_t n, int n_pairs, const double *a, const double *b, int on_device,
                     int method, int64_t lag_begin, int64_t n_lags, double *out)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(n >= 0 && n_pairs >= 0 && n_lags >= 0 && lag_begin >= 0 && lag_begin + n_lags <= n,
               "bad sizes (lag_begin + n_lags must be <= n)");
    MD_REQUIRE(method == MDHIP_XCORR_FFT || method == MDHIP_XCORR_DIRECT, "unknown method %d", method);
    MD_REQUIRE(lag_begin == 0 || method == MDHIP_XCORR_DIRECT, "a lag range needs the direct method");
    if (n == 0 || n_pairs == 0 || n_lags == 0) return MDHIP_OK;
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
    MD_WS(d_out, double, WS_OUT, out_b);
    KernelTimer timer(ctx, n_pairs);
    if (method == MDHIP_XCORR_FFT && n == 1) method = MDHIP_XCORR_DIRECT;  // one sample: the product itself
    ctx->last_kernel = method == MDHIP_XCORR_FFT ? "fft_pass_kernel<0, 0>" : "xcorr_direct_kernel";
    rc = method == MDHIP_XCORR_FFT ? xcorr_fft(ctx, n, n_pairs, d_a, d_b, same, n_lags, d_out)
                                   : xcorr_direct(ctx, n, n_pairs, d_a, d_b, lag_begin, n_lags, d_out);
    timer.stop();
    if (rc) return rc;
    MD_HIP(hipMemcpyAsync(out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
    MD_HIP(hipStreamSynchronize(ctx->stream));
    timer.collect();
    return MDHIP_OK;
}

}  // extern "C"
