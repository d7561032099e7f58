// Full lag x origin MSD through the autocorrelation theorem (the O(F log F) alternative to the
// lag-tile kernel of msd.hip; `lag_variant` 2 / 3).
//
//   sum_t0 |x(t0+k) - x(t0)|^2 = S1(k) - 2 S2(k)
//   S1(k) = sum_{t<F-k} x(t)^2 + sum_{t>=k} x(t)^2        (prefix sums of the per-frame squares)
//   S2(k) = sum_t0 x(t0) x(t0+k)                          (inverse transform of the power spectrum)
//
// Both sums are linear in the series, so the power spectra of all series of one (axis, entity group) are
// added BEFORE the inverse transform: one batched forward real transform (fft_pow2.hip) over every series, one column reduction of
// |X|^2, and a 3 x n_groups inverse transform. Every series is centred on its own mean first (the MSD does
// not see a constant offset), which keeps S1 as small as the data allow: the result carries an absolute
// rounding error of a few eps * log2(L) * S1(k) per lag, i.e. a relative error that grows with
// <x^2> / MSD(k). The caller gets that bound back and (variant 3) falls back to the exact-difference kernel
// when it exceeds its tolerance.
//
// Layout: r [F][3][E] as handed in (frame-major) -> padded series [batch][L] (L = power of two >= F +
// max_lag, zeros behind F) -> spectra [batch][L/2+1]. The series are processed in batches of <= ~4 GiB of
// workspace. All reductions have a fixed order (no floating-point atomics): results are reproducible.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#include "ctx.h"

namespace {

constexpr int MF_SLABS = 16;    // frame slabs of the column-mean pass
constexpr int MF_SPLITS = 32;   // row splits of the power-spectrum column reduction

// partial[slab][c] = sum over the slab's frames of r[t][c]   (c = axis * E + entity)
__global__ __launch_bounds__(256) void col_sum_kernel(const double *__restrict__ r, long long F,
                                                      long long cols, double *__restrict__ partial)
{
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long t0 = F * blockIdx.y / gridDim.y, t1 = F * (blockIdx.y + 1) / gridDim.y;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    long long t = t0;
    for (; t + 4 <= t1; t += 4) {
        s0 += r[t * cols + c];
        s1 += r[(t + 1) * cols + c];
        s2 += r[(t + 2) * cols + c];
        s3 += r[(t + 3) * cols + c];
    }
    for (; t < t1; ++t) s0 += r[t * cols + c];
    partial[(size_t)blockIdx.y * cols + c] = (s0 + s1) + (s2 + s3);
}

// The same over a SAMPLE of the frames (round 6, the long-series paths): the MSD does not see a constant offset of a
// series, so the series may be centred on any constant as long as S1 and the correlations are made of the same centred
// values — the mean only keeps S1, and with it the error bound (which is computed from the values actually used), as small
// as the data allow. The mean of one frame in `stride`, jittered inside its stride so that no periodic motion aliases,
// does that as well as the mean of all frames, and costs 1/stride of a pass over the trajectory (2.0 of the 16 ms of a
// 10 000-frame call at C4's size). Frame i of a slab: t0 + i stride + (hash(i, slab) mod stride), clipped to the slab.
__device__ __forceinline__ long long sample_frame(long long t0, long long t1, long long i, long long stride, unsigned slab)
{
    unsigned h = (unsigned)i * 2654435761u + slab * 40503u + 12345u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    const long long t = t0 + i * stride + (long long)(h % (unsigned)stride);
    return t < t1 ? t : t1 - 1;
}

inline long long sample_count(long long F, int slabs, long long stride)
{
    long long n = 0;
    for (int y = 0; y < slabs; ++y) {
        const long long t0 = F * y / slabs, t1 = F * (y + 1) / slabs;
        n += (t1 - t0 + stride - 1) / stride;
    }
    return n;
}

__global__ __launch_bounds__(256) void col_sum_sample_kernel(const double *__restrict__ r, long long F, long long cols,
                                                             long long stride, double *__restrict__ partial)
{
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long t0 = F * blockIdx.y / gridDim.y, t1 = F * (blockIdx.y + 1) / gridDim.y;
    const long long n = (t1 - t0 + stride - 1) / stride;
    double s0 = 0.0, s1 = 0.0;
    long long i = 0;
    for (; i + 2 <= n; i += 2) {
        s0 += r[sample_frame(t0, t1, i, stride, blockIdx.y) * cols + c];
        s1 += r[sample_frame(t0, t1, i + 1, stride, blockIdx.y) * cols + c];
    }
    if (i < n) s0 += r[sample_frame(t0, t1, i, stride, blockIdx.y) * cols + c];
    partial[(size_t)blockIdx.y * cols + c] = s0 + s1;
}

// mean[c] = scale * sum_slabs partial / F   (F: the number of frames summed)
__global__ void col_mean_kernel(const double *__restrict__ partial, int slabs, long long F, long long cols,
                                double scale, double *__restrict__ mean)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    double s = 0.0;
    for (int k = 0; k < slabs; ++k) s += partial[(size_t)k * cols + c];
    mean[c] = scale * s / (double)F;
}

__device__ inline double block_sum_256(double v, double *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Q[(a * G + g)][t] = sum over the group's entities of (r[t][a][e] * scale - mean[a][e])^2
__global__ __launch_bounds__(256) void frame_sq_kernel(const double *__restrict__ r,
                                                       const double *__restrict__ mean, long long E,
                                                       double scale, const long long *__restrict__ goff,
                                                       int G, long long F, double *__restrict__ Q)
{
    __shared__ double red[4];
    const long long t = blockIdx.x;
    const int a = blockIdx.y;
    const double *row = r + ((size_t)t * 3 + a) * E;
    const double *m = mean + (size_t)a * E;
    for (int g = 0; g < G; ++g) {
        double s = 0.0;
        for (long long e = goff[g] + threadIdx.x; e < goff[g + 1]; e += 256) {
            const double v = row[e] * scale - m[e];
            s += v * v;
        }
        s = block_sum_256(s, red);
        if (threadIdx.x == 0) Q[((size_t)a * G + g) * F + t] = s;
    }
}

// pad[c - c_first][t] = t < F ? r[t][c] * scale - mean[c] : 0 for the nb series from c_first, t < L
__global__ __launch_bounds__(256) void transpose_pad_kernel(const double *__restrict__ r,
                                                            const double *__restrict__ mean, long long F,
                                                            long long cols, long long c_first, long long nb,
                                                            long long L, double scale,
                                                            double *__restrict__ pad)
{
    __shared__ double tile[32][33];
    const long long c0 = c_first + (long long)blockIdx.x * 32, t0 = (long long)blockIdx.y * 32;
    const long long c_end = c_first + nb;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    if (t0 < F) {
        for (int k = ty; k < 32; k += 8) {
            const long long tt = t0 + k, cc = c0 + tx;
            tile[k][tx] = (tt < F && cc < c_end) ? r[tt * cols + cc] * scale - mean[cc] : 0.0;
        }
        __syncthreads();
    }
    for (int k = ty; k < 32; k += 8) {
        const long long cc = c0 + k, tt = t0 + tx;
        if (tt < L && cc < c_end) pad[(size_t)(cc - c_first) * L + tt] = t0 < F ? tile[tx][k] : 0.0;
    }
}

// ser[c - c_first][t] = r[t][c] * scale - mean[c], t < F, for the nb series from c_first: the same values as
// transpose_pad_kernel writes below F, through 64 x 64 tiles (runs of 64 doubles in both directions, every element
// touched once: non-temporal) instead of 32 x 32 — round 6, for the batched full-lag path whose first transform pass
// reads the series where this kernel leaves them (0.31 -> 0.17 ms per batch of 5357 series x 10 000 frames).
__global__ __launch_bounds__(256) void transpose_centre64_kernel(const double *__restrict__ r,
                                                                 const double *__restrict__ mean, long long F,
                                                                 long long cols, long long c_first, long long nb,
                                                                 double scale, double *__restrict__ ser)
{
    __shared__ double tile[64][65];
    const long long c0 = c_first + (long long)blockIdx.x * 64, t0 = (long long)blockIdx.y * 64;
    const long long c_end = c_first + nb;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const long long cc_in = c0 + tx;
    const double m = cc_in < c_end ? mean[cc_in] : 0.0;
#pragma unroll 4
    for (int k = ty; k < 64; k += 4) {
        const long long tt = t0 + k;
        tile[k][tx] = (tt < F && cc_in < c_end) ? __builtin_nontemporal_load(r + tt * cols + cc_in) * scale - m : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < 64; k += 4) {
        const long long cc = c0 + k, tt = t0 + tx;
        if (tt < F && cc < c_end) __builtin_nontemporal_store(tile[tx][k], ser + (size_t)(cc - c_first) * F + tt);
    }
}

// transpose_centre64_kernel for the columns [c_first, c_first + nb) of ONE (axis, group) segment, with the squares of the
// centred values added per frame on the way: Qpart[tile][t] = sum over the tile's (at most 64) series of ser[.][t]^2, tile =
// blockIdx.x = TSQ_TILES tiles of 64 columns (the residue-class path: frame_sq_kernel read the whole trajectory once more for the same sums).
// ser row of column c: c - c_first + row0.
constexpr int TSQ_TILES = 8;  // tiles of 64 columns per block
__global__ __launch_bounds__(256) void transpose_centre64_sq_kernel(const double *__restrict__ r,
                                                                    const double *__restrict__ mean, long long F,
                                                                    long long cols, long long c_first, long long nb, long long row0,
                                                                    double scale, double *__restrict__ ser,
                                                                    double *__restrict__ Qpart)
{
    __shared__ double tile[64][65];
    __shared__ double red[4][64];
    const long long t0 = (long long)blockIdx.y * 64, c_end = c_first + nb;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    double sq = 0.0;  // frame t0 + tx, series c0 + ty, + 4, ...: zeros beyond the segment and the trajectory
    for (int ti = 0; ti < TSQ_TILES; ++ti) {  // (several tiles of columns per block: as many times fewer rows to fold)
        const long long c0 = c_first + ((long long)blockIdx.x * TSQ_TILES + ti) * 64;
        if (c0 >= c_end) break;
        const long long cc_in = c0 + tx;
        const double m = cc_in < c_end ? mean[cc_in] : 0.0;
        if (ti) __syncthreads();
#pragma unroll 4
        for (int k = ty; k < 64; k += 4) {
            const long long tt = t0 + k;
            tile[k][tx] = (tt < F && cc_in < c_end) ? __builtin_nontemporal_load(r + tt * cols + cc_in) * scale - m : 0.0;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = ty; k < 64; k += 4) {
            const long long cc = c0 + k, tt = t0 + tx;
            const double v = tile[tx][k];
            sq = __builtin_fma(v, v, sq);
            if (tt < F && cc < c_end) __builtin_nontemporal_store(v, ser + (size_t)(cc - c_first + row0) * F + tt);
        }
    }
    red[ty][tx] = sq;
    __syncthreads();
    if (ty == 0 && t0 + tx < F) Qpart[(size_t)blockIdx.x * F + t0 + tx] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// The same for trajectories of 12 288 < F <= 24 576 frames, folded once on the way: frames t < 12 288 and t + 12 288 of the
// centred series (zero beyond F) leave as their sum and difference, ser[row][t] = a + b, ser[row][12288 + t] = a - b (rows of
// 24 576 doubles) — what the residue-class kernels of the padded length 49 152 read (msd_fft_w12r.h). The squares stay per
// frame: Qpart[.][t] takes a^2, Qpart[.][t + 12288] b^2. grid (blocks of TSQ_TILES column tiles, 192 frame tiles).
__global__ __launch_bounds__(256) void transpose_fold64_sq_kernel(const double *__restrict__ r, const double *__restrict__ mean,
                                                                  long long F, long long cols, long long c_first, long long nb,
                                                                  long long row0, double scale, double *__restrict__ ser,
                                                                  double *__restrict__ Qpart)
{
    constexpr long long HF = 12288;
    __shared__ double tile[64][65];
    __shared__ double red[2][4][64];
    const long long t0 = (long long)blockIdx.y * 64, c_end = c_first + nb;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    double sqa = 0.0, sqb = 0.0;
    for (int ti = 0; ti < TSQ_TILES; ++ti) {
        const long long c0 = c_first + ((long long)blockIdx.x * TSQ_TILES + ti) * 64;
        if (c0 >= c_end) break;
        const long long cc_in = c0 + tx;
        const double m = cc_in < c_end ? mean[cc_in] : 0.0;
        double va[16];
        for (int half = 0; half < 2; ++half) {
            __syncthreads();  // (the previous readers of the tile are done)
#pragma unroll 4
            for (int k = ty; k < 64; k += 4) {
                const long long tt = t0 + k + half * HF;
                tile[k][tx] = (tt < F && cc_in < c_end) ? __builtin_nontemporal_load(r + tt * cols + cc_in) * scale - m : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = ty + 4 * q;
                const double v = tile[tx][k];
                if (half == 0) {
                    va[q] = v;
                    sqa = __builtin_fma(v, v, sqa);
                } else {
                    sqb = __builtin_fma(v, v, sqb);
                    const long long cc = c0 + k, tt = t0 + tx;
                    if (cc < c_end) {
                        double *row = ser + (size_t)(cc - c_first + row0) * (size_t)(2 * HF);
                        __builtin_nontemporal_store(va[q] + v, row + tt);
                        __builtin_nontemporal_store(va[q] - v, row + HF + tt);
                    }
                }
            }
        }
    }
    red[0][ty][tx] = sqa;
    red[1][ty][tx] = sqb;
    __syncthreads();
    if (ty < 2) {
        const long long tt = t0 + tx + ty * HF;
        if (tt < F) Qpart[(size_t)blockIdx.x * F + tt] = (red[ty][0][tx] + red[ty][1][tx]) + (red[ty][2][tx] + red[ty][3][tx]);
    }
}

// partial[split][k] = sum over the split's rows of |spec[row][k]|^2
__global__ __launch_bounds__(256) void power_rows_kernel(const double2 *__restrict__ spec, long long K,
                                                         long long row0, long long row1,
                                                         double *__restrict__ partial)
{
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const long long n = row1 - row0;
    const long long ra = row0 + n * blockIdx.y / gridDim.y, rb = row0 + n * (blockIdx.y + 1) / gridDim.y;
    double s0 = 0.0, s1 = 0.0;
    long long q = ra;
    for (; q + 2 <= rb; q += 2) {
        const double2 z0 = spec[(size_t)q * K + k], z1 = spec[(size_t)(q + 1) * K + k];
        s0 += z0.x * z0.x + z0.y * z0.y;
        s1 += z1.x * z1.x + z1.y * z1.y;
    }
    if (q < rb) {
        const double2 z = spec[(size_t)q * K + k];
        s0 += z.x * z.x + z.y * z.y;
    }
    partial[(size_t)blockIdx.y * K + k] = s0 + s1;
}

// P[k] += sum over the splits of partial[split][k]
__global__ void power_fold_kernel(const double *__restrict__ partial, int splits, long long K,
                                  double *__restrict__ P)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int q = 0; q < splits; ++q) s += partial[(size_t)q * K + k];
    P[k] += s;
}

__global__ void real_to_complex_kernel(const double *__restrict__ P, long long n, double2 *__restrict__ Z)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Z[i] = make_double2(P[i], 0.0);
}

// ---------------------------------------------------------------------------------------------
// Fused path (L <= 16384): the whole padded series lives in LDS. One block per CU walks a contiguous range
// of series of one (axis, group) segment; per series it loads the F samples (coalesced, from the
// time-major copy), removes the mean, adds the squares into per-lane accumulators (S1), runs an in-place
// radix-8 decimation-in-frequency transform of the L/2-point complex packing x[2n] + i x[2n+1], untangles
// the real spectrum and adds |X_k|^2 into per-lane accumulators. Nothing but the two partial sums per block
// goes back to HBM: the spectra never exist in memory.
//
// LDS layout: re[] and im[] as separate double arrays, logical index i stored at i + (i >> 4) so that the
// stride-2 / stride-16 passes stay at <= 2-way bank conflicts. Twiddles come from a two-level table
// w^i = A[i >> 7] * B[i & 127] (4 KiB) and w^2..w^7 by multiplication.
// ---------------------------------------------------------------------------------------------

constexpr int FT_THREADS = 512;                 // 2 waves per SIMD, 256 VGPRs each: the accumulators stay in registers
constexpr int FT_MAX_M = 13;                  // N = L/2 <= 8192 complex points
constexpr int FT_PREGS = 17;                  // k = 0..N: <= 17 per lane

struct FftItem {
    long long c_lo, c_hi;  // series c_lo, c_lo + step, ... < c_hi of one segment
    int step;              // 1: a contiguous range; 16: one column of every 16-column tile (the direct-read kernel)
    int row;               // row of Qpart / Ppart the block writes (rows of one segment are consecutive)
};

// SRC == 2 of msd_power_lds3_kernel (round 4: a cluster of 16 blocks transposes its tiles itself): what a member needs
// beside its FftItem, whose c_lo / c_hi are the cluster's range of 16-column TILES there
struct FftStage {
    long long lo, hi;  // the segment's columns [lo, hi): a member whose column 16 T + k lies outside transforms zeros
    int k, cluster;    // member 0 .. 15: column 16 T + k of every tile; rows [k Fc, (k + 1) Fc) are its staging share
};
constexpr int ST_AHEAD = 4;  // tile i + ST_AHEAD is staged while tile i is transformed
constexpr int ST_BUF = 8;    // ring of staged tiles per cluster (3 ST_AHEAD - 4 can be live at once)
constexpr int ST_UNITS = 8;  // 16-byte units a lane moves per tile at most (template parameter UN: 5 for rows per member
                             // Fc <= 320, i.e. F <= 5120; 8 beyond)
constexpr int ST_FLAG_STRIDE = 32;  // words between two ready counters (a 128-byte line each)
#ifndef ST_SKIP
#define ST_SKIP 0  // timing experiments only (wrong results): 1 = no polls, 2 = no staging loads / stores inside the series loop,
                   // 4 = no wait + signal, 16 = no ring stores, 32 = no staging loads
#endif

__device__ __forceinline__ int ft_skew(int i) { return i + (i >> 4); }

// From here on the arithmetic may fuse multiply-adds (the library is built with -ffp-contract=off for the pair
// kernels' bit-exactness; this path is tolerance parity and returns a computed rounding bound).
#pragma clang fp contract(fast)
struct Cx {
    double x, y;
};
__device__ __forceinline__ Cx cx_mul(Cx a, Cx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ Cx cx_add(Cx a, Cx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ Cx cx_sub(Cx a, Cx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ Cx cx_mul_mi(Cx a) { return {a.y, -a.x}; }  // a * (-i)

// w_L^i = exp(-2 pi i / L), 0 <= i < L
__device__ __forceinline__ Cx ft_tw(const double2 *tabA, const double2 *tabB, int i)
{
    const double2 a = tabA[i >> 7], b = tabB[i & 127];
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// position of frequency k after the in-place passes (radix 8 ... 8, then 2^rem)
__device__ __forceinline__ int ft_pos(int k, int m)
{
    int p = 0, bits = m;
    while (bits >= 3) {
        bits -= 3;
        p += (k & 7) << bits;
        k >>= 3;
    }
    return p + k;  // the last digit (rem bits) lands at stride 1
}

__device__ __forceinline__ void dft4(Cx c0, Cx c1, Cx c2, Cx c3, Cx &z0, Cx &z1, Cx &z2, Cx &z3)
{
    const Cx d0 = cx_add(c0, c2), d1 = cx_sub(c0, c2), d2 = cx_add(c1, c3), d3 = cx_mul_mi(cx_sub(c1, c3));
    z0 = cx_add(d0, d2);
    z2 = cx_sub(d0, d2);
    z1 = cx_add(d1, d3);
    z3 = cx_sub(d1, d3);
}

// In-place DIF transform of the N = 2^m points in re/im (skewed indexing); every thread of the block calls it.
// tw_shift: twiddle table index of w_N^1 relative to the table's base root (1 for a table of w_{2N}).
__device__ void ft_transform(double *re, double *im, int m, const double2 *tabA, const double2 *tabB)
{
    const int N = 1 << m;
    int lb = m;  // log2 of the current sub-transform length
    while (lb >= 3) {
        const int ls = lb - 3, s = 1 << ls;  // butterfly stride
        for (int b = threadIdx.x; b < (N >> 3); b += FT_THREADS) {
            const int j = b & (s - 1), base = (b >> ls) << lb;
            int idx[8];
            Cx a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                idx[q] = ft_skew(base + j + (q << ls));
                a[q] = {re[idx[q]], im[idx[q]]};
            }
            const Cx b0 = cx_add(a[0], a[4]), b4 = cx_sub(a[0], a[4]);
            const Cx b1 = cx_add(a[1], a[5]), t5 = cx_sub(a[1], a[5]);
            const Cx b2 = cx_add(a[2], a[6]), t6 = cx_sub(a[2], a[6]);
            const Cx b3 = cx_add(a[3], a[7]), t7 = cx_sub(a[3], a[7]);
            constexpr double H = 0.70710678118654752440;
            const Cx b5 = {(t5.x + t5.y) * H, (t5.y - t5.x) * H};   // t5 * (1 - i)/sqrt2
            const Cx b6 = cx_mul_mi(t6);                            // t6 * (-i)
            const Cx b7 = {(t7.y - t7.x) * H, -(t7.x + t7.y) * H};  // t7 * (-1 - i)/sqrt2
            Cx y[8];
            dft4(b0, b1, b2, b3, y[0], y[2], y[4], y[6]);
            dft4(b4, b5, b6, b7, y[1], y[3], y[5], y[7]);
            if (ls > 0) {
                // w_len^(j p) = w_L^(j p L / len), L = 2N: table index of p = 1 is j << (m + 1 - lb)
                const Cx w1 = ft_tw(tabA, tabB, j << (m + 1 - lb));
                const Cx w2 = cx_mul(w1, w1), w3 = cx_mul(w2, w1), w4 = cx_mul(w2, w2);
                const Cx w5 = cx_mul(w4, w1), w6 = cx_mul(w4, w2), w7 = cx_mul(w4, w3);
                y[1] = cx_mul(y[1], w1);
                y[2] = cx_mul(y[2], w2);
                y[3] = cx_mul(y[3], w3);
                y[4] = cx_mul(y[4], w4);
                y[5] = cx_mul(y[5], w5);
                y[6] = cx_mul(y[6], w6);
                y[7] = cx_mul(y[7], w7);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                re[idx[q]] = y[q].x;
                im[idx[q]] = y[q].y;
            }
        }
        __syncthreads();
        lb -= 3;
    }
    if (lb == 2) {  // contiguous groups of four
        for (int b = threadIdx.x; b < (N >> 2); b += FT_THREADS) {
            int idx[4];
            Cx a[4], y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                idx[q] = ft_skew(4 * b + q);
                a[q] = {re[idx[q]], im[idx[q]]};
            }
            dft4(a[0], a[1], a[2], a[3], y[0], y[1], y[2], y[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                re[idx[q]] = y[q].x;
                im[idx[q]] = y[q].y;
            }
        }
        __syncthreads();
    } else if (lb == 1) {
        for (int b = threadIdx.x; b < (N >> 1); b += FT_THREADS) {
            const int i0 = ft_skew(2 * b), i1 = ft_skew(2 * b + 1);
            const Cx a0 = {re[i0], im[i0]}, a1 = {re[i1], im[i1]};
            re[i0] = a0.x + a1.x;
            im[i0] = a0.y + a1.y;
            re[i1] = a0.x - a1.x;
            im[i1] = a0.y - a1.y;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double ft_block_sum(double v, double *red)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < FT_THREADS / 64; ++w) s += red[w];
    return s;
}

// LDS bytes of the two fused kernels for N = 2^m
size_t ft_lds_bytes(int m)
{
    const size_t np = ((size_t)1 << m) + ((size_t)1 << m >> 4) + 2;
    return 2 * np * 8 + 256 * 16 + 32 * 8;
}

// x: time-major series [cols][F] (already scaled). Qpart [items][F], Ppart [items][N + 1].
template <int QR>
__global__ __launch_bounds__(FT_THREADS) void msd_power_lds_kernel(
    const double *__restrict__ x, int F, int m, const FftItem *__restrict__ items,
    const double2 *__restrict__ tab, double *__restrict__ Qpart, double *__restrict__ Ppart)
{
    extern __shared__ double ft_lds[];
    const int N = 1 << m, np = N + (N >> 4) + 2;
    double *re = ft_lds, *im = ft_lds + np;
    double2 *tabA = reinterpret_cast<double2 *>(im + np), *tabB = tabA + 128;
    double *red = reinterpret_cast<double *>(tabB + 128);
    const int tid = threadIdx.x;
    if (tid < 256) tabA[tid] = tab[tid];
    const FftItem it = items[blockIdx.x];
    double qacc[QR], pacc[FT_PREGS];
#pragma unroll
    for (int i = 0; i < QR; ++i) qacc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < FT_PREGS; ++i) pacc[i] = 0.0;
    const int half = (F + 1) >> 1;  // packed points that hold data
    for (long long c = it.c_lo; c < it.c_hi; ++c) {
        const double *row = x + (size_t)c * F;
        double v[QR];
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            const int t = tid + i * FT_THREADS;
            v[i] = t < F ? row[t] : 0.0;
            sum += v[i];
        }
        const double mean = ft_block_sum(sum, red) / (double)F;  // (also orders the previous untangle reads)
#pragma unroll
        for (int i = 0; i < QR; ++i) {
            const int t = tid + i * FT_THREADS;
            if (t < F) {
                const double d = v[i] - mean;
                qacc[i] += d * d;
                const int n = ft_skew(t >> 1);
                if (t & 1)
                    im[n] = d;
                else {
                    re[n] = d;
                    if (t == F - 1) im[n] = 0.0;
                }
            }
        }
        for (int n = half + tid; n < N; n += FT_THREADS) {
            const int p = ft_skew(n);
            re[p] = 0.0;
            im[p] = 0.0;
        }
        __syncthreads();
        ft_transform(re, im, m, tabA, tabB);
#pragma unroll
        for (int i = 0; i < FT_PREGS; ++i) {
            const int k = tid + i * FT_THREADS;
            if (k <= N) {
                const int ia = ft_skew(ft_pos(k & (N - 1), m)), ib = ft_skew(ft_pos((N - k) & (N - 1), m));
                const Cx zk = {re[ia], im[ia]}, zn = {re[ib], im[ib]};
                const Cx e = {0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y)};
                const Cx o = {0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x)};
                const Cx w = ft_tw(tabA, tabB, k);
                const Cx xk = cx_add(e, cx_mul(w, o));
                pacc[i] += xk.x * xk.x + xk.y * xk.y;
            }
        }
        // the next series' block sum has two barriers before LDS is written again
    }
    double *q = Qpart + (size_t)blockIdx.x * F, *p = Ppart + (size_t)blockIdx.x * (N + 1);
#pragma unroll
    for (int i = 0; i < QR; ++i) {
        const int t = tid + i * FT_THREADS;
        if (t < F) q[t] = qacc[i];
    }
#pragma unroll
    for (int i = 0; i < FT_PREGS; ++i) {
        const int k = tid + i * FT_THREADS;
        if (k <= N) p[k] = pacc[i];
    }
}


// ---------------------------------------------------------------------------------------------
// Round 3: the fused kernel re-organised around what its counters said (profiles/r03_pmc_secondary_summary.txt: the
// LDS array busy half of the time, 47 % of those cycles bank conflicts; 2 waves per SIMD waiting 64 % of their cycles).
//
//  * Layout: logical point i at i + (i >> 5) (one pad double per 32). Every access of every phase then touches 32
//    different banks per 32-lane group: lanes that walk CONSECUTIVE points (the load, the passes with butterfly
//    stride >= 32, the accumulation's own point) stay inside one aligned run of 32; the in-register radix-16 tail,
//    where lane b owns the 16 points from 16 b, reads 16 b + e + (b >> 1) — for a fixed e the 32 lanes of a group cover
//    e .. e + 31; the one pass with butterfly stride 16 deals its 128-point blocks to the two halves of a lane group
//    four blocks apart (a bit swap of the block index: pad shift 16). The old i + (i >> 4) skew made the tail and the
//    stride-16 pass conflict-free but cost every run of consecutive points one 2-way conflict, and the real-spectrum
//    pass, which walked FREQUENCIES (digit-reversed positions), ran 8-way.
//  * Passes: 2^m = [2 | 4] x 8 x ... x 8 x 16 — radix-8 passes through LDS down to blocks of 16 consecutive points,
//    which one lane then transforms in registers: 4 LDS round trips at m = 13 instead of 5.
//  * The spectrum is accumulated BILINEARLY, per position and not per frequency. With Z the packed transform, X the
//    real spectrum, w = e^{-2 pi i k/L}:   |X_k|^2 = (|Z_k|^2 + |Z_{N-k}|^2)/2 + Im(w) (|Z_k|^2 - |Z_{N-k}|^2)/2 + Re(w) Im(Z_k Z_{N-k}),
//    so a lane only adds |Z_p|^2 and Im(Z_p Z_p') for its own positions p (p' holds the frequency N - k: a fixed
//    position per lane, read with few conflicts because the digit reversal maps a run of positions onto a run of
//    partners) — 4 fused multiply-adds per point instead of the ~30 operations + table lookups of the untangling —
//    and the frequencies are sorted out ONCE per block, after its last series.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int f2_skew(int i) { return i + (i >> 5); }

struct F2Plan {
    int n_pass;      // LDS passes before the in-register tail
    int lr[6];       // log2 radix of pass q (1, 2 or 3)
    int ls[6];       // log2 butterfly stride of pass q
};

__host__ __device__ inline F2Plan f2_plan(int m)
{
    F2Plan p{};
    int lb = m, q = 0;
    const int r0 = (m - 4) % 3;
    if (r0) {
        p.lr[q] = r0;
        p.ls[q] = lb - r0;
        lb -= r0;
        ++q;
    }
    while (lb > 4) {
        p.lr[q] = 3;
        p.ls[q] = lb - 3;
        lb -= 3;
        ++q;
    }
    p.n_pass = q;
    return p;
}

// position of frequency k after the in-place passes + tail (and back): every pass peels the lowest remaining
// frequency digit and leaves it at its butterfly stride; the tail's 16-point DFT (two radix-4 stages) leaves
// frequency f0 + 4 f1 at local index 4 f0 + f1
__device__ __forceinline__ int f2_pos(int k, const F2Plan &pl)
{
    int p = 0;
    for (int q = 0; q < pl.n_pass; ++q) {
        p += (k & ((1 << pl.lr[q]) - 1)) << pl.ls[q];
        k >>= pl.lr[q];
    }
    return p + 4 * (k & 3) + (k >> 2);
}

__device__ __forceinline__ int f2_freq(int p, const F2Plan &pl)
{
    int k = 0, w = 0;
    for (int q = 0; q < pl.n_pass; ++q) {
        k += ((p >> pl.ls[q]) & ((1 << pl.lr[q]) - 1)) << w;
        w += pl.lr[q];
    }
    const int t = p & 15;
    return k + (((t >> 2) + 4 * (t & 3)) << w);
}

// position (in a tail block) of the tail digit (16 - d) mod 16, d the digit position t holds
__device__ __forceinline__ constexpr int f2_tail_neg(int t)
{
    const int d = (t >> 2) + 4 * (t & 3), e = (16 - d) & 15;
    return 4 * (e & 3) + (e >> 2);
}

// One radix-8 DIF butterfly on a[0..7] (elements j + q s of a sub-transform of length 8 s), outputs in a[] with
// output digit d at a[d], twiddled by w1^d (w1 = w_len^j); first = no twiddle (s == 1 never happens here: the tail
// takes the last 16 points)
__device__ __forceinline__ void f2_bfly8(Cx *a, Cx w1, bool twiddle)
{
    const Cx b0 = cx_add(a[0], a[4]), b4 = cx_sub(a[0], a[4]);
    const Cx b1 = cx_add(a[1], a[5]), t5 = cx_sub(a[1], a[5]);
    const Cx b2 = cx_add(a[2], a[6]), t6 = cx_sub(a[2], a[6]);
    const Cx b3 = cx_add(a[3], a[7]), t7 = cx_sub(a[3], a[7]);
    constexpr double H = 0.70710678118654752440;
    const Cx b5 = {(t5.x + t5.y) * H, (t5.y - t5.x) * H};
    const Cx b6 = cx_mul_mi(t6);
    const Cx b7 = {(t7.y - t7.x) * H, -(t7.x + t7.y) * H};
    Cx y[8];
    dft4(b0, b1, b2, b3, y[0], y[2], y[4], y[6]);
    dft4(b4, b5, b6, b7, y[1], y[3], y[5], y[7]);
    if (twiddle) {
        const Cx w2 = cx_mul(w1, w1), w3 = cx_mul(w2, w1), w4 = cx_mul(w2, w2);
        const Cx w5 = cx_mul(w4, w1), w6 = cx_mul(w4, w2), w7 = cx_mul(w4, w3);
        y[1] = cx_mul(y[1], w1);
        y[2] = cx_mul(y[2], w2);
        y[3] = cx_mul(y[3], w3);
        y[4] = cx_mul(y[4], w4);
        y[5] = cx_mul(y[5], w5);
        y[6] = cx_mul(y[6], w6);
        y[7] = cx_mul(y[7], w7);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] = y[q];
}

// 16-point DFT in registers, DIF as two radix-4 stages; output frequency f0 + 4 f1 ends up in x[4 f0 + f1]
__device__ __forceinline__ void f2_dft16(Cx *x)
{
    constexpr double C1 = 0.92387953251128675613, S1 = 0.38268343236508977173, H = 0.70710678118654752440;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        Cx y0, y1, y2, y3;
        dft4(x[e], x[e + 4], x[e + 8], x[e + 12], y0, y1, y2, y3);
        // twiddle w16^(e d), d = output digit
        if (e == 1) {
            y1 = cx_mul(y1, Cx{C1, -S1});
            y2 = Cx{(y2.x + y2.y) * H, (y2.y - y2.x) * H};
            y3 = cx_mul(y3, Cx{S1, -C1});
        } else if (e == 2) {
            y1 = Cx{(y1.x + y1.y) * H, (y1.y - y1.x) * H};
            y2 = cx_mul_mi(y2);
            y3 = Cx{(y3.y - y3.x) * H, -(y3.x + y3.y) * H};
        } else if (e == 3) {
            y1 = cx_mul(y1, Cx{S1, -C1});
            y2 = Cx{(y2.y - y2.x) * H, -(y2.x + y2.y) * H};
            y3 = cx_mul(y3, Cx{-C1, S1});  // w16^9 = -w16^1
        }
        x[e] = y0;
        x[e + 4] = y1;
        x[e + 8] = y2;
        x[e + 12] = y3;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        Cx y0, y1, y2, y3;
        dft4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3], y0, y1, y2, y3);
        x[4 * g] = y0;
        x[4 * g + 1] = y1;
        x[4 * g + 2] = y2;
        x[4 * g + 3] = y3;
    }
}

// Every pass's first twiddles w_len^j come from a per-pass LDS table walked by consecutive lanes (`twp`; the two-level
// table is walked with strides of 2^k entries there: 2- to 8-way bank conflicts). Variant for A/B builds
// (tools/build_variant.sh): F2V_PREFETCH = the next series' samples fetched into registers under the transform
// (measured 6 % SLOWER at C4: 20 more live registers at a budget the accumulators already fill).
// The LDS passes, in place on re/im (f2_skew layout); every thread of the block calls it. The in-register radix-16
// tail is the caller's (it accumulates the spectrum sums from the registers the tail leaves). `half` = packed points
// that hold data: the rest of the input is the zero padding, which the FIRST pass neither needs in LDS (nobody
// writes it) nor reads — a butterfly input at or beyond `half` is a zero in a register; with F = 5000 in L = 16384
// five of the eight inputs of every first-pass butterfly are.
template <int NT>
__device__ __forceinline__ void f2_passes(double *re, double *im, int m, int half, const F2Plan &pl, const double2 *twp)
{
    const int N = 1 << m;
    int off = 0;
    for (int q = 0; q < pl.n_pass; ++q) {
        const int lr = pl.lr[q], ls = pl.ls[q], lb = ls + lr, s = 1 << ls;
        const int nbf = N >> lr;  // butterflies of this pass
        const int lim = q == 0 ? half : N;  // inputs at or beyond it are zeros (first pass: one block, base = j)
        // the stride-16 pass: the two 16-lane halves of a lane group take blocks 4 apart (pad shift 16), when the
        // transform has at least 8 such blocks
        const bool swz = ls == 4 && (N >> lb) >= 8;
        for (int b = threadIdx.x; b < nbf; b += NT) {
            const int j = b & (s - 1);
            int blk = b >> ls;
            if (swz) blk = (blk & ~5) | ((blk & 1) << 2) | ((blk >> 2) & 1);
            const int base = (blk << lb) + j;
            const double2 wv = twp[off + j];  // w_len^j = w_L^(j L / len), L = 2 N
            const Cx w1 = {wv.x, wv.y};
            if (lr == 3) {
                int idx[8];
                Cx a[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    idx[e] = f2_skew(base + (e << ls));
                    if ((e << ls) < lim) {  // (wave-uniform: whole rows of the zero padding are not read at all)
                        a[e] = base + (e << ls) < lim ? Cx{re[idx[e]], im[idx[e]]} : Cx{0.0, 0.0};
                    } else {
                        a[e] = Cx{0.0, 0.0};
                    }
                }
                f2_bfly8(a, w1, true);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    re[idx[e]] = a[e].x;
                    im[idx[e]] = a[e].y;
                }
            } else if (lr == 2) {
                int idx[4];
                Cx a[4], y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    idx[e] = f2_skew(base + (e << ls));
                    a[e] = base + (e << ls) < lim ? Cx{re[idx[e]], im[idx[e]]} : Cx{0.0, 0.0};
                }
                dft4(a[0], a[1], a[2], a[3], y[0], y[1], y[2], y[3]);
                const Cx w2 = cx_mul(w1, w1), w3 = cx_mul(w2, w1);
                y[1] = cx_mul(y[1], w1);
                y[2] = cx_mul(y[2], w2);
                y[3] = cx_mul(y[3], w3);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    re[idx[e]] = y[e].x;
                    im[idx[e]] = y[e].y;
                }
            } else {
                const int i0 = f2_skew(base), i1 = f2_skew(base + s);
                const Cx a0 = base < lim ? Cx{re[i0], im[i0]} : Cx{0.0, 0.0};
                const Cx a1 = base + s < lim ? Cx{re[i1], im[i1]} : Cx{0.0, 0.0};
                const Cx d = cx_mul(cx_sub(a0, a1), w1);
                re[i0] = a0.x + a1.x;
                im[i0] = a0.y + a1.y;
                re[i1] = d.x;
                im[i1] = d.y;
            }
        }
        off += s;
        __syncthreads();
    }
}

// LDS bytes of the round-3 kernel for N = 2^m: data planes + the two-level twiddle table (+ the per-pass tables)
size_t f2_lds_bytes(int m)
{
    const size_t np = ((size_t)1 << m) + ((size_t)1 << m >> 5) + 2;
    size_t tw = 0;
    const F2Plan pl = f2_plan(m);
    for (int q = 0; q < pl.n_pass; ++q) tw += (size_t)1 << pl.ls[q];
    return 2 * np * 8 + 256 * 16 + 32 * 8 + tw * 16;
}

// x: time-major series [cols][F] (already scaled). Qpart [items][F], Ppart [items][N + 1] as msd_power_lds_kernel.
// QR2 = sample PAIRS per lane that can hold data. Lane b < N / 16 owns the 16 consecutive positions of tail block b
// (N <= 8192 = 16 x FT_THREADS): the in-register radix-16 leaves their transform values in its registers, and the
// spectrum sums are taken from there — |Z|^2 without touching LDS again, Im(Z Z') against the partner frequencies, which
// for the 16 positions of one block all lie in ONE other block (N - k flips the low digits together and reverses the
// tail digit), read as 16 consecutive points.
template <int QR2>
__global__ __launch_bounds__(FT_THREADS) void msd_power_lds2_kernel(
    const double *__restrict__ x, int F, int m, const FftItem *__restrict__ items,
    const double2 *__restrict__ tab, double *__restrict__ Qpart, double *__restrict__ Ppart)
{
    constexpr int PR = 16;
    extern __shared__ double ft_lds[];
    const int N = 1 << m, np = N + (N >> 5) + 2;
    double *re = ft_lds, *im = ft_lds + np;
    double2 *tabA = reinterpret_cast<double2 *>(im + np), *tabB = tabA + 128;
    double *red = reinterpret_cast<double *>(tabB + 128);
    double2 *twp = reinterpret_cast<double2 *>(red + 32);
    const int tid = threadIdx.x;
    if (tid < 256) tabA[tid] = tab[tid];
    const F2Plan pl = f2_plan(m);
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    {
        int off = 0;
        for (int q = 0; q < pl.n_pass; ++q) {
            const int ls = pl.ls[q], lb = ls + pl.lr[q], s = 1 << ls;
            for (int j = tid; j < s; j += FT_THREADS) {
                const Cx w = ft_tw(tabA, tabB, j << (m + 1 - lb));
                twp[off + j] = make_double2(w.x, w.y);
            }
            off += s;
        }
    }
    // this lane's positions p = 16 tid + t and the position of each one's partner frequency N - k
    const bool owner = tid < (N >> 4);
    const int p0 = f2_skew(16 * tid);  // 16 consecutive points never cross a pad (pads sit at multiples of 32)
    // position 16 b + t holds frequency K_b + t N/16 (K_b < N/16: the digits of the LDS passes); its partner N - k is
    // K' + (15 - t) N/16 with K' = N/16 - K_b, all in the block that holds K' — except K_b = 0 (lane 0), whose partners
    // stay in its own block, at the position of tail digit (16 - d) mod 16 (a position t holds tail digit
    // d = (t >> 2) + 4 (t & 3), so 15 - d sits at 15 - t)
    const int kb = owner ? f2_freq(16 * tid, pl) : 0;
    const int pb = owner ? f2_skew(f2_pos(((N >> 4) - kb) & ((N >> 4) - 1), pl)) : 0;
    const bool kb0 = kb == 0;
#define F2_PARTNER(t) (pb + (kb0 ? f2_tail_neg(t) : 15 - (t)))
    double qa[QR2], qb[QR2], sacc[PR], tacc[9];
#pragma unroll
    for (int i = 0; i < QR2; ++i) qa[i] = qb[i] = 0.0;
#pragma unroll
    for (int t = 0; t < PR; ++t) sacc[t] = 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) tacc[u] = 0.0;
    const int half = (F + 1) >> 1;  // packed points that hold data
    double va[QR2], vb[QR2];
    auto fetch = [&](long long c) {
        const double *row = x + (size_t)c * F;
        const bool al16 = (reinterpret_cast<unsigned long long>(row) & 15ull) == 0ull;
#pragma unroll
        for (int i = 0; i < QR2; ++i) {
            const int n = tid + i * FT_THREADS;  // packed point: samples 2 n, 2 n + 1
            va[i] = vb[i] = 0.0;
            if (2 * n + 1 < F) {
                if (al16) {
                    typedef double d2_t __attribute__((ext_vector_type(2)));
                    const d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(row + 2 * n));
                    va[i] = v[0];
                    vb[i] = v[1];
                } else {
                    va[i] = row[2 * n];
                    vb[i] = row[2 * n + 1];
                }
            } else if (2 * n < F) {
                va[i] = row[2 * n];
            }
        }
    };
    for (long long c = it.c_lo; c < it.c_hi; ++c) {
        fetch(c);
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < QR2; ++i) sum += va[i] + vb[i];
        const double mean = ft_block_sum(sum, red) / (double)F;  // (its barriers also order the previous series' reads)
#pragma unroll
        for (int i = 0; i < QR2; ++i) {
            const int n = tid + i * FT_THREADS;
            if (n < half) {
                const double da = va[i] - mean;
                const double db = 2 * n + 1 < F ? vb[i] - mean : 0.0;
                qa[i] += da * da;
                qb[i] += db * db;
                const int q = f2_skew(n);
                re[q] = da;
                im[q] = db;
            }
        }
        if (pl.n_pass == 0) {  // (no LDS pass prunes the padding: N = 16, the tail reads all of it)
            for (int n = half + tid; n < N; n += FT_THREADS) {
                const int q = f2_skew(n);
                re[q] = 0.0;
                im[q] = 0.0;
            }
        }
        __syncthreads();
        f2_passes<FT_THREADS>(re, im, m, half, pl, twp);
        Cx z[PR];
        if (owner) {
#pragma unroll
            for (int e = 0; e < PR; ++e) z[e] = {re[p0 + e], im[p0 + e]};
            f2_dft16(z);
#pragma unroll
            for (int e = 0; e < PR; ++e) {
                re[p0 + e] = z[e].x;
                im[p0 + e] = z[e].y;
            }
        }
        if (owner) {
#pragma unroll
            for (int t = 0; t < PR; ++t) {
                sacc[t] = __builtin_fma(z[t].x, z[t].x, sacc[t]);
                sacc[t] = __builtin_fma(z[t].y, z[t].y, sacc[t]);
            }
        }
        __syncthreads();
        if (owner) {
            // T is the same number at a position and at its partner: a lane takes it for 8 of its 16 positions, the
            // partner block's lane for the other 8 (lane 0, its own partner block, for one position of each of its 9
            // pairs: slot 8)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t0 = u < 3 ? u : u + 1;  // lane 0: 0 1 2 4 5 6 7 8
                const double zx = kb0 ? z[t0].x : z[u].x, zy = kb0 ? z[t0].y : z[u].y;
                const int pt = pb + (kb0 ? f2_tail_neg(t0) : 15 - u);
                const double cx = re[pt], cy = im[pt];
                tacc[u] = __builtin_fma(zx, cy, tacc[u]);
                tacc[u] = __builtin_fma(zy, cx, tacc[u]);
            }
            if (kb0) {
                const int pt = pb + f2_tail_neg(9);
                tacc[8] = __builtin_fma(z[9].x, im[pt], tacc[8]);
                tacc[8] = __builtin_fma(z[9].y, re[pt], tacc[8]);
            }
        }
        // the next series' block sum has two barriers before LDS is written again
    }
    double *q = Qpart + (size_t)blockIdx.x * F, *pp = Ppart + (size_t)blockIdx.x * (N + 1);
#pragma unroll
    for (int i = 0; i < QR2; ++i) {
        const int n = tid + i * FT_THREADS;
        if (2 * n < F) q[2 * n] = qa[i];
        if (2 * n + 1 < F) q[2 * n + 1] = qb[i];
    }
    // frequencies, once per block: S of the partner frequency through LDS, then
    //   P_k = (S_k + S_{N-k})/2 + Im(w) (S_k - S_{N-k})/2 + Re(w) T_k,   w = e^{-2 pi i k/L};   P_N = S_0 - T_0
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int t = 0; t < PR; ++t) re[p0 + t] = sacc[t];
#pragma unroll
        for (int u = 0; u < 8; ++u) im[p0 + (kb0 ? (u < 3 ? u : u + 1) : u)] = tacc[u];
        if (kb0) im[p0 + 9] = tacc[8];
    }
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int t = 0; t < PR; ++t) {
            const int k = f2_freq(16 * tid + t, pl);
            const bool mine = kb0 ? (t != 3 && t < 10) : t < 8;  // T taken at this position, else at its partner
            const double sk = sacc[t], sn = re[F2_PARTNER(t)], tk = im[mine ? p0 + t : F2_PARTNER(t)];
            const Cx w = ft_tw(tabA, tabB, k);  // (cos, -sin) of 2 pi k / L
            pp[k] = 0.5 * (sk + sn) + w.y * (0.5 * (sk - sn)) + w.x * tk;
            if (k == 0) pp[N] = sk - tk;
        }
    }
#undef F2_PARTNER
}

// ---------------------------------------------------------------------------------------------
// Round 3, second step: the same transform with the block barriers taken out of its middle. The counters of the kernel
// above (profiles/r03_pmc_secondary_summary.txt) showed the LDS array busy 41 % and the f64 pipes ~30 % of the time,
// one after the other: with one 512-thread block per CU (the data planes fill LDS) every pass was "all waves read, all
// waves compute, all waves write, barrier", so neither unit ever worked under the other.
//
//  * First pass from REGISTERS. It is always radix 8 at stride N/8, and lane `tid` loads exactly the samples its own
//    butterflies j = tid (+ 512) need: packed points j + e N/8. The series never goes to LDS untransformed — no input
//    store, no first-pass reads, and the zero padding (e >= QE) is a literal zero that prunes half the butterfly.
//  * After it the transform is 8 independent sub-transforms of N/8 points: wave w takes sub-transform w through
//    all remaining LDS passes and the in-register radix-16 tail with NO block barrier (LDS serves one wave's
//    accesses in order), so the eight waves drift apart and one wave's LDS phase runs under another's arithmetic.
//    Three block barriers per series are left: first-pass writes -> sub-transforms, tail write-back -> partner reads,
//    partner reads -> the next series' first-pass writes.
//  * The next series' samples are fetched right after the first pass has consumed the current ones (the registers
//    are free again), land during the sub-transforms, and their block sum rides on the second barrier.
// Plan: 2^m = 8 x [4 x [4]] x 8 ... x 16 (f3_plan); positions/frequencies through f2_pos / f2_freq as above.
// ---------------------------------------------------------------------------------------------
#ifndef F3_MIN_M
#define F3_MIN_M 12  // smallest log2 N the kernel below is used for (it is correct from 9; A/B builds lower this)
#endif
__host__ __device__ inline F2Plan f3_plan(int m)  // m >= 8
{
    F2Plan p{};
    int q = 0, lb = m;
    p.lr[q] = 3;
    p.ls[q] = lb - 3;
    lb -= 3;
    ++q;
    int rem = lb - 4;
    while (rem % 3 != 0) {
        const int r = rem >= 2 ? 2 : 1;
        p.lr[q] = r;
        p.ls[q] = lb - r;
        lb -= r;
        rem -= r;
        ++q;
    }
    while (rem > 0) {
        p.lr[q] = 3;
        p.ls[q] = lb - 3;
        lb -= 3;
        rem -= 3;
        ++q;
    }
    p.n_pass = q;
    return p;
}

size_t f3_lds_bytes(int m)
{
    const size_t np = ((size_t)1 << m) + ((size_t)1 << m >> 5) + 2;
    size_t tw = 0;
    const F2Plan pl = f3_plan(m);
    for (int q = 0; q < pl.n_pass; ++q) tw += (size_t)1 << pl.ls[q];
    return 2 * np * 8 + 256 * 16 + 32 * 8 + tw * 16;
}

// dft4 with a zero fourth input
__device__ __forceinline__ void dft4_3(Cx c0, Cx c1, Cx c2, Cx &z0, Cx &z1, Cx &z2, Cx &z3)
{
    const Cx d0 = cx_add(c0, c2), d1 = cx_sub(c0, c2), d3 = cx_mul_mi(c1);
    z0 = cx_add(d0, c1);
    z2 = cx_sub(d0, c1);
    z1 = cx_add(d1, d3);
    z3 = cx_sub(d1, d3);
}

// The radix-8 butterfly of the first pass with inputs a[0 .. QE) and zeros beyond; y[d] = output digit d, twiddled by
// w1^d. QE <= 4: the first stage (a_e +- a_{e+4}) is the identity.
template <int QE>
__device__ __forceinline__ void f3_head8(const Cx *a, Cx w1, Cx *y)
{
    if constexpr (QE <= 4) {
        constexpr double H = 0.70710678118654752440;
        const Cx b5 = {(a[1].x + a[1].y) * H, (a[1].y - a[1].x) * H};
        const Cx b6 = cx_mul_mi(a[2]);
        if constexpr (QE <= 3) {
            dft4_3(a[0], a[1], a[2], y[0], y[2], y[4], y[6]);
            dft4_3(a[0], b5, b6, y[1], y[3], y[5], y[7]);
        } else {
            const Cx b7 = {(a[3].y - a[3].x) * H, -(a[3].x + a[3].y) * H};
            dft4(a[0], a[1], a[2], a[3], y[0], y[2], y[4], y[6]);
            dft4(a[0], b5, b6, b7, y[1], y[3], y[5], y[7]);
        }
        const Cx w2 = cx_mul(w1, w1), w3 = cx_mul(w2, w1), w4 = cx_mul(w2, w2);
        const Cx w5 = cx_mul(w4, w1), w6 = cx_mul(w4, w2), w7 = cx_mul(w4, w3);
        y[1] = cx_mul(y[1], w1);
        y[2] = cx_mul(y[2], w2);
        y[3] = cx_mul(y[3], w3);
        y[4] = cx_mul(y[4], w4);
        y[5] = cx_mul(y[5], w5);
        y[6] = cx_mul(y[6], w6);
        y[7] = cx_mul(y[7], w7);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = e < QE ? a[e] : Cx{0.0, 0.0};
        f2_bfly8(y, w1, true);
    }
}

// One LDS pass (radix 2^LR at stride 2^LS) over the `len` points from `org` that ONE wave owns; no block barrier. The
// elements of a butterfly sit at fixed distances in the padded layout (e 2^LS + (e 2^LS >> 5): the butterfly's first
// point is at a multiple of 32 plus j, and j < 16 when the stride is 16), immediates of the LDS instructions.
// 8-byte LDS reads the compiler cannot pair: `ds_read2_b64` moves 16 bytes per lane in 8 LDS cycles, two `ds_read_b64`
// the same bytes in 4 (MI355X_MICROARCH.md, LDS table), and the backend pairs every two reads whose offsets allow it —
// all of the stride-16 pass, the tail and the partner reads. Hand-issued instead: f3_rd<byte offset>(LDS byte address),
// then ONE f3_wait_*() through which every loaded value passes (the dependency keeps their uses behind the wait; the
// compiler's own lgkmcnt bookkeeping does not know these reads, which only makes its waits stricter than needed).
// F3_NO_READ2 = 0 builds the plain C++ reads for A/B.
#ifndef F3_NO_READ2
#define F3_NO_READ2 1
#endif
__device__ __forceinline__ unsigned f3_lds_addr(const double *p) { return (unsigned)reinterpret_cast<size_t>(p); }

template <int OFF>
__device__ __forceinline__ double f3_rd(unsigned a)
{
    double v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF) : "memory");
    return v;
}

__device__ __forceinline__ void f3_wait_4(Cx *a)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[2].x), "+v"(a[2].y), "+v"(a[3].x),
                   "+v"(a[3].y)::"memory");
}

__device__ __forceinline__ void f3_pass_8(Cx *a)  // (no instruction: the values of a[0 .. 7] pass through)
{
    asm volatile(""
                 : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[2].x), "+v"(a[2].y), "+v"(a[3].x),
                   "+v"(a[3].y), "+v"(a[4].x), "+v"(a[4].y), "+v"(a[5].x), "+v"(a[5].y), "+v"(a[6].x), "+v"(a[6].y),
                   "+v"(a[7].x), "+v"(a[7].y)::"memory");
}

__device__ __forceinline__ void f3_wait_8(Cx *a)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[2].x), "+v"(a[2].y), "+v"(a[3].x),
                   "+v"(a[3].y), "+v"(a[4].x), "+v"(a[4].y), "+v"(a[5].x), "+v"(a[5].y), "+v"(a[6].x), "+v"(a[6].y),
                   "+v"(a[7].x), "+v"(a[7].y)::"memory");
}

// a[e] = point (first + e 2^LS) of a butterfly: ar / ai = LDS byte addresses of its first point in the two planes
template <int LS, int... E>
__device__ __forceinline__ void f3_rd_bfly(Cx *a, unsigned ar, unsigned ai, std::integer_sequence<int, E...>)
{
    ((a[E].x = f3_rd<8 * ((E << LS) + ((E << LS) >> 5))>(ar), a[E].y = f3_rd<8 * ((E << LS) + ((E << LS) >> 5))>(ai)), ...);
}

// a[e] = the 8-byte words at word offsets D0 + S e from ar / ai (S may be negative)
template <int D0, int S, int... E>
__device__ __forceinline__ void f3_rd_run(Cx *a, unsigned ar, unsigned ai, std::integer_sequence<int, E...>)
{
    ((a[E].x = f3_rd<8 * (D0 + S * E)>(ar), a[E].y = f3_rd<8 * (D0 + S * E)>(ai)), ...);
}

template <int LR>
__device__ __forceinline__ void f3_bfly(Cx *a, Cx w1)
{
    if constexpr (LR == 3) {
        f2_bfly8(a, w1, true);
    } else {
        Cx y0, y1, y2, y3;
        dft4(a[0], a[1], a[2], a[3], y0, y1, y2, y3);
        const Cx w2 = cx_mul(w1, w1), w3 = cx_mul(w2, w1);
        a[0] = y0;
        a[1] = cx_mul(y1, w1);
        a[2] = cx_mul(y2, w2);
        a[3] = cx_mul(y3, w3);
    }
}

#ifndef F3_PAIR
#define F3_PAIR 0  // 1: both butterflies of a lane in flight together where a pass has two. Measured at C4: 8.5 ms
                   // against 7.1 ms for the call (9.2 against 7.5 while the pair still spilled 50 registers: a scratch
                   // reload waits for the prefetched samples like any other vector-memory load) - eight waves a CU already
                   // keep the LDS queue full, a second butterfly per wave only lengthens every wave's wait for its data.
#endif
#ifndef F3_PRIO
#define F3_PRIO 0
#endif
#ifndef F3_SKIP
#define F3_SKIP 0  // timing experiments only (wrong results): 2 = no wave passes, 4 = no tail arithmetic, 8 = no partner reads
#endif
#ifndef F3_PREFETCH
#define F3_PREFETCH 1  // the next series' samples fetched under the sub-transforms (0: at the top of each series)
#endif
// `b_lo`, `b_hi`: the lane's butterflies b_lo + lane, b_lo + lane + 64, ... < b_hi of the pass (all of them by default;
// the SRC == 2 kernel runs a pass in two halves with a staging unit moved in between)
template <int LR, int LS>
__device__ __forceinline__ void f3_wave_pass(double *re, double *im, int org, int len, const double2 *tw, int lane,
                                             int b_lo = 0, int b_hi = 1 << 30)
{
    constexpr int R = 1 << LR, lb = LS + LR, s = 1 << LS;
    const int nb = len >> LR;
    const bool swz = LS == 4 && (len >> lb) >= 8;  // stride 16: the halves of a lane group take blocks 4 apart
    auto first = [&](int b) {
        const int j = b & (s - 1);
        int blk = b >> LS;
        if (swz) blk = (blk & ~5) | ((blk & 1) << 2) | ((blk >> 2) & 1);
        return f2_skew(org + (blk << lb) + j);
    };
#define F3_AT(i0, e) ((i0) + ((e) << LS) + (((e) << LS) >> 5))
    if (F3_PAIR && nb == 128) {
        // the second butterfly's reads are in flight under the first one's arithmetic, the first one's writes under
        // the second one's
        const int iA = first(lane), iB = first(lane + 64);
        const double2 wa = tw[lane & (s - 1)], wb = tw[(lane + 64) & (s - 1)];
        Cx a[R], c[R];
#pragma unroll
        for (int e = 0; e < R; ++e) a[e] = {re[F3_AT(iA, e)], im[F3_AT(iA, e)]};
#pragma unroll
        for (int e = 0; e < R; ++e) c[e] = {re[F3_AT(iB, e)], im[F3_AT(iB, e)]};
        f3_bfly<LR>(a, Cx{wa.x, wa.y});
#pragma unroll
        for (int e = 0; e < R; ++e) {
            re[F3_AT(iA, e)] = a[e].x;
            im[F3_AT(iA, e)] = a[e].y;
        }
        f3_bfly<LR>(c, Cx{wb.x, wb.y});
#pragma unroll
        for (int e = 0; e < R; ++e) {
            re[F3_AT(iB, e)] = c[e].x;
            im[F3_AT(iB, e)] = c[e].y;
        }
    } else {
        for (int b = b_lo + lane; b < nb && b < b_hi; b += 64) {
            const int i0 = first(b);
            const double2 wv = tw[b & (s - 1)];
            Cx a[R];
#if F3_NO_READ2
            f3_rd_bfly<LS>(a, f3_lds_addr(re + i0), f3_lds_addr(im + i0), std::make_integer_sequence<int, R>{});
            if constexpr (LR == 3) f3_wait_8(a);
            else f3_wait_4(a);
#else
#pragma unroll
            for (int e = 0; e < R; ++e) a[e] = {re[F3_AT(i0, e)], im[F3_AT(i0, e)]};
#endif
            f3_bfly<LR>(a, Cx{wv.x, wv.y});
#pragma unroll
            for (int e = 0; e < R; ++e) {
                re[F3_AT(i0, e)] = a[e].x;
                im[F3_AT(i0, e)] = a[e].y;
            }
        }
    }
#undef F3_AT
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// x, Qpart, Ppart as msd_power_lds2_kernel. JJ = first-pass butterflies per lane (N/8 / 512, at least 1), QE = inputs
// of a first-pass butterfly that can hold data (ceil(ceil(F / 2) / (N/8)) <= 8).
// DIRECT (round 4): x is the caller's trajectory itself, [F][cols] (= [F][3][E]), times `scale`: sample t of series c is
// x[t cols + c] — no transposed copy is made (rounds 2-3: transpose_scale_kernel wrote one and this kernel read it
// back: two thirds of the call's HBM traffic). A lane's samples then sit in lines of their own (a row is cols * 8
// bytes long), 8 useful bytes per 128-byte line — so the blocks are dealt series in CLUSTERS: the 16 blocks that the
// hardware places on one XCD one dispatch round after another (block b runs on XCD b % 8) take the 16 columns of one
// aligned tile, member k column 16 T + k, tile after tile (FftItem::step 16); they do the same work per series and stay
// within a fraction of an iteration of each other, so the line one of them brings into the XCD's L2 serves the other
// fifteen. Nothing but the hit rate depends on that placement or on the blocks keeping step: no barrier, no flag.
// SRC == 2 (round 4, the default where the shape allows it): x is the caller's trajectory as for SRC == 1, and the
// transposition happens INSIDE this kernel, spread over the transforms: the 16 blocks of a cluster walk the same
// 16-column tiles (one 128-byte line per row); member k copies rows [k Fc, (k + 1) Fc) of tile i + ST_AHEAD — whole
// lines, every line of the trajectory fetched once, by one block — into a scratch ring [cluster][tile % 8][16][16 Fc],
// time-major and scaled, a 16-byte load and two 8-byte stores per lane at five points of every series' iteration, and
// reads column k of tile i + 1 from there, 40 KB contiguous, as it read the transposed copy before. What rounds 2-4
// paid for the copy — a 2.2 ms pass at C4, and 6 GB of workspace — is gone; the ring is 16 x 8 x 16 x 16 Fc doubles (84 MB).
// Hand-off between blocks (MI355X_MICROARCH.md, "inter-workgroup visibility", first row of the measured table): the
// scratch is written and read with device-scope (sc1) accesses only — __hip_atomic_store / _load, relaxed, agent —;
// every wave waits for its stores (s_waitcnt vmcnt(0)) in front of a block barrier, behind which ONE lane adds to the
// tile's ready counter (agent-scope atomic); a consumer wave polls that counter itself with sc1 loads and reads the
// column only after its own poll has seen all 16 members. No placement assumption: any 16 co-resident blocks work.
// Every block must be resident (the grid is one block per CU, as the LDS footprint allows one): a poll that does not
// complete within ~2 s gives up, raises `stall` and the block runs to its end on whatever it reads (the host then
// takes the transposed path).
template <int JJ, int QE, int SRC, int UN = 5>
__global__ __launch_bounds__(FT_THREADS) void msd_power_lds3_kernel(
    const double *__restrict__ x, int F, int m, const FftItem *__restrict__ items,
    const double2 *__restrict__ tab, double *__restrict__ Qpart, double *__restrict__ Ppart, long long cols, double scale,
    const FftStage *__restrict__ stg, double *__restrict__ scratch, unsigned *__restrict__ ready, int Fc_arg)
{
    constexpr bool DIRECT = SRC == 1;
    // (a negative Fc is the test suite's request that ONE member withholds its signals: the polls then run out and
    // the host's repeat over the transposed copy is exercised — option lag_direct 3)
    const int Fc = Fc_arg < 0 ? -Fc_arg : Fc_arg;
    const bool withhold = Fc_arg < 0 && blockIdx.x == 0;
    constexpr int PR = 16;
    extern __shared__ double ft_lds[];
    const int N = 1 << m, np = N + (N >> 5) + 2;
    double *re = ft_lds, *im = ft_lds + np;
    double2 *tabA = reinterpret_cast<double2 *>(im + np), *tabB = tabA + 128;
    double *red = reinterpret_cast<double *>(tabB + 128);
    double2 *twp = reinterpret_cast<double2 *>(red + 32);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#if F3_PRIO
    // one of the two waves of every SIMD issues first whenever both can: its LDS reads are served first, it computes
    // while the other's are served, and the two stay half a phase apart instead of meeting at every LDS queue
    if (wv < 4) __builtin_amdgcn_s_setprio(F3_PRIO);
#endif
    if (tid < 256) tabA[tid] = tab[tid];
    const F2Plan pl = f3_plan(m);
    const FftItem it = items[blockIdx.x];
    __syncthreads();
    {
        int off = 0;
        for (int q = 0; q < pl.n_pass; ++q) {
            const int ls = pl.ls[q], lb = ls + pl.lr[q], s = 1 << ls;
            for (int j = tid; j < s; j += FT_THREADS) {
                const Cx w = ft_tw(tabA, tabB, j << (m + 1 - lb));
                twp[off + j] = make_double2(w.x, w.y);
            }
            off += s;
        }
    }
    // first-pass stride = points per wave afterwards; two first-pass butterflies per lane only at N = 8192
    const int ls0 = JJ == 2 ? 10 : m - 3, s0 = 1 << ls0;
    // lane < s0 / 16 of wave wv owns the 16 positions of tail block `lane` of its sub-transform
    const bool owner = lane < (s0 >> 4);
    const int pidx = owner ? wv * s0 + 16 * lane : 0;
    const int p0 = f2_skew(pidx);
    const int kb = f2_freq(pidx, pl);
    const int pb = f2_skew(f2_pos(((N >> 4) - kb) & ((N >> 4) - 1), pl));
    const bool kb0 = kb == 0;
#define F3_PARTNER(t) (pb + (kb0 ? f2_tail_neg(t) : 15 - (t)))
    double qa[JJ][QE], qb[JJ][QE], sacc[PR], tacc[9];
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj)
#pragma unroll
        for (int e = 0; e < QE; ++e) qa[jj][e] = qb[jj][e] = 0.0;
#pragma unroll
    for (int t = 0; t < PR; ++t) sacc[t] = 0.0;
#pragma unroll
    for (int u = 0; u < 9; ++u) tacc[u] = 0.0;
    const int half = (F + 1) >> 1;  // packed points that hold data
    double va[JJ][QE], vb[JJ][QE];
    // ---- SRC == 2: the cluster's staging ring ----
    typedef double st2_t __attribute__((ext_vector_type(2)));
    FftStage sg{};
    if constexpr (SRC == 2) sg = stg[blockIdx.x];
    const long long Fs = 16LL * Fc, nt = it.c_hi - it.c_lo;  // column stride of the ring; tiles of this cluster
    st2_t sx = {0.0, 0.0}, sy = {0.0, 0.0};                  // the units in flight
    // unit r of tile i (relative to c_lo): 64 rows x 128 bytes per round of the block. Wave w takes 16 of the rows
    // (w >> 1) and half of every line (w & 1); lanes l, l + 16, l + 32, l + 48 sit side by side in one row (64 bytes),
    // lanes l and l ^ 1 in consecutive rows: a load instruction touches 16 half lines (lane = row would touch 64 lines
    // for the same bytes: the vector-memory path works per line), and after the lane-pair swap below a store
    // instruction writes whole lines. What a lane needs per unit is kept to four registers beside the data (the kernel
    // has none to spare: a spilled register's reload waits for whatever load is in flight): its pointer into the
    // trajectory, its byte offset in the ring, and how many rounds it takes part in; the rest is scalar arithmetic.
    const int st_rr = (wv >> 1) * 16 + (lane & 15), st_p = (wv & 1) * 4 + (lane >> 4);
    // Nothing here branches: a lane that has no part in a unit (rows beyond the member's share or the series' end, tiles
    // beyond the cluster's last, columns beyond the matrix) addresses the buffer beyond its end — in the per-lane offset,
    // which is what the hardware's range check looks at; the scalar offset stays the same for all lanes (a per-lane
    // "scalar" operand makes the compiler wrap the instruction in a loop over its distinct values) —, where the hardware
    // returns zeros and drops stores. Branch-free matters for more than the branch: the compiler counts vector-memory
    // operations to place its waits, operations inside conditions count as "maybe not issued", and every wait behind
    // one turns into "wait for everything" — the unit just requested included (measured: 0.6 ms of the call).
    int st_lim = 0, st_lim2 = 0;  // unit r: this lane loads iff 64 r < st_lim, its pair stores iff 64 r < st_lim2
    unsigned st_vi = 0u, st_vo = 0u;  // the lane's byte offsets: in its member's rows of the trajectory, in the ring
    constexpr unsigned ST_OOB = 0xFFFFF000u;
    const bool odd_cols = (cols & 1LL) != 0;  // (uniform: one of two forms of the staging load for the whole launch)
    typedef unsigned st4_t __attribute__((ext_vector_type(4)));
    if constexpr (SRC == 2) {
        const int row = sg.k * Fc + st_rr, row2 = row & ~1;
        st_lim = Fc - st_rr < F - row ? Fc - st_rr : F - row;
        st_lim2 = Fc - (st_rr & ~1) < F - row2 ? Fc - (st_rr & ~1) : F - row2;
        st_vi = (unsigned)(((size_t)st_rr * (size_t)cols + 2 * st_p) * 8);
        st_vo = (unsigned)(((size_t)(2 * st_p + (lane & 1)) * (size_t)Fs + (size_t)row2) * 8);
    }
    // the member's rows of the trajectory, [k Fc, (k + 1) Fc) x cols doubles (the host checks: below 4 GB), and the ring of
    // this cluster, as buffer resources. The ring takes 16-byte device-scope (sc1) accesses — buffer_load / buffer_store
    // dwordx4 with the sc1 bit (aux 16) —, which the 8-byte __hip_atomic forms cannot give (an 8-byte sc1 store is a
    // fabric write of its own: the first version of this kernel, 8.8 ms per C4 call, was bound by 750 M of them).
    // (SRC != 2: scratch is null and the resources are never used)
    const long long st_rows = (long long)F - (long long)sg.k * Fc < Fc ? (long long)F - (long long)sg.k * Fc : (long long)Fc;
    const __amdgpu_buffer_rsrc_t traj = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(x) + (size_t)sg.k * (size_t)Fc * (size_t)cols, 0,
        (int)(unsigned)((size_t)(st_rows > 0 ? st_rows : 0) * (size_t)cols * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(
        scratch + (size_t)sg.cluster * ST_BUF * 16 * (size_t)Fs, 0, (int)((size_t)ST_BUF * 16 * (size_t)Fs * 8), 0x00020000);
    constexpr int SC1 = 16, NT_HINT = 2;
    bool st_on = true;  // (ST_SKIP & 2: off inside the series loop)
#ifndef ST_LOAD8
#define ST_LOAD8 0
#endif
    auto stage_load = [&](long long i, int r, st2_t &sv) {
        const long long T = it.c_lo + i;
        const bool row_in = st_on && i < nt && 64 * r < st_lim && !(ST_SKIP & 32);
        const unsigned soff = (unsigned)(((size_t)(64 * r) * (size_t)cols + (size_t)(16 * T)) * 8);
        if (ST_LOAD8 || odd_cols) {
            // an odd number of columns: rows start on odd multiples of 8 bytes and the matrix's last column has no
            // partner — two 8-byte loads, each with its own range test (only the matrix's last tile can reach beyond a
            // row's end, where the next row's first columns must not be taken for this tile's)
            typedef unsigned u2_t __attribute__((ext_vector_type(2)));
            const bool in0 = row_in && 16 * T + 2 * st_p < cols, in1 = row_in && 16 * T + 2 * st_p + 1 < cols;
            const u2_t a = __builtin_amdgcn_raw_buffer_load_b64(traj, in0 ? st_vi : ST_OOB, soff, NT_HINT);
            const u2_t b = __builtin_amdgcn_raw_buffer_load_b64(traj, in1 ? st_vi + 8u : ST_OOB, soff, NT_HINT);
            sv = st2_t{__builtin_bit_cast(double, a), __builtin_bit_cast(double, b)};
        } else {
            const bool in = row_in && (16 * T + 16 <= cols || 16 * T + 2 * st_p + 1 < cols);
            sv = __builtin_bit_cast(st2_t, __builtin_amdgcn_raw_buffer_load_b128(traj, in ? st_vi : ST_OOB, soff, NT_HINT));
        }
    };
    // Two lanes hold rows t, t + 1 (t even) of a column pair (a, b): they swap one value, so that the even lane has
    // (a[t], a[t + 1]) and the odd one (b[t], b[t + 1]) — 16 bytes of ONE column each, and a store instruction writes
    // whole 128-byte lines (eight lanes per line)
    auto stage_store = [&](long long i, int r, const st2_t &sv) {
        const bool odd = lane & 1;
        const double a = sv[0] * scale, b = sv[1] * scale;
        const double send = odd ? a : b;
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(send), 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(send), 0xB1, 0xF, 0xF, true);
        const double recv = __hiloint2double(hi, lo);
        const bool in = st_on && i < nt && 64 * r < st_lim2 && !(ST_SKIP & 16);
        const st2_t out = odd ? st2_t{recv, b} : st2_t{a, recv};
        const unsigned soff = (unsigned)(((size_t)(i & (ST_BUF - 1)) * 16 * (size_t)Fs + (size_t)(64 * r)) * 8);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(st4_t, out), ring, in ? st_vo : ST_OOB, soff, SC1);
        asm volatile("s_nop 1" ::"v"(out));  // (the store's data registers stay untouched for two cycles: see W12_STORE_GUARD)
    };
    auto st_flag = [&](long long i) { return ready + ((size_t)sg.cluster * ST_BUF + (size_t)(i & (ST_BUF - 1))) * ST_FLAG_STRIDE; };
    // (behind a block barrier that every wave entered after its own s_waitcnt vmcnt(0))
    auto st_signal = [&](long long i) {
        if (tid == 0 && i >= 0 && i < nt && !(withhold && i >= ST_AHEAD))
            __hip_atomic_fetch_add(st_flag(i), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    bool stalled = false;  // (wave-uniform) a poll gave up, here or in another block: no more waiting in this launch
    // (`seen` = an earlier look at the counter, requested a wave pass ago: the common case costs no round trip here)
    auto st_peek = [&](long long i) {
        return __hip_atomic_load(st_flag(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto st_wait = [&](long long i, unsigned seen) {
        const unsigned need = 16u * (unsigned)((i >> 3) + 1);
        if (stalled || seen >= need || (ST_SKIP & 1)) return;
        const unsigned *w = st_flag(i);
        unsigned *stall = ready + (size_t)(gridDim.x / 16) * ST_BUF * ST_FLAG_STRIDE;
        for (int spin = 1; __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need; ++spin) {
            __builtin_amdgcn_s_sleep(8);
            if ((spin & 255) == 0 && __hip_atomic_load(stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                stalled = true;
                break;
            }
            if (spin > (1 << 19)) {  // ~1 s: a member is not running
                __hip_atomic_store(stall, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stalled = true;
                break;
            }
        }
    };
    auto fetch = [&](long long c, unsigned seen = 0u) {
        if constexpr (SRC == 2) {
            // c = the tile; this member's column of it, from the ring (device-scope loads, after this wave's own poll)
            const long long i = c - it.c_lo, col = 16 * c + sg.k;
            const bool valid = col >= sg.lo && col < sg.hi;
            st_wait(i, seen);
            const unsigned row = (unsigned)((((size_t)(i & (ST_BUF - 1))) * 16 + sg.k) * (size_t)Fs * 8);  // (scalar)
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
#pragma unroll
                for (int e = 0; e < QE; ++e) {
                    const int j = tid + jj * FT_THREADS, n = j + (e << ls0);  // packed point: samples 2 n, 2 n + 1
                    // (beyond the series, or a column outside the segment: zeros from beyond the buffer's end; row F of
                    // the ring holds a zero where F is odd: the pair store wrote it)
                    const bool in = valid && j < s0 && 2 * n < F;
                    const st2_t v = __builtin_bit_cast(
                        st2_t, __builtin_amdgcn_raw_buffer_load_b128(ring, in ? (unsigned)n * 16u : ST_OOB, row, SC1));
                    va[jj][e] = v[0];
                    vb[jj][e] = v[1];
                }
            }
            return;
        }
        if constexpr (DIRECT) {
            const double *col = x + c;
#pragma unroll
            for (int jj = 0; jj < JJ; ++jj) {
#pragma unroll
                for (int e = 0; e < QE; ++e) {
                    const int j = tid + jj * FT_THREADS, n = j + (e << ls0);  // packed point: samples 2 n, 2 n + 1
                    va[jj][e] = vb[jj][e] = 0.0;
                    if (j < s0 && 2 * n < F) va[jj][e] = col[(size_t)(2 * n) * (size_t)cols] * scale;
                    if (j < s0 && 2 * n + 1 < F) vb[jj][e] = col[(size_t)(2 * n + 1) * (size_t)cols] * scale;
                }
            }
            return;
        }
        const double *row = x + (size_t)c * F;
        const bool al16 = (reinterpret_cast<unsigned long long>(row) & 15ull) == 0ull;
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
#pragma unroll
            for (int e = 0; e < QE; ++e) {
                const int j = tid + jj * FT_THREADS, n = j + (e << ls0);  // packed point: samples 2 n, 2 n + 1
                va[jj][e] = vb[jj][e] = 0.0;
                if (j < s0 && 2 * n + 1 < F) {
                    if (al16) {
                        typedef double d2_t __attribute__((ext_vector_type(2)));
                        const d2_t v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(row + 2 * n));
                        va[jj][e] = v[0];
                        vb[jj][e] = v[1];
                    } else {
                        va[jj][e] = row[2 * n];
                        vb[jj][e] = row[2 * n + 1];
                    }
                } else if (j < s0 && 2 * n < F) {
                    va[jj][e] = row[2 * n];
                }
            }
        }
    };
    auto lane_sum = [&]() {
        double sum = 0.0;
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj)
#pragma unroll
            for (int e = 0; e < QE; ++e) sum += va[jj][e] + vb[jj][e];
        return sum;
    };
    double mean = 0.0;
    if constexpr (SRC == 2) {
        // the first ST_AHEAD tiles, before anything is transformed
        for (int i = 0; i < ST_AHEAD; ++i)
            for (int r = 0; r < UN; ++r) {
                stage_load(i, r, sx);
                stage_store(i, r, sx);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = 0; i < ST_AHEAD; ++i) st_signal(i);
    }
    if (it.c_lo < it.c_hi) {  // (a cluster member whose column lies outside the segment has nothing to do)
        fetch(it.c_lo);
        mean = ft_block_sum(lane_sum(), red) / (double)F;
    }
    const int cstep = SRC == 2 ? 1 : it.step;
    if (ST_SKIP & 2) st_on = false;
    for (long long c = it.c_lo; c < it.c_hi; c += cstep) {
        // SRC == 2: tile st_i is staged under this series, five 16-byte units per lane, each stored TWO points of the
        // iteration after it was requested (4100 - 6500 cycles: a load from HBM takes ~2500 here, and with one point
        // between request and use the kernel stalled at every one of them): requests at A (here), B (behind the first
        // pass), C (middle of the first wave pass), C2 (between the wave passes), D (middle of the second); stores at
        // C, C2, D, E (behind the wave passes) and A. Two units in flight from B to E, one elsewhere — the registers for
        // the second come from the next series' samples, which this variant requests at E, behind the wave passes,
        // instead of at B (their block sum then rides on the THIRD barrier).
        const long long st_i = c - it.c_lo + ST_AHEAD;
        // Point P of the iteration (NP points: A, [A2,] B, C, C2, D, E [, G]; the bracketed ones only with UN = 8 units,
        // F > 5120): the unit requested two points ago is stored — P - 2 of this tile, or P - 2 + NP of the tile staged
        // under the previous series — and unit P is requested, units alternating between the two register pairs
        constexpr int NP = UN == 5 ? 6 : 8;
        constexpr int P_A = 0, P_A2 = UN == 5 ? -1 : 1, P_B = UN == 5 ? 1 : 2, P_C = P_B + 1, P_C2 = P_B + 2, P_D = P_B + 3,
                      P_E = P_B + 4, P_G = UN == 5 ? -1 : 7;
        auto point = [&](auto pk) {
            constexpr int P = decltype(pk)::value;
            if constexpr (SRC == 2 && P >= 0) {
                st2_t &reg = (P & 1) ? sy : sx;
                if constexpr (P >= 2) {
                    if constexpr (P - 2 < UN) stage_store(st_i, P - 2, reg);
                } else if constexpr (P - 2 + NP < UN) {
                    if (c > it.c_lo) stage_store(st_i - 1, P - 2 + NP, reg);
                }
                if constexpr (P < UN) stage_load(st_i, P, reg);
            }
        };
#define ST_POINT(P) point(std::integral_constant<int, (P)>())
        ST_POINT(P_A);
        // first pass, from the registers (the previous series' partner reads are behind the barrier that ended it)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            if (jj == 1) ST_POINT(P_A2);
            const int j = tid + jj * FT_THREADS;
            if (j < s0) {
                Cx a[QE], y[8];
#pragma unroll
                for (int e = 0; e < QE; ++e) {
                    const int n = j + (e << ls0);
                    double da = 0.0, db = 0.0;
                    if (n < half) {
                        da = va[jj][e] - mean;
                        db = 2 * n + 1 < F ? vb[jj][e] - mean : 0.0;
                    }
                    qa[jj][e] = __builtin_fma(da, da, qa[jj][e]);
                    qb[jj][e] = __builtin_fma(db, db, qb[jj][e]);
                    a[e] = {da, db};
                }
                const double2 wv1 = twp[j];
                f3_head8<QE>(a, Cx{wv1.x, wv1.y}, y);
                const int i0 = f2_skew(j), st = s0 + (s0 >> 5);  // (s0 is a multiple of 32)
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    re[i0 + d * st] = y[d].x;
                    im[i0 + d * st] = y[d].y;
                }
            }
        }
        const bool more = (F3_PREFETCH || SRC == 2) && c + cstep < it.c_hi;
        if constexpr (SRC == 2) ST_POINT(P_B);
        else if (more) fetch(c + cstep);
        __syncthreads();
        // this wave's sub-transform: the remaining LDS passes, then the tail in registers
        {
            // the (radix, stride) pairs f3_plan makes, spelled out per m: a plan array indexed at run time would
            // live in scratch memory, and a scratch load waits for the prefetch above like any other vector load
            const int org = wv * s0;
            const double2 *tw1 = twp + s0;
            unsigned seen = 0u;
            auto pc = [&]() {
                if constexpr (SRC == 2) {
                    // what this wave stored for the tile staged under the PREVIOUS series (its last units at A / A2) has
                    // long been issued: waiting for everything in flight here is free (the youngest request is a phase
                    // old), and lets the second barrier below carry the signal for that tile
                    if (!(ST_SKIP & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ST_POINT(P_C);
                }
            };
            auto pc2 = [&]() {
                if constexpr (SRC == 2) {
                    if (more) seen = st_peek(c + cstep - it.c_lo);
                    ST_POINT(P_C2);
                }
            };
            auto pd = [&]() { ST_POINT(P_D); };
#if F3_SKIP & 2
            if (F == 1) {
#else
            if (JJ == 2 || m == 13) {
#endif
                if constexpr (SRC == 2) {
                    // (two butterflies per lane and pass: the halves of a pass touch different points, no barrier between)
                    // (the bound is opaque: with trip counts it can see, the compiler unrolls the halves into one
                    // schedule with both butterflies' reads in flight — 120 spilled registers, as F3_PAIR found)
                    int h64 = 64;
                    asm volatile("" : "+s"(h64));
                    f3_wave_pass<3, 7>(re, im, org, s0, tw1, lane, 0, h64);
                    pc();
                    f3_wave_pass<3, 7>(re, im, org, s0, tw1, lane, h64);
                    pc2();
                    f3_wave_pass<3, 4>(re, im, org, s0, tw1 + 128, lane, 0, h64);
                    pd();
                    f3_wave_pass<3, 4>(re, im, org, s0, tw1 + 128, lane, h64);
                } else {
                    f3_wave_pass<3, 7>(re, im, org, s0, tw1, lane);
                    f3_wave_pass<3, 4>(re, im, org, s0, tw1 + 128, lane);
                }
            } else if (m == 12) {
                f3_wave_pass<2, 7>(re, im, org, s0, tw1, lane);
                pc();
                pc2();
                f3_wave_pass<3, 4>(re, im, org, s0, tw1 + 128, lane);
                pd();
            } else if (m == 11) {
                f3_wave_pass<2, 6>(re, im, org, s0, tw1, lane);
                pc();
                pc2();
                f3_wave_pass<2, 4>(re, im, org, s0, tw1 + 64, lane);
                pd();
            } else if (m == 10) {
                f3_wave_pass<3, 4>(re, im, org, s0, tw1, lane);
                pc();
                pc2();
                pd();
            } else {
                f3_wave_pass<2, 4>(re, im, org, s0, tw1, lane);
                pc();
                pc2();
                pd();
            }
            if constexpr (SRC == 2) {
                ST_POINT(P_E);
                if (more) fetch(c + cstep, seen);
            }
        }
        Cx z[PR];
        if (owner) {
#if F3_NO_READ2
            f3_rd_run<0, 1>(z, f3_lds_addr(re + p0), f3_lds_addr(im + p0), std::make_integer_sequence<int, PR>{});
            f3_wait_8(z);
            f3_pass_8(z + 8);
#else
#pragma unroll
            for (int e = 0; e < PR; ++e) z[e] = {re[p0 + e], im[p0 + e]};
#endif
#if !(F3_SKIP & 4)
            f2_dft16(z);
#endif
            // only positions 8 .. 15 go back to LDS: they are what the partner block's lane reads (its 15 - u,
            // u < 8); lane 0 of wave 0 pairs inside its own block and takes its partners from the registers
#pragma unroll
            for (int e = 8; e < PR; ++e) {
                re[p0 + e] = z[e].x;
                im[p0 + e] = z[e].y;
            }
#pragma unroll
            for (int t = 0; t < PR; ++t) {
                sacc[t] = __builtin_fma(z[t].x, z[t].x, sacc[t]);
                sacc[t] = __builtin_fma(z[t].y, z[t].y, sacc[t]);
            }
        }
        if (owner && kb0) {  // (one lane of the block) T for its 9 pairs, all inside its own block
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t0 = u < 3 ? u : u + 1;  // 0 1 2 4 5 6 7 8
                tacc[u] = __builtin_fma(z[t0].x, z[f2_tail_neg(t0)].y, tacc[u]);
                tacc[u] = __builtin_fma(z[t0].y, z[f2_tail_neg(t0)].x, tacc[u]);
            }
            tacc[8] = __builtin_fma(z[9].x, z[f2_tail_neg(9)].y, tacc[8]);
            tacc[8] = __builtin_fma(z[9].y, z[f2_tail_neg(9)].x, tacc[8]);
        }
        if (more && SRC != 2) {  // the next series' block sum rides on the barrier below
            double v = lane_sum();
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) red[8 + wv] = v;
        }
        __syncthreads();
        if constexpr (SRC == 2) {
            if (c > it.c_lo && !(ST_SKIP & 4)) st_signal(st_i - 1);  // (the prologue signalled its own tiles)
            ST_POINT(P_G);
        }
#if F3_SKIP & 8
        if (owner && F == 1) {
#else
        if (owner && !kb0) {
#endif
            // T at 8 of the 16 positions, the partner block's lane has the other 8 (see msd_power_lds2_kernel)
            Cx pz[8];  // pz[u] = the partner block's position 15 - u
#if F3_NO_READ2
            f3_rd_run<15, -1>(pz, f3_lds_addr(re + pb), f3_lds_addr(im + pb), std::make_integer_sequence<int, 8>{});
            f3_wait_8(pz);
#else
#pragma unroll
            for (int u = 0; u < 8; ++u) pz[u] = {re[pb + 15 - u], im[pb + 15 - u]};
#endif
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                tacc[u] = __builtin_fma(z[u].x, pz[u].y, tacc[u]);
                tacc[u] = __builtin_fma(z[u].y, pz[u].x, tacc[u]);
            }
        }
        if (more && SRC != 2) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < FT_THREADS / 64; ++w) sum += red[8 + w];
            mean = sum / (double)F;
        }
        if (more && SRC == 2) {  // (requested behind the wave passes: its block sum rides on this last barrier)
            double v = lane_sum();
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) red[8 + wv] = v;
        }
        __syncthreads();
        if (more && SRC == 2) {
            // (nobody writes red[8 ...] again before the first barrier of the next series, which this wave joins
            // after these reads)
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < FT_THREADS / 64; ++w) sum += red[8 + w];
            mean = sum / (double)F;
        }
        if (!F3_PREFETCH && SRC != 2 && c + cstep < it.c_hi) {
            fetch(c + cstep);
            mean = ft_block_sum(lane_sum(), red) / (double)F;
        }
    }
#undef ST_POINT
    double *q = Qpart + (size_t)it.row * F, *pp = Ppart + (size_t)it.row * (N + 1);
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) {
#pragma unroll
        for (int e = 0; e < QE; ++e) {
            const int j = tid + jj * FT_THREADS, n = j + (e << ls0);
            if (j < s0 && 2 * n < F) q[2 * n] = qa[jj][e];
            if (j < s0 && 2 * n + 1 < F) q[2 * n + 1] = qb[jj][e];
        }
    }
    // frequencies, once per block (the loop ended on a barrier): as msd_power_lds2_kernel
    if (owner) {
#pragma unroll
        for (int t = 0; t < PR; ++t) re[p0 + t] = sacc[t];
#pragma unroll
        for (int u = 0; u < 8; ++u) im[p0 + (kb0 ? (u < 3 ? u : u + 1) : u)] = tacc[u];
        if (kb0) im[p0 + 9] = tacc[8];
    }
    __syncthreads();
    if (owner) {
#pragma unroll
        for (int t = 0; t < PR; ++t) {
            const int k = f2_freq(pidx + t, pl);
            const bool mine = kb0 ? (t != 3 && t < 10) : t < 8;
            const double sk = sacc[t], sn = re[F3_PARTNER(t)], tk = im[mine ? p0 + t : F3_PARTNER(t)];
            const Cx w = ft_tw(tabA, tabB, k);  // (cos, -sin) of 2 pi k / L
            pp[k] = 0.5 * (sk + sn) + w.y * (0.5 * (sk - sn)) + w.x * tk;
            if (k == 0) pp[N] = sk - tk;
        }
    }
#undef F3_PARTNER
}

#include "msd_fft_w12.h"
#include "msd_fft_w12r.h"

// out[s][i] = sum over the items of segment s of part[item][i]
__global__ void fold_items_kernel(const double *__restrict__ part, const int *__restrict__ seg_item_off,
                                  long long width, double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width) return;
    const int s = blockIdx.y;
    double acc = 0.0;
    // (eight rows requested before the first is added — the additions keep their order: same bits — instead of a load
    // and a wait per row: the grid is only (width / 256) x segments blocks, 60 at C4, and took 26 + 30 us per call)
    int q = seg_item_off[s];
    const int q1 = seg_item_off[s + 1];
    for (; q + 8 <= q1; q += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(part + (size_t)(q + u) * width + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; q < q1; ++q) acc += part[(size_t)q * width + i];
    out[(size_t)s * width + i] = acc;
}

// corr[s][k] = sum_t x(t) x(t+k) summed over the segment = inverse real transform of P[s][0..N], k < n_lags
__global__ __launch_bounds__(FT_THREADS) void msd_inverse_lds_kernel(const double *__restrict__ P, int m,
                                                                     const double2 *__restrict__ tab,
                                                                     int n_lags, double *__restrict__ corr)
{
    extern __shared__ double ft_lds[];
    const int N = 1 << m, np = N + (N >> 4) + 2;
    double *re = ft_lds, *im = ft_lds + np;
    double2 *tabA = reinterpret_cast<double2 *>(im + np), *tabB = tabA + 128;
    const int tid = threadIdx.x;
    if (tid < 256) tabA[tid] = tab[tid];
    __syncthreads();
    const double *p = P + (size_t)blockIdx.x * (N + 1);
    // packed half-length spectrum of the real, even sequence: conj(Z_k), Z_k = E_k + i O_k
    for (int k = tid; k < N; k += FT_THREADS) {
        const double pk = p[k], pn = p[N - k];
        const double e = 0.5 * (pk + pn), d = 0.5 * (pk - pn);
        const Cx w = ft_tw(tabA, tabB, k);  // (cos, -sin)
        const int n = ft_skew(k);
        re[n] = e + d * w.y;
        im[n] = -d * w.x;
    }
    __syncthreads();
    ft_transform(re, im, m, tabA, tabB);
    const double inv = 1.0 / (double)N;
    double *out = corr + (size_t)blockIdx.x * n_lags;
    for (int k = tid; k < n_lags; k += FT_THREADS) {
        const int n = ft_skew(ft_pos(k >> 1, m));
        out[k] = (k & 1) ? -im[n] * inv : re[n] * inv;
    }
}

__global__ __launch_bounds__(256) void transpose_scale_kernel(const double *__restrict__ in,
                                                              double *__restrict__ out, long long rows,
                                                              long long cols, double scale)
{
    // out[col][row] = in[row][col] * scale, 64 x 64 tiles through LDS: a wave reads and writes runs of 64 doubles
    // (512 B; the 32 x 32 tiles of round 2 moved 256-byte runs: 4.2 TB/s of read + write at C4), every element is
    // touched once in each direction: nontemporal. With even `rows` and `cols` and 16-byte aligned bases (V2) a lane
    // moves 16 bytes per access in both directions — half the vector-memory instructions.
    __shared__ double tile[64][65];
    const long long c0 = (long long)blockIdx.x * 64, r0 = (long long)blockIdx.y * 64;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const bool v2 = ((rows | cols) & 1LL) == 0 &&
                    ((reinterpret_cast<unsigned long long>(in) | reinterpret_cast<unsigned long long>(out)) & 15ull) == 0;
    if (v2) {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 pairs x 8
#pragma unroll 8
        for (int k = ty; k < 64; k += 8) {
            const long long rr = r0 + k, cc = c0 + 2 * tx;
            d2_t v = {0.0, 0.0};
            if (rr < rows && cc < cols) v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(in + rr * cols + cc));
            tile[k][2 * tx] = v[0] * scale;
            tile[k][2 * tx + 1] = v[1] * scale;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = ty; k < 64; k += 8) {
            const long long cc = c0 + k, rr = r0 + 2 * tx;
            if (rr < rows && cc < cols) {
                const d2_t v = {tile[2 * tx][k], tile[2 * tx + 1][k]};
                __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(out + cc * rows + rr));
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll 4
    for (int k = ty; k < 64; k += 4) {
        const long long rr = r0 + k, cc = c0 + tx;
        tile[k][tx] = (rr < rows && cc < cols) ? __builtin_nontemporal_load(in + rr * cols + cc) * scale : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = ty; k < 64; k += 4) {
        const long long cc = c0 + k, rr = r0 + tx;
        if (rr < rows && cc < cols) __builtin_nontemporal_store(tile[tx][k], out + cc * rows + rr);
    }
}

// transform length of the large-lag path: the next power of two (fft_pow2.hip)
long long pow2_length(long long n)
{
    long long L = 2;
    while (L < n) L <<= 1;
    return L;
}

// ---------------------------------------------------------------------------------------------
// Round 5: the finish of the fused path ON THE DEVICE (rounds 2-4: a host pass in long double behind a
// device-to-host copy of Q and the correlations and a host wait in the middle of every call — 0.8 of the 1.03 ms a C4
// step took on the shard one of eight ranks holds). What the host did with a 64-bit mantissa is done here in
// double-double (two-sum arithmetic, ~106 bits): prefix sums of the per-frame squares Q, S1(k) = pre[F - k] + (pre[F] -
// pre[k]), v = S1 - 2 S2, the means v / ((F - k) n_g) and their total (x + y) + z, and the same error bound,
// eps_l * 2 pre[F] / min |v|. One block per group, the three axes one after the other; pre (hi, lo) in a workspace.
// -ffp-contract / the pragma above only fuse multiply-adds: the two-sums below hold nothing but additions.
// ---------------------------------------------------------------------------------------------
struct DD {
    double hi, lo;
};
__device__ __forceinline__ DD dd_two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ DD dd_add(DD a, DD b)
{
    DD s = dd_two_sum(a.hi, b.hi);
    const DD t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_two_sum(s.hi, s.lo);  // (fast two-sum would do: |hi| >= |lo|)
    s.lo += t.lo;
    return dd_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ DD dd_add_d(DD a, double b)
{
    DD s = dd_two_sum(a.hi, b);
    s.lo += a.lo;
    return dd_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ DD dd_neg(DD a) { return {-a.hi, -a.lo}; }

// the relative bound a lag must keep for the spectral result to stand (lag_variant 3), and the words behind a call's
// bounds: S bounds | 2 status words | per segment (lo_max, hi_min, the bound of the lags that keep LAG_BOUND_OK)
constexpr double LAG_BOUND_OK = 1e-10;
inline size_t lag_bound_words(long long S) { return (size_t)(4 * S + 2); }

// what the completion step reads from the words a call copied back
inline void lag_collect_bounds(LagFftResult *res, const double *h, long long S, long long n_lags)
{
    double worst = 0.0, lo = 0.0, hi = (double)n_lags, ok = 0.0;
    for (long long q = 0; q < S; ++q) {
        worst = std::max(worst, h[q]);
        lo = std::max(lo, h[S + 2 + 3 * q]);
        hi = std::min(hi, h[S + 2 + 3 * q + 1]);
        ok = std::max(ok, h[S + 2 + 3 * q + 2]);
    }
    res->bound = worst;
    res->k_lo = (long long)lo;
    res->k_hi = (long long)hi;
    res->bound_ok = ok;
    res->ends_valid = true;
}

// Q [3 G][F], corr [3 G][corr_row] (S2(k) = corr * corr_scale), n_g [G] entities per group -> out [n_lags][G][4] (device:
// components 0..2 here, their total by lag_total_kernel), bound [lag_bound_words(3 G)]: the segments' bounds, two status
// words (lag_total_kernel), and per segment which lags miss LAG_BOUND_OK; pre: workspace [3 G][F + 1] of DD.
// One block per (axis, group) segment.
__global__ __launch_bounds__(256) void lag_finish_dd_kernel(const double *__restrict__ Q, const double *__restrict__ corr,
                                                           long long corr_row, double corr_scale, long long F, long long n_lags,
                                                           int G, const double *__restrict__ n_g, double eps_l,
                                                           DD *__restrict__ pre_ws, double *__restrict__ out,
                                                           double *__restrict__ bound)
{
    __shared__ DD wtot[4];
    __shared__ double vmin_s[4];
    __shared__ double ends_s[4][3];
    const int s = blockIdx.x, a = s / G, g = s % G, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double ng = n_g[g];
    const long long chunk = (F + 255) / 256;
    const double *q = Q + (size_t)s * F;
    DD *pre = pre_ws + (size_t)s * (F + 1);
    const long long i0 = (long long)tid * chunk, i1 = i0 + chunk < F ? i0 + chunk : F;
    DD acc = {0.0, 0.0};
    for (long long i = i0; i < i1; ++i) acc = dd_add_d(acc, q[i]);
    // inclusive scan of the 256 chunk sums: inside the wave by shuffles, the four wave totals in order
    DD incl = acc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const DD up = {__shfl_up(incl.hi, d, 64), __shfl_up(incl.lo, d, 64)};
        if (lane >= d) incl = dd_add(up, incl);
    }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    DD before = {0.0, 0.0};
    for (int w = 0; w < wv; ++w) before = dd_add(before, wtot[w]);
    DD tot = before;
    for (int w = wv; w < 4; ++w) tot = dd_add(tot, wtot[w]);
    // exclusive prefix of this thread's chunk = before + (incl - acc): recomputed as before + (sum of the lanes below)
    DD run = before;
    {
        const DD below = {__shfl_up(incl.hi, 1, 64), __shfl_up(incl.lo, 1, 64)};
        if (lane > 0) run = dd_add(before, below);
    }
    for (long long i = i0; i < i1; ++i) {
        pre[i] = run;
        run = dd_add_d(run, q[i]);
    }
    if (tid == 255) pre[F] = tot;
    __threadfence_block();
    __syncthreads();
    double vmin = 0.0;  // smallest non-zero |S1 - 2 S2| over the lags k > 0
    // (round 6) which lags miss LAG_BOUND_OK: the transform's error is the same absolute amount at every lag, so a lag misses
    // the bound iff its |S1 - 2 S2| lies below vthr. For diffusive and ballistic data those are the first few lags and the
    // last few (few origins): lo_max = the largest such k in the lower half of the lag range, hi_min = the smallest one in the
    // upper half, vok = the smallest |.| among the lags that keep the bound.
    const double e2 = eps_l * (2.0 * (tot.hi + tot.lo)), vthr = e2 / LAG_BOUND_OK;
    long long lo_max = 0, hi_min = n_lags;
    double vok = 0.0;
    for (long long k = tid; k < n_lags; k += 256) {
        const DD s1 = dd_add(pre[F - k], dd_add(tot, dd_neg(pre[k])));
        const double s2 = corr[(size_t)s * corr_row + k] * corr_scale;
        const DD vd = dd_add_d(s1, -2.0 * s2);
        double v = vd.hi + vd.lo;
        if (k == 0) v = 0.0;  // exactly, as the difference form gives
        const double cnt = (double)(F - k) * ng;
        out[((size_t)k * G + g) * 4 + a] = cnt > 0.0 ? v / cnt : 0.0;
        const double av = fabs(v);
        if (k > 0 && av != 0.0 && (vmin == 0.0 || av < vmin)) vmin = av;
        if (k > 0 && av != 0.0) {
            if (av < vthr) {
                if (2 * k < n_lags) lo_max = k > lo_max ? k : lo_max;
                else hi_min = k < hi_min ? k : hi_min;
            } else if (vok == 0.0 || av < vok) {
                vok = av;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double other = __shfl_down(vmin, o, 64);
        if (other != 0.0 && (vmin == 0.0 || other < vmin)) vmin = other;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const long long lo_o = __shfl_down(lo_max, o, 64), hi_o = __shfl_down(hi_min, o, 64);
        const double vo = __shfl_down(vok, o, 64);
        lo_max = lo_o > lo_max ? lo_o : lo_max;
        hi_min = hi_o < hi_min ? hi_o : hi_min;
        if (vo != 0.0 && (vok == 0.0 || vo < vok)) vok = vo;
    }
    if (lane == 0) {
        vmin_s[wv] = vmin;
        ends_s[wv][0] = (double)lo_max;
        ends_s[wv][1] = (double)hi_min;
        ends_s[wv][2] = vok;
    }
    __syncthreads();
    if (tid == 0) {
        double m = 0.0, lo = 0.0, hi = (double)n_lags, ok = 0.0;
        for (int w = 0; w < 4; ++w) {
            if (vmin_s[w] != 0.0 && (m == 0.0 || vmin_s[w] < m)) m = vmin_s[w];
            lo = ends_s[w][0] > lo ? ends_s[w][0] : lo;
            hi = ends_s[w][1] < hi ? ends_s[w][1] : hi;
            if (ends_s[w][2] != 0.0 && (ok == 0.0 || ends_s[w][2] < ok)) ok = ends_s[w][2];
        }
        double *ends = bound + gridDim.x + 2 + 3 * s;  // (behind the S bounds and the two status words)
        ends[0] = lo;
        ends[1] = hi;
        ends[2] = ok > 0.0 ? e2 / ok : 0.0;
        // the transform's rounding error in S2(k) scales with the energy of the WHOLE series at every lag: the worst
        // relative error is at the lag with the smallest |v| (2 pre[F] >= S1(k), equal at small lags)
        bound[s] = m > 0.0 ? eps_l * (2.0 * (tot.hi + tot.lo)) / m : 0.0;
    }
}

// out[k][g][3] = (x + y) + z; bound[S] = the largest of the S segment bounds, or +infinity when the staging ring of the
// power kernel stalled (`stall` non-null and set): the call's STATUS as one device-resident number, which the multi-GPU
// step passes through its all-reduce so that every rank learns of a result that must be redone without a host wait
__global__ void lag_total_kernel(double *__restrict__ out, long long n, double *__restrict__ bound, int S,
                                 const unsigned *__restrict__ stall)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[4 * i + 3] = (out[4 * i] + out[4 * i + 1]) + out[4 * i + 2];
    if (i == 0) {
        double m = 0.0;
        for (int q = 0; q < S; ++q) m = fmax(m, bound[q]);
        if (stall && *stall) m = __longlong_as_double(0x7ff0000000000000LL);
        bound[S] = m;
    }
}

// The fused LDS path: L = 2^(m+1) <= 16384.
// w12 (round 5): N = 6144 = 12 x 512, L = 12288 (msd_fft_w12.h) instead of N = 2^m; `m` is ignored then.
// The means go to `out` (host or device memory) from here: finished on the device, copied on the stream; res->bound is
// known when the call has completed.
int lag_msd_fft_fused(CallScope &cs, long long F, long long E, const double *d_r, double scale, int max_lag,
                      long long G, const int64_t *group_off, int m, const std::shared_ptr<LagFftResult> &res,
                      double *out, int out_on_device, int src_want = -1 /* -1: the context's option lag_direct */,
                      bool w12 = false)
{
    mdhip_ctx *ctx = cs.ctx;
    const long long n_lags = (long long)max_lag + 1, cols = 3 * E, S = 3 * G;
    const long long N = w12 ? (long long)W12_N : 1LL << m, L = 2 * N;
    const size_t lds_b = w12 ? (size_t)W12_N * 16 + 256 * 16 : ft_lds_bytes(m);
    // round-3 kernel (conflict-free layout, bilinear spectrum accumulation): N = 2^m a multiple of the block size
    const bool v2 = !w12 && ctx->opt_lag_fft_kernel != 0 && m >= 9 && f2_lds_bytes(m) <= ctx->lds_max;
    // second step of round 3 (first pass from registers, wave-private sub-transforms): lag_fft_kernel >= 2
    const bool v3 = !w12 && ctx->opt_lag_fft_kernel >= 2 && m >= F3_MIN_M && f3_lds_bytes(m) <= ctx->lds_max;
    // round 4, option `lag_direct` (off by default: measured slower, see ctx.h): that kernel reads the trajectory as it
    // is, [F][3 E], no transposed copy (see msd_power_lds3_kernel). Needs the blocks in whole clusters of 16 per XCD:
    // 128 | cu_count.
    // round 4, option `lag_direct` = 2 (default): no transposed copy either, the clusters of 16 blocks transpose their own
    // tiles inside the kernel through a small ring (SRC == 2 of msd_power_lds3_kernel): any 16 blocks, rows per member
    // Fc = F / 16 rounded up to whole 128-byte lines, at most 64 ST_UNITS.
    const int n_clusters = ctx->cu_count / 16;
#ifndef LAG_DIRECT_DEFAULT
#define LAG_DIRECT_DEFAULT 2
#endif
    const int src_opt = src_want >= 0 ? src_want : ctx->opt_lag_direct >= 0 ? ctx->opt_lag_direct : LAG_DIRECT_DEFAULT;
    const int Fc = (int)((((F + 15) / 16) + 15) / 16 * 16);
    const bool staged = (v3 || w12) && (src_opt == 2 || src_opt == 3) && ctx->cu_count % 16 == 0 && n_clusters >= 1 &&
                        (w12 ? Fc <= 16 * (W12_NW / 2) * W12_UN
                             : Fc <= 64 * ST_UNITS && (Fc <= 64 * 5 || (m == 13 && F <= 8192))) &&  // (eight units: the N = 8192 kernels only)
                        cols >= 16 * (long long)n_clusters &&
                        (unsigned long long)Fc * (unsigned long long)cols * 8ull < 0xFFFFF000ull &&  // (a member's rows: one buffer)
                        (reinterpret_cast<unsigned long long>(d_r) & 15ull) == 0ull;  // (16-byte loads of column pairs
                                                                                     // where the column count is even)
    const bool direct = staged || (v3 && src_opt == 1 && ctx->cu_count % 128 == 0 && cols >= 16 * (long long)n_clusters);
    // work items: every non-empty segment gets a share of ~one block per CU, each a contiguous series range — or, for
    // the direct-read kernel, whole clusters of 16 blocks that walk the segment's 16-column tiles (aligned to 16 columns
    // of the [F][cols] matrix = one 128-byte line per row), member k taking column 16 T + k
    std::vector<FftItem> items;
    std::vector<FftStage> stages;
    std::vector<int> seg_off((size_t)S + 1, 0);
    if (!direct) {
        for (long long s = 0; s < S; ++s) {
            const long long a = s / G, g = s % G;
            const long long lo = a * E + group_off[g], hi = a * E + group_off[g + 1], n = hi - lo;
            seg_off[s] = (int)items.size();
            if (n <= 0) continue;
            long long k = (n * ctx->cu_count + cols / 2) / cols;
            k = std::max<long long>(1, std::min(k, n));
            for (long long q = 0; q < k; ++q)
                items.push_back({lo + n * q / k, lo + n * (q + 1) / k, 1, (int)items.size()});
        }
        seg_off[S] = (int)items.size();
    } else {
        // clusters per segment by largest remainder (every non-empty segment at least one while clusters last)
        std::vector<long long> seg_n((size_t)S), seg_lo((size_t)S);
        std::vector<int> seg_c((size_t)S, 0);
        int nonempty = 0, given = 0;
        for (long long s = 0; s < S; ++s) {
            const long long a = s / G, g = s % G;
            seg_lo[s] = a * E + group_off[g];
            seg_n[s] = group_off[g + 1] - group_off[g];
            if (seg_n[s] > 0) ++nonempty;
        }
        const bool enough = nonempty <= n_clusters;
        for (long long s = 0; s < S && enough; ++s)
            if (seg_n[s] > 0) {
                seg_c[s] = std::max<int>(1, (int)(seg_n[s] * n_clusters / cols));
                given += seg_c[s];
            }
        while (enough && given > n_clusters) {  // (the floor of 1 can overshoot when many segments are tiny)
            long long best = -1;
            for (long long s = 0; s < S; ++s)
                if (seg_c[s] > 1 && (best < 0 || seg_n[s] * seg_c[best] < seg_n[best] * seg_c[s])) best = s;
            if (best < 0) break;
            --seg_c[best];
            --given;
        }
        while (enough && given < n_clusters) {  // the segment with the most columns per cluster takes the next one
            long long best = -1;
            for (long long s = 0; s < S; ++s)
                if (seg_n[s] > 0 && (best < 0 || seg_n[s] * seg_c[best] > seg_n[best] * seg_c[s])) best = s;
            ++seg_c[best];
            ++given;
        }
        if (!enough || given != n_clusters) {
            // more non-empty segments than clusters: this shape keeps the transposed path (re-enter without `direct`)
            return lag_msd_fft_fused(cs, F, E, d_r, scale, max_lag, G, group_off, m, res, out, out_on_device, 0, w12);
        }
        // rows (= Qpart / Ppart rows, consecutive per segment): cluster q, member k -> row 16 q + k
        std::vector<FftItem> rows;
        if (staged) {
            // block 16 q + k = member k of cluster q (no placement assumption); c_lo / c_hi = the cluster's tiles
            for (long long s = 0; s < S; ++s) {
                seg_off[s] = (int)rows.size();
                if (seg_c[s] == 0) continue;
                const long long lo = seg_lo[s], hi = lo + seg_n[s];
                const long long t0 = lo / 16, t1 = (hi + 15) / 16, nt = t1 - t0;
                for (int q = 0; q < seg_c[s]; ++q) {
                    const long long ta = t0 + nt * q / seg_c[s], tb = t0 + nt * (q + 1) / seg_c[s];
                    for (int k = 0; k < 16; ++k) {
                        stages.push_back({lo, hi, k, (int)(rows.size() / 16)});
                        rows.push_back({ta, tb, 1, (int)rows.size()});
                    }
                }
            }
            seg_off[S] = (int)rows.size();
            items = rows;
        } else {
        for (long long s = 0; s < S; ++s) {
            seg_off[s] = (int)rows.size();
            if (seg_c[s] == 0) continue;
            const long long lo = seg_lo[s], hi = lo + seg_n[s];
            const long long t0 = lo / 16, t1 = (hi + 15) / 16, nt = t1 - t0;
            for (int q = 0; q < seg_c[s]; ++q) {
                const long long ta = t0 + nt * q / seg_c[s], tb = t0 + nt * (q + 1) / seg_c[s];
                for (int k = 0; k < 16; ++k) {
                    long long c0 = 16 * ta + k, c1 = 16 * tb;  // columns 16 T + k, ta <= T < tb, inside [lo, hi)
                    while (c0 < lo) c0 += 16;
                    c1 = std::min(c1, hi);
                    rows.push_back({c0, std::max(c0, c1), 16, (int)rows.size()});
                }
            }
        }
        seg_off[S] = (int)rows.size();
        // block b runs on XCD b % 8, dispatch round b / 8: the 16 members of a cluster are the blocks of one XCD in 16
        // consecutive rounds
        const int per_xcd = ctx->cu_count / 8;  // dispatch rounds = blocks per XCD
        items.resize(rows.size());
        for (int b = 0; b < (int)rows.size(); ++b) {
            const int xcd = b % 8, round = b / 8;
            const int q = (round / 16) * 8 + xcd, k = round % 16;
            (void)per_xcd;
            items[(size_t)b] = rows[(size_t)q * 16 + k];
        }
        }
    }
    const long long n_items = (long long)items.size();

    // twiddle table of w_L: A[i] = w^(128 i), B[i] = w^i
    std::vector<double> tab(512);
    const long double step = -2.0L * 3.14159265358979323846264338327950288L / (long double)L;
    for (int i = 0; i < 128; ++i) {
        const long long ia = (128LL * i) % L;
        tab[2 * i] = (double)cosl(step * ia);
        tab[2 * i + 1] = (double)sinl(step * ia);
        tab[256 + 2 * i] = (double)cosl(step * (i % L));
        tab[256 + 2 * i + 1] = (double)sinl(step * (i % L));
    }

    double *d_x = nullptr;  // the transposed, scaled copy [cols][F] (not made for the direct-read kernels)
    if (!direct) {
        d_x = (double *)mdhip_ws(ctx, WS_AUX1, (size_t)cols * F * 8 + 256);
        if (!d_x) return MDHIP_ENOMEM;
    }
    double *d_ring = nullptr;  // SRC == 2: the clusters' staging rings [cluster][ST_BUF][16][16 Fc]
    if (staged) {
        d_ring = (double *)mdhip_ws(ctx, WS_AUX1, (size_t)n_clusters * ST_BUF * 16 * 16 * (size_t)Fc * 8 + 256);
        if (!d_ring) return MDHIP_ENOMEM;
    }
    const size_t qp_b = (size_t)n_items * F * 8, pp_b = (size_t)n_items * (N + 1) * 8;
    MD_WS(d_part, double, WS_PART, qp_b + pp_b);
    double *d_Qpart = d_part, *d_Ppart = d_part + (size_t)n_items * F;
    const size_t q_b = (size_t)S * F * 8, p_b = (size_t)S * (N + 1) * 8, c_b = (size_t)S * n_lags * 8;
    const size_t it_b = (size_t)n_items * sizeof(FftItem), so_b = (((size_t)S + 1) * 4 + 7) / 8 * 8;
    const size_t sg_b = (stages.size() * sizeof(FftStage) + 7) / 8 * 8, ng_b = (size_t)G * 8;
    // ready counters of the rings, a 128-byte line each, and the stall word behind them
    const size_t rd_b = staged ? ((size_t)n_clusters * ST_BUF * ST_FLAG_STRIDE + ST_FLAG_STRIDE) * 4 : 0;
    MD_WS(d_small, unsigned char, WS_AUX3, q_b + p_b + c_b + 4096 + it_b + so_b + sg_b + ng_b + rd_b + 256);
    double *d_Q = reinterpret_cast<double *>(d_small);
    double *d_P = reinterpret_cast<double *>(d_small + q_b);
    double *d_corr = reinterpret_cast<double *>(d_small + q_b + p_b);
    double2 *d_tab = reinterpret_cast<double2 *>(d_small + q_b + p_b + c_b);
    FftItem *d_items = reinterpret_cast<FftItem *>(d_small + q_b + p_b + c_b + 4096);
    int *d_seg_off = reinterpret_cast<int *>(d_small + q_b + p_b + c_b + 4096 + it_b);
    FftStage *d_stages = reinterpret_cast<FftStage *>(d_small + q_b + p_b + c_b + 4096 + it_b + so_b);
    double *d_ng = reinterpret_cast<double *>(d_small + q_b + p_b + c_b + 4096 + it_b + so_b + sg_b);  // entities per group
    // (on a 128-byte boundary: d_small is, and so is everything in front once rounded up)
    const size_t rd_off = (q_b + p_b + c_b + 4096 + it_b + so_b + sg_b + ng_b + 127) / 128 * 128;
    unsigned *d_ready = reinterpret_cast<unsigned *>(d_small + rd_off);
    {
        // twiddles | items | segment offsets | stages: one pinned staging block (the vectors above are locals), one copy
        MD_PIN(h_tab, unsigned char, 4096 + it_b + so_b + sg_b + ng_b);
        memcpy(h_tab, tab.data(), 4096);
        memcpy(h_tab + 4096, items.data(), it_b);
        memcpy(h_tab + 4096 + it_b, seg_off.data(), ((size_t)S + 1) * 4);
        if (!stages.empty()) memcpy(h_tab + 4096 + it_b + so_b, stages.data(), stages.size() * sizeof(FftStage));
        double *h_ng = reinterpret_cast<double *>(h_tab + 4096 + it_b + so_b + sg_b);
        for (long long g = 0; g < G; ++g) h_ng[g] = (double)(group_off[g + 1] - group_off[g]);
        {  // (a kernel on the launch stream, not a copy engine's job: no hand-over between queues — mdhip_copy_small)
            const int rcc = mdhip_copy_small(ctx, d_tab, h_tab, 4096 + it_b + so_b + sg_b + ng_b, hipMemcpyHostToDevice);
            if (rcc) return rcc;
        }
        if (staged) MD_HIP(hipMemsetAsync(d_ready, 0, rd_b, ctx->stream));
    }

    KernelTimer timer(ctx);
    if (!direct) {
        hipLaunchKernelGGL(transpose_scale_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)((F + 63) / 64)), dim3(256),
                           0, ctx->stream, d_r, d_x, F, cols, scale);
        MD_HIP(hipGetLastError());
    }
    const int qr = (int)((F + FT_THREADS - 1) / FT_THREADS);
    if (w12) {
        const int qe = std::max(4, (int)(((F + 1) / 2 + W12_SUB - 1) / W12_SUB));  // first-pass inputs that hold data: 4 .. 6
        const size_t ldsw = w12_lds_bytes(qe);
#define MD_W12_GO(QE, SRC, SH, X, SC)                                                                          \
    {                                                                                                          \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_w12_kernel<QE, SRC, SH>),          \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));                    \
        hipLaunchKernelGGL((msd_power_w12_kernel<QE, SRC, SH>), dim3((unsigned)n_items), dim3(W12_THREADS), ldsw, \
                           ctx->stream, X, (int)F, d_items, d_tab, d_Qpart, d_Ppart, cols, SC, d_stages, d_ring, \
                           d_ready, src_opt == 3 ? -Fc : Fc);                                                  \
    }
#define MD_W12_LAUNCH(QE, SH)                                                                                  \
    {                                                                                                          \
        if (staged) MD_W12_GO(QE, 2, SH, d_r, scale)                                                           \
        else MD_W12_GO(QE, 0, SH, d_x, 1.0)                                                                    \
    }
        if (F < 6 * W12_SUB) MD_W12_LAUNCH(4, true)  // (1536 <= F < 3072: the host's condition, mdhip_lag_msd_fft)
        else if (qe <= 4) MD_W12_LAUNCH(4, false)
        else if (qe == 5) MD_W12_LAUNCH(5, false)
        else MD_W12_LAUNCH(6, false)
#undef MD_W12_LAUNCH
#undef MD_W12_GO
    } else if (v3) {
        const size_t lds3 = f3_lds_bytes(m);
        const long long s0 = N >> 3;
        const int qe = (int)(((F + 1) / 2 + s0 - 1) / s0);  // <= 8
#define MD_F3_GO(JJ, QE, SRC, X, SC)                                                                           \
    {                                                                                                          \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_lds3_kernel<JJ, QE, SRC>),         \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));                    \
        hipLaunchKernelGGL((msd_power_lds3_kernel<JJ, QE, SRC>), dim3((unsigned)n_items), dim3(FT_THREADS), lds3, \
                           ctx->stream, X, (int)F, m, d_items, d_tab, d_Qpart, d_Ppart, cols, SC, d_stages,    \
                           d_ring, d_ready, src_opt == 3 ? -Fc : Fc);                                          \
    }
#define MD_F3_LAUNCH(JJ, QE)                                                                                   \
    {                                                                                                          \
        if (staged) MD_F3_GO(JJ, QE, 2, d_r, scale)                                                            \
        else if (direct) MD_F3_GO(JJ, QE, 1, d_r, scale)                                                       \
        else MD_F3_GO(JJ, QE, 0, d_x, 1.0)                                                                     \
    }
// (eight staging units per lane and tile: rows per member beyond 320, F > 5120)
#define MD_F3_GO8(QE)                                                                                          \
    {                                                                                                          \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_lds3_kernel<2, QE, 2, 8>),         \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));                    \
        hipLaunchKernelGGL((msd_power_lds3_kernel<2, QE, 2, 8>), dim3((unsigned)n_items), dim3(FT_THREADS), lds3, \
                           ctx->stream, d_r, (int)F, m, d_items, d_tab, d_Qpart, d_Ppart, cols, scale, d_stages, \
                           d_ring, d_ready, src_opt == 3 ? -Fc : Fc);                                          \
    }
        if (staged && Fc > 64 * 5) {
            if (qe <= 3) MD_F3_GO8(3)
            else MD_F3_GO8(4)
        } else if (s0 > FT_THREADS) {
            if (qe <= 3) MD_F3_LAUNCH(2, 3)
            else if (qe <= 4) MD_F3_LAUNCH(2, 4)
            else MD_F3_LAUNCH(2, 8)
        } else {
            if (qe <= 3) MD_F3_LAUNCH(1, 3)
            else if (qe <= 4) MD_F3_LAUNCH(1, 4)
            else MD_F3_LAUNCH(1, 8)
        }
#undef MD_F3_LAUNCH
#undef MD_F3_GO8
#undef MD_F3_GO
    } else if (v2) {
        const size_t lds2 = f2_lds_bytes(m);
        const int qr2 = (int)(((F + 1) / 2 + FT_THREADS - 1) / FT_THREADS);  // sample pairs per lane, <= N / 512
#define MD_F2_LAUNCH(QR2)                                                                                      \
    {                                                                                                          \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_lds2_kernel<QR2>),                 \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));                    \
        hipLaunchKernelGGL((msd_power_lds2_kernel<QR2>), dim3((unsigned)n_items), dim3(FT_THREADS), lds2,      \
                           ctx->stream, d_x, (int)F, m, d_items, d_tab, d_Qpart, d_Ppart);                     \
    }
        // QR2 = sample pairs per lane: ceil(ceil(F / 2) / 512) <= N / 512 = 16
        if (qr2 <= 1) MD_F2_LAUNCH(1)
        else if (qr2 <= 2) MD_F2_LAUNCH(2)
        else if (qr2 <= 3) MD_F2_LAUNCH(3)
        else if (qr2 <= 5) MD_F2_LAUNCH(5)
        else if (qr2 <= 8) MD_F2_LAUNCH(8)
        else MD_F2_LAUNCH(16)  // (F > N: only with max_lag < F - 1)
#undef MD_F2_LAUNCH
    } else {
#define MD_FT_CASE(QR)                                                                                         \
    {                                                                                                          \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_lds_kernel<QR>),                   \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));                   \
        hipLaunchKernelGGL(msd_power_lds_kernel<QR>, dim3((unsigned)n_items), dim3(FT_THREADS), lds_b,         \
                           ctx->stream, d_x, (int)F, m, d_items, d_tab, d_Qpart, d_Ppart);                     \
    }
    if (qr <= 1) MD_FT_CASE(1)
    else if (qr <= 2) MD_FT_CASE(2)
    else if (qr <= 4) MD_FT_CASE(4)
    else if (qr <= 8) MD_FT_CASE(8)
    else if (qr <= 10) MD_FT_CASE(10)
    else if (qr <= 12) MD_FT_CASE(12)
    else if (qr <= 16) MD_FT_CASE(16)
    else if (qr <= 24) MD_FT_CASE(24)
    else MD_FT_CASE(32)
#undef MD_FT_CASE
    }
    MD_HIP(hipGetLastError());
    hipLaunchKernelGGL(fold_items_kernel, dim3((unsigned)((F + 255) / 256), (unsigned)S), dim3(256), 0, ctx->stream,
                       d_Qpart, d_seg_off, F, d_Q);
    hipLaunchKernelGGL(fold_items_kernel, dim3((unsigned)((N + 1 + 255) / 256), (unsigned)S), dim3(256), 0,
                       ctx->stream, d_Ppart, d_seg_off, N + 1, d_P);
    if (w12) {
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_inverse_w12_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
        hipLaunchKernelGGL(msd_inverse_w12_kernel, dim3((unsigned)S), dim3(512), lds_b, ctx->stream, d_P, d_tab,
                           (int)n_lags, d_corr);
    } else {
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_inverse_lds_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
        hipLaunchKernelGGL(msd_inverse_lds_kernel, dim3((unsigned)S), dim3(FT_THREADS), lds_b, ctx->stream, d_P, m, d_tab,
                           (int)n_lags, d_corr);
    }
    MD_HIP(hipGetLastError());
    timer.stop();
    ctx->last_kernel = w12 ? "msd_power_w12_kernel" : "msd_power_lds_kernel";

    // the finish, on the device (lag_finish_dd_kernel): means into `d_fin`, from there to the caller's buffer on the
    // stream; only the bound of every group (and the ring's stall word) comes back for the completion step
    const size_t fin_b = (size_t)n_lags * G * 4 * 8;
    MD_WS(d_fin_ws, unsigned char, WS_OUT3, fin_b + lag_bound_words(S) * 8 + (size_t)S * (F + 1) * sizeof(DD) + 64);
    double *d_fin = reinterpret_cast<double *>(d_fin_ws), *d_bound = d_fin + (size_t)n_lags * G * 4;
    DD *d_pre = reinterpret_cast<DD *>(d_bound + lag_bound_words(S));
    const double eps_l = 4.0 * 2.220446049250313e-16 * std::log2((double)L);
    hipLaunchKernelGGL(lag_finish_dd_kernel, dim3((unsigned)S), dim3(256), 0, ctx->stream, d_Q, d_corr, n_lags, 1.0, F, n_lags,
                       (int)G, d_ng, eps_l, d_pre, d_fin, d_bound);
    hipLaunchKernelGGL(lag_total_kernel, dim3((unsigned)((n_lags * G + 255) / 256)), dim3(256), 0, ctx->stream, d_fin,
                       n_lags * G, d_bound, (int)S,
                       staged ? d_ready + (size_t)n_clusters * ST_BUF * ST_FLAG_STRIDE : (const unsigned *)nullptr);
    ctx->lag_status_dev = d_bound + S;  // (mdhip_lag_msd_status_dev: valid until the next call that uses WS_OUT3)
    MD_HIP(hipGetLastError());
    {
        const int rcr = mdhip_result(cs, out, d_fin, fin_b, out_on_device);
        if (rcr) return rcr;
    }
    MD_PIN(h_bound, double, lag_bound_words(S) * 8 + 8);
    unsigned *h_stall = reinterpret_cast<unsigned *>(h_bound + lag_bound_words(S));
    *h_stall = 0u;
    {
        const int rcc = mdhip_copy_small(ctx, h_bound, d_bound, lag_bound_words(S) * 8, hipMemcpyDeviceToHost);
        if (rcc) return rcc;
    }
    if (staged) {
        const int rcc = mdhip_copy_small(ctx, h_stall, d_ready + (size_t)n_clusters * ST_BUF * ST_FLAG_STRIDE, 4,
                                         hipMemcpyDeviceToHost);
        if (rcc) return rcc;
    }
    cs.defer([=]() {
        timer.collect();
        if (*h_stall) {
            // a cluster member never ran (the grid was not resident as a whole): the same call over the transposed copy,
            // inside a synchronous call of its own (d_r is the caller's, or the call's staging: valid until completion);
            // it writes `out` again
            ++ctx->cur_fallbacks;  // visible: mdhip_ticket_status / mdhip_fallbacks (the 2 s poll is otherwise silent)
            ++ctx->fallbacks_total;
            CallScope again(ctx);
            const int rc2 = lag_msd_fft_fused(again, F, E, d_r, scale, max_lag, G, res->group_off.data(), m, res, out,
                                              out_on_device, 0, w12);
            if (rc2 != MDHIP_OK) return rc2;
            const int rc3 = again.end();
            ctx->last_kernel = w12 ? "msd_power_w12_kernel (repeated over the transposed copy: a cluster member did not run)"
                                   : "msd_power_lds_kernel (repeated over the transposed copy: a cluster member did not run)";
            return rc3;
        }
        lag_collect_bounds(res.get(), h_bound, S, n_lags);
        return MDHIP_OK;
    });
    return MDHIP_OK;
}

// Round 6: series beyond the fused kernels through msd_power_w12r_kernel (msd_fft_w12r.h): padded length L' = 4 x 6144 = 24 576
// >= F + max_lag, F <= 12 288. The trajectory is transposed and centred batch by batch (transpose_centre64_kernel, as the
// batched path), every batch's series are dealt to one block per CU, the blocks' partial spectra are folded per segment, and
// the correlations come from msd_residue_inverse_kernel; the finish is the fused kernels' (lag_finish_dd_kernel).
// frames between two samples of the series' means (col_sum_sample_kernel): ~512 samples of a long trajectory (128 of one of at
// most 1536 frames); option
// `lag_mean_sample` 0 = every frame, n > 0 = about n samples
inline long long lag_mean_stride(const mdhip_ctx *ctx, long long F, long long dflt = 512)
{
    if (ctx->opt_lag_mean_sample == 0) return 1;
    const long long want = ctx->opt_lag_mean_sample > 0 ? ctx->opt_lag_mean_sample : dflt;
    return std::max<long long>(1, F / want);
}

// short_d2 > 0: trajectories BELOW msd_power_w12_kernel's range (F + max_lag <= 3072, F <= 1536) through msd_power_w1_kernel<short_d2>
// — padded length short_d2 x 1024, one wave per series; the same preparation, folds, inverse and finish.
int lag_msd_fft_residue(CallScope &cs, long long F, long long E, const double *d_r, double scale, int max_lag, long long G,
                        const int64_t *group_off, const std::shared_ptr<LagFftResult> &res, double *out, int out_on_device,
                        int short_d2 = 0)
{
    mdhip_ctx *ctx = cs.ctx;
    // D = 4: padded length 24 576, the series as they are; D = 8: 49 152, the series folded once by the transposition (rows
    // [g | h] of 24 576 doubles): the even frequencies by the D = 4 kernel over those rows, the odd ones by msd_power_w12o_kernel
    const int D = short_d2 ? 4 : (F <= 2LL * W12_N && F + max_lag <= 4LL * W12_N) ? 4 : 8;  // (short: as D = 4 in what follows)
    const long long LP = short_d2 ? 1024LL * short_d2 : (long long)D * W12_N, K = LP / 2 + 1;
    const long long LP4 = short_d2 ? LP : 4LL * W12_N;  // the length of the first twiddle table (msd_power_w12p_kernel's)
    const int rows_per_item = 1;                        // partial spectra a block writes
    const long long n_lags = (long long)max_lag + 1, cols = 3 * E, S = 3 * G;
    const long long row_len = D == 4 ? F : 4LL * W12_N;  // doubles per series of the time-major copy
    res->delivered = true;
    // Round 6, `lag_overlap` (an experiment that did not pay; off by default, kept behind its option and its test): the
    // transposition of batch k + 1 on a quarter of the CUs WHILE the transform kernel of batch k (compute-bound, one workgroup
    // per CU) runs on the other three quarters — two CU-masked streams (hipExtStreamCreateWithCUMask), two buffers, at least
    // six batches, the first transposed and the last transformed on the whole chip. The masks work as advertised; what does
    // not is the premise that a streaming kernel needs few CUs: the transposition moves 4.8 TB/s on 256 CUs and 1.7 TB/s on
    // 64 (a CU cannot hold the ~160 KB in flight that a quarter of the chip would need to cover HBM's latency), so the call
    // takes 16.7 ms instead of 13.2 (profiles/r06_ab_lag_overlap.txt).
    const bool want_overlap = ctx->opt_lag_overlap != 0 && !short_d2 &&
                              (cols * row_len * 8 >= (512LL << 20) || ctx->opt_lag_overlap >= 2 /* tests: whatever the size */) &&
                              mdhip_part_streams(ctx);
    long long nb_max = std::max<long long>(1, ((long long)ctx->opt_lag_batch_mb << 20) / (row_len * 8) / (want_overlap ? 2 : 1));
    if (want_overlap) nb_max = std::min(nb_max, std::max<long long>(ctx->opt_lag_overlap >= 2 ? 1 : 4LL * ctx->cu_count, (cols + 5) / 6));
    const long long n_batches = (cols + nb_max - 1) / nb_max;
    const long long nb0 = (cols + n_batches - 1) / n_batches;
    const bool overlap = want_overlap && n_batches >= 3;
    // (CUs the transform kernel of batch b runs on: it is given one workgroup per CU)
    auto batch_cus = [&](long long b) { return overlap && b + 1 < n_batches ? ctx->part_cus[0] : ctx->cu_count; };
    // work items, batch by batch: every (segment, batch) overlap gets its share of ~one block per CU
    std::vector<FftItem> items;
    struct Fold {
        long long batch, seg, c_lo, c_n;  // the segment's columns inside the batch
        int first, count;                 // its rows of the blocks' partial spectra
    };
    std::vector<Fold> folds;
    std::vector<int> batch_off((size_t)n_batches + 1, 0);
    int max_items = 0;
    long long max_tiles = 1;
    for (long long b = 0; b < n_batches; ++b) {
        const long long c_first = b * nb0, nb = std::min(nb0, cols - c_first);
        batch_off[(size_t)b] = (int)items.size();
        int row = 0;
        for (long long s = 0; s < S; ++s) {
            const long long a = s / G, g = s % G;
            const long long lo = std::max(c_first, a * E + (long long)group_off[g]);
            const long long hi = std::min(c_first + nb, a * E + (long long)group_off[g + 1]);
            if (lo >= hi) continue;
            const long long n = hi - lo;
            long long k = (n * batch_cus(b) + nb / 2) / nb;
            k = std::max<long long>(1, std::min(k, n));
            folds.push_back({b, s, lo, n, row, (int)k});
            max_tiles = std::max(max_tiles, (n + 64 * TSQ_TILES - 1) / (64 * TSQ_TILES));
            for (long long q = 0; q < k; ++q)
                items.push_back({lo - c_first + n * q / k, lo - c_first + n * (q + 1) / k, 1, row++});
        }
        max_items = std::max(max_items, row);
    }
    batch_off[(size_t)n_batches] = (int)items.size();

    // twiddle tables: B[i] = w^i (i < 256), A[i] = w^(256 i) of w_24576 (msd_power_w12p_kernel / _w12r_), behind it of w_49152
    const int n_tab4 = 256 + (int)(LP4 / 256), n_tab8 = D == 8 ? 256 + (int)(LP / 256) : 0, n_tab = n_tab4 + n_tab8;
    std::vector<double> tab((size_t)2 * n_tab);
    for (int part = 0; part < (D == 8 ? 2 : 1); ++part) {
        const long long len = part == 0 ? LP4 : LP;
        const long double step = -2.0L * 3.14159265358979323846264338327950288L / (long double)len;
        const int base = part == 0 ? 0 : n_tab4, cnt = part == 0 ? n_tab4 : n_tab8;
        for (int i = 0; i < cnt; ++i) {
            const long long idx = i < 256 ? i : 256LL * (i - 256);
            tab[2 * (base + i)] = (double)cosl(step * idx);
            tab[2 * (base + i) + 1] = (double)sinl(step * idx);
        }
    }

    MD_WS(d_mean, double, WS_AUX0, (size_t)(MF_SLABS + 1) * cols * 8);
    double *d_msum = d_mean + cols;
    MD_WS(d_pad0, double, WS_AUX1, (size_t)nb0 * row_len * 8 * (overlap ? 2 : 1) + 512);
    double *d_pads[2] = {d_pad0, overlap ? d_pad0 + (((size_t)nb0 * row_len + 31) & ~(size_t)31) : d_pad0};
    MD_WS(d_part, double, WS_PART, (size_t)max_items * rows_per_item * K * 8);
    MD_WS(d_qpart, double, WS_AUX2, (size_t)max_tiles * F * 8);
    // (Q | P | correlations, the three together rounded up to 16 bytes: the twiddle table behind them is read as double2)
    const size_t q_b = (size_t)S * F * 8, p_b = (size_t)S * K * 8, c_b = ((size_t)S * (F + K + n_lags) * 8 + 15) / 16 * 16 - q_b - p_b;
    const size_t tab_b = (size_t)n_tab * 16, it_b = (items.size() * sizeof(FftItem) + 15) / 16 * 16;
    const size_t go_b = (size_t)(G + 1) * 8, ng_b = (size_t)G * 8;
    MD_WS(d_small, unsigned char, WS_AUX3, q_b + p_b + c_b + tab_b + it_b + go_b + ng_b + 256);
    double *d_Q = reinterpret_cast<double *>(d_small);
    double *d_P = reinterpret_cast<double *>(d_small + q_b);
    double *d_corr = reinterpret_cast<double *>(d_small + q_b + p_b);
    double2 *d_tab = reinterpret_cast<double2 *>(d_small + q_b + p_b + c_b);
    FftItem *d_items = reinterpret_cast<FftItem *>(d_small + q_b + p_b + c_b + tab_b);
    long long *d_goff = reinterpret_cast<long long *>(d_small + q_b + p_b + c_b + tab_b + it_b);
    double *d_ng = reinterpret_cast<double *>(d_small + q_b + p_b + c_b + tab_b + it_b + go_b);
    (void)d_goff;
    {
        MD_PIN(h_blk, unsigned char, tab_b + it_b + go_b + ng_b);
        memcpy(h_blk, tab.data(), tab_b);
        memcpy(h_blk + tab_b, items.data(), items.size() * sizeof(FftItem));
        memcpy(h_blk + tab_b + it_b, group_off, go_b);
        double *h_ng = reinterpret_cast<double *>(h_blk + tab_b + it_b + go_b);
        for (long long g = 0; g < G; ++g) h_ng[g] = (double)(group_off[g + 1] - group_off[g]);
        const int rcc = mdhip_copy_small(ctx, d_tab, h_blk, tab_b + it_b + go_b + ng_b, hipMemcpyHostToDevice);
        if (rcc) return rcc;
    }
    MD_HIP(hipMemsetAsync(d_Q, 0, q_b + p_b, ctx->stream));  // (Q | P: both are added to, batch by batch)

    KernelTimer timer(ctx);
    const long long m_stride = lag_mean_stride(ctx, F, short_d2 ? 128 : 512);
    if (m_stride > 1)
        hipLaunchKernelGGL(col_sum_sample_kernel, dim3((unsigned)((cols + 255) / 256), MF_SLABS), dim3(256), 0, ctx->stream, d_r, F,
                           cols, m_stride, d_msum);
    else
        hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)((cols + 255) / 256), MF_SLABS), dim3(256), 0, ctx->stream, d_r, F, cols,
                           d_msum);
    hipLaunchKernelGGL(col_mean_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, d_msum, MF_SLABS,
                       m_stride > 1 ? sample_count(F, MF_SLABS, m_stride) : F, cols, scale, d_mean);
    MD_HIP(hipGetLastError());
    // lag_residue 1 (default): two transforms per series (the even frequencies packed, the odd ones as class 1); 2: three
    // classes (0, 1, 2), nothing packed — the first form of the kernel, kept for A/B
    const bool packed = ctx->opt_lag_residue != 2 || D == 8;
    const size_t ldsr = short_d2 ? w1_lds_bytes(short_d2) : packed ? w12p_lds_bytes() : w12r_lds_bytes(4);
    MD_HIP(hipFuncSetAttribute(short_d2 == 1   ? reinterpret_cast<const void *>(msd_power_w1_kernel<1>)
                               : short_d2 == 2 ? reinterpret_cast<const void *>(msd_power_w1_kernel<2>)
                               : short_d2 == 3 ? reinterpret_cast<const void *>(msd_power_w1_kernel<3>)
                               : !packed       ? reinterpret_cast<const void *>(msd_power_w12r_kernel<4>)
                               : D == 4        ? reinterpret_cast<const void *>(msd_power_w12p_kernel<true>)
                                               : reinterpret_cast<const void *>(msd_power_w12p_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr));
    if (D == 8) {
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_w12o_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)w12o_lds_bytes()));
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_power_w12o_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)w12o_lds_bytes()));
    }
    // first fold record of every batch
    std::vector<size_t> fold_first((size_t)n_batches + 1, folds.size());
    for (size_t fi = folds.size(); fi-- > 0;) fold_first[(size_t)folds[fi].batch] = fi;
    for (long long b = n_batches - 1; b >= 0; --b)
        if (fold_first[(size_t)b] == folds.size()) fold_first[(size_t)b] = fold_first[(size_t)b + 1];
    // the batch's series into `pad`, segment by segment (a tile's per-frame squares belong to one segment), S1's terms with them
    auto transpose_batch = [&](long long b, hipStream_t st, double *pad) {
        const long long c_first = b * nb0;
        for (size_t fi = fold_first[(size_t)b]; fi < folds.size() && folds[fi].batch == b; ++fi) {
            const long long lo = folds[fi].c_lo, n = folds[fi].c_n, tiles = (n + 64 * TSQ_TILES - 1) / (64 * TSQ_TILES);
            if (D == 4)
                hipLaunchKernelGGL(transpose_centre64_sq_kernel, dim3((unsigned)tiles, (unsigned)((F + 63) / 64)), dim3(256), 0, st,
                                   d_r, d_mean, F, cols, lo, n, lo - c_first, scale, pad, d_qpart);
            else
                hipLaunchKernelGGL(transpose_fold64_sq_kernel, dim3((unsigned)tiles, (unsigned)(2 * W12_N / 64)), dim3(256), 0, st,
                                   d_r, d_mean, F, cols, lo, n, lo - c_first, scale, pad, d_qpart);
            hipLaunchKernelGGL(power_fold_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, d_qpart, (int)tiles, F,
                               d_Q + (size_t)folds[fi].seg * F);
        }
    };
    // the batch's power spectra from `pad`, folded into the segments' sums
    auto power_batch = [&](long long b, hipStream_t st, const double *pad) {
        const int n_it = batch_off[(size_t)b + 1] - batch_off[(size_t)b];
        const FftItem *its = d_items + batch_off[(size_t)b];
        if (short_d2 == 1)
            hipLaunchKernelGGL(msd_power_w1_kernel<1>, dim3((unsigned)n_it), dim3(W12_THREADS), ldsr, st, pad, (int)F, its, d_tab, d_part);
        else if (short_d2 == 2)
            hipLaunchKernelGGL(msd_power_w1_kernel<2>, dim3((unsigned)n_it), dim3(W12_THREADS), ldsr, st, pad, (int)F, its, d_tab, d_part);
        else if (short_d2 == 3)
            hipLaunchKernelGGL(msd_power_w1_kernel<3>, dim3((unsigned)n_it), dim3(W12_THREADS), ldsr, st, pad, (int)F, its, d_tab, d_part);
        else if (D == 8) {
            hipLaunchKernelGGL(msd_power_w12p_kernel<false>, dim3((unsigned)n_it), dim3(W12_THREADS), ldsr, st, pad, row_len, 2 * W12_N,
                               2 * W12_N, 2 * W12_N, 2, (int)K, its, d_tab, d_part);
            hipLaunchKernelGGL(msd_power_w12o_kernel<1>, dim3((unsigned)n_it), dim3(W12_THREADS), w12o_lds_bytes(), st, pad, its,
                               d_tab + n_tab4, d_part);
            hipLaunchKernelGGL(msd_power_w12o_kernel<3>, dim3((unsigned)n_it), dim3(W12_THREADS), w12o_lds_bytes(), st, pad, its,
                               d_tab + n_tab4, d_part);
        } else if (packed)
            hipLaunchKernelGGL(msd_power_w12p_kernel<true>, dim3((unsigned)n_it), dim3(W12_THREADS), ldsr, st, pad, row_len, (int)F, 0,
                               (int)F, 1, (int)K, its, d_tab, d_part);
        else
            hipLaunchKernelGGL((msd_power_w12r_kernel<4>), dim3((unsigned)n_it), dim3(W12_THREADS), ldsr, st, pad, (int)F, its, d_tab,
                               d_part);
        for (size_t fi = fold_first[(size_t)b]; fi < folds.size() && folds[fi].batch == b; ++fi)
            hipLaunchKernelGGL(power_fold_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st,
                               d_part + (size_t)folds[fi].first * rows_per_item * K, folds[fi].count * rows_per_item, K,
                               d_P + (size_t)folds[fi].seg * K);
    };
    if (!overlap) {
        for (long long b = 0; b < n_batches; ++b) {
            transpose_batch(b, ctx->stream, d_pads[0]);
            power_batch(b, ctx->stream, d_pads[0]);
            MD_HIP(hipGetLastError());
        }
    } else {
        // batch 0 is transposed on the whole chip; then stream A (3/4 of the CUs) transforms batch b while stream B (1/4)
        // transposes batch b + 1 into the other buffer; the last batch is transformed on the whole chip again
        hipStream_t sa = ctx->part_stream[0], sb = ctx->part_stream[1];
        hipEvent_t ev_fork = ctx->part_ev[0], *ev_ready = ctx->part_ev + 1, *ev_free = ctx->part_ev + 3, ev_join = ctx->part_ev[5];
        transpose_batch(0, ctx->stream, d_pads[0]);
        MD_HIP(hipEventRecord(ev_fork, ctx->stream));
        MD_HIP(hipStreamWaitEvent(sa, ev_fork, 0));
        MD_HIP(hipStreamWaitEvent(sb, ev_fork, 0));
        for (long long b = 0; b + 1 < n_batches; ++b) {
            const int cur = (int)(b & 1), nxt = cur ^ 1;
            if (b >= 1) MD_HIP(hipStreamWaitEvent(sb, ev_free[nxt], 0));  // (batch b - 1 has been transformed out of that buffer)
            transpose_batch(b + 1, sb, d_pads[nxt]);
            MD_HIP(hipEventRecord(ev_ready[nxt], sb));
            if (b >= 1) MD_HIP(hipStreamWaitEvent(sa, ev_ready[cur], 0));
            power_batch(b, sa, d_pads[cur]);
            MD_HIP(hipEventRecord(ev_free[cur], sa));
            MD_HIP(hipGetLastError());
        }
        const int last = (int)((n_batches - 1) & 1);
        MD_HIP(hipEventRecord(ev_join, sa));
        MD_HIP(hipStreamWaitEvent(ctx->stream, ev_join, 0));
        MD_HIP(hipStreamWaitEvent(ctx->stream, ev_ready[last], 0));
        power_batch(n_batches - 1, ctx->stream, d_pads[last]);
        MD_HIP(hipGetLastError());
    }
    const size_t ldsi = (size_t)(LP / 4 + 1 + RI_WAVES * 64 * 2) * 8;
    MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(msd_residue_inverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)ldsi));
    hipLaunchKernelGGL(msd_residue_inverse_kernel, dim3((unsigned)((n_lags + 63) / 64), (unsigned)S), dim3(64 * RI_WAVES), ldsi, ctx->stream,
                       d_P, (int)LP, (int)n_lags, d_corr);
    MD_HIP(hipGetLastError());
    timer.stop();
    ctx->last_kernel = short_d2 ? "msd_power_w1_kernel"
                       : D == 8 ? "msd_power_w12p_kernel + msd_power_w12o_kernel"
                       : packed ? "msd_power_w12p_kernel"
                                : "msd_power_w12r_kernel";

    // the finish, on the device, as the fused kernels'
    const size_t fin_b = (size_t)n_lags * G * 4 * 8;
    MD_WS(d_fin_ws, unsigned char, WS_OUT3, fin_b + lag_bound_words(S) * 8 + (size_t)S * (F + 1) * sizeof(DD) + 64);
    double *d_fin = reinterpret_cast<double *>(d_fin_ws), *d_bound = d_fin + (size_t)n_lags * G * 4;
    DD *d_pre = reinterpret_cast<DD *>(d_bound + lag_bound_words(S));
    const double eps_l = 4.0 * 2.220446049250313e-16 * std::log2((double)LP);
    hipLaunchKernelGGL(lag_finish_dd_kernel, dim3((unsigned)S), dim3(256), 0, ctx->stream, d_Q, d_corr, n_lags, 1.0, F, n_lags,
                       (int)G, d_ng, eps_l, d_pre, d_fin, d_bound);
    hipLaunchKernelGGL(lag_total_kernel, dim3((unsigned)((n_lags * G + 255) / 256)), dim3(256), 0, ctx->stream, d_fin, n_lags * G,
                       d_bound, (int)S, (const unsigned *)nullptr);
    ctx->lag_status_dev = d_bound + S;
    MD_HIP(hipGetLastError());
    {
        const int rcr = mdhip_result(cs, out, d_fin, fin_b, out_on_device);
        if (rcr) return rcr;
    }
    MD_PIN(h_bound, double, lag_bound_words(S) * 8);
    {
        const int rcc = mdhip_copy_small(ctx, h_bound, d_bound, lag_bound_words(S) * 8, hipMemcpyDeviceToHost);
        if (rcc) return rcc;
    }
    cs.defer([timer, res, h_bound, S, n_lags]() {
        timer.collect();
        lag_collect_bounds(res.get(), h_bound, S, n_lags);
        return MDHIP_OK;
    });
    return MDHIP_OK;
}

}  // namespace

// d_r: device [F][3][E]. out: host [max_lag+1][G][4] means as mdhip_lag_msd. *rel_bound: the largest
// estimated relative rounding error over all (lag >= 1, group, axis) entries with a non-zero value.
// Both paths — the fused kernels (padded length <= 16384) and the batched transforms below — finish on the device and
// deliver the means themselves (res->delivered).
int mdhip_lag_msd_fft(CallScope &cs, int64_t n_frames, int64_t n_ent, const double *d_r, double scale,
                      int max_lag, int n_groups, const int64_t *group_off, const std::shared_ptr<LagFftResult> &res,
                      double *out, int out_on_device)
{
    mdhip_ctx *ctx = cs.ctx;
    const long long F = n_frames, E = n_ent, G = n_groups;
    res->group_off.assign(group_off, group_off + n_groups + 1);
    res->out.clear();
    res->bound = 0.0;
    const long long n_lags = (long long)max_lag + 1;
    const long long cols = 3 * E;
    // round 6: trajectories below msd_power_w12_kernel's range, one wave per series (msd_fft_w12r.h): padded length 1024 / 2048 / 3072
    // (the residue-class host path launches its transposition and folds per (axis, group) segment: with many groups the
    // block-wide kernels, which take every segment in one launch, stay the faster choice for calls of a millisecond)
    if (ctx->opt_lag_variant != 4 && ctx->opt_lag_w1 != 0 && F + max_lag <= 3072 && F <= 1536 && F >= 2 && n_groups <= 16 &&
        (F < std::max(3 * W12_SUB, ctx->opt_lag_w12_min_f) || ctx->opt_lag_w12_min_f <= 0 || F + max_lag <= 2048) &&
        w1_lds_bytes(3) <= ctx->lds_max)
        return lag_msd_fft_residue(cs, F, E, d_r, scale, max_lag, G, group_off, res, out, out_on_device,
                                   (int)((F + max_lag + 1023) / 1024));
    if (ctx->opt_lag_variant != 4) {
        // fused LDS path when the padded series fits: L = power of two >= max(16, F + max_lag)
        int m = 3;
        while ((2LL << m) < F + max_lag) ++m;
        // round 5: padded length 12288 = 3 * 2^12 where 16384 would be the next power of two (msd_fft_w12.h)
        // round 6: the same kernel for 2048 < F + max_lag <= 8192 (m == 11, 12) from `lag_w12_min_f` frames on (default 1536,
        // the SHORT instance's lower limit): its twelve register-resident 512-point sub-transforms cost less per series than
        // the power-of-two kernels' 2048- and 4096-point transforms through LDS although it transforms 1.5-3 x the points
        // (tools/lag_sizes.py, profiles/r06_lag_sizes_ab.txt: E = 50 000, full lag, F = 1536 3.77 vs 3.78 ms, 2048 3.62 vs
        // 4.06, 3000 3.79 vs 5.43, 4096 4.02 vs 7.01)
        const bool w12_long = m == 13 && F >= 6 * W12_SUB;
        const bool w12_short = (m == 12 || m == 11) && F >= std::max(3 * W12_SUB, ctx->opt_lag_w12_min_f) && ctx->opt_lag_w12_min_f > 0;
        if (ctx->opt_lag_fft_kernel >= 3 && (w12_long || w12_short) && F + max_lag <= 2 * W12_N && (F + 1) / 2 <= 6 * W12_SUB &&
            w12_lds_bytes(6) <= ctx->lds_max) {
            res->delivered = true;
            return lag_msd_fft_fused(cs, F, E, d_r, scale, max_lag, G, group_off, m, res, out, out_on_device, -1, true);
        }
        if (m <= FT_MAX_M && ft_lds_bytes(m) <= ctx->lds_max) {
            res->delivered = true;
            return lag_msd_fft_fused(cs, F, E, d_r, scale, max_lag, G, group_off, m, res, out, out_on_device);
        }
    }
    // round 6: 16 384 < F + max_lag <= 24 576 (F <= 12 288) in residue classes of a 4 x 6144-point transform, F + max_lag <=
    // 49 152 (F <= 24 576) of an 8 x 6144-point one: no transform pass through HBM
    if (ctx->opt_lag_variant != 4 && ctx->opt_lag_residue != 0 && F + max_lag <= 8LL * W12_N && F <= 4LL * W12_N &&
        std::max(std::max(w12r_lds_bytes(4), w12p_lds_bytes()), w12o_lds_bytes()) <= ctx->lds_max)
        return lag_msd_fft_residue(cs, F, E, d_r, scale, max_lag, G, group_off, res, out, out_on_device);
    const long long L = pow2_length(F + max_lag);
    MD_REQUIRE(L < (1LL << 30), "series too long for the FFT path (%lld)", L);
    // (round 6: this path finishes on the device as well — lag_finish_dd_kernel on Q and the correlations where they are —
    // so that series of more than 16 384 padded points, trajectories of 10^4+ frames, no longer cost two device-to-host
    // copies and a host pass in long double, and carry the same device status word as the fused kernels)
    res->delivered = true;
    const long long K = L / 2 + 1;
    const long long S = 3 * G;  // (axis, group) segments

    // Round 6 (`lag_batched_fuse`, default): the first transform pass reads the centred series [nb][F] where the transposition
    // left them (implicit zero padding: no padded copy is written or read) and the column sums of |X_k|^2 are taken straight
    // from the packed transform (r2c_power_rows_kernel: the half spectra are never written) — 2.1 -> 1.3 MB of HBM traffic per
    // series at F = 10 000 (profiles/r06_lag_long_kernel_stats.csv before, r06_lag_sizes.txt after). 0: the round-2 sequence.
    // Second step (`lag_batched_fuse` 2, default): the transform in TWO passes, the second one fused with the column sums
    // (fft_power_pass_kernel): the packed transform is never written either — 1.3 -> 0.9 MB per series (one transform buffer).
    const bool fuse = ctx->opt_lag_batched_fuse != 0;
    const bool fuse2 = ctx->opt_lag_batched_fuse >= 2 && mdhip_fft_power2_plan(ctx, L);
    constexpr int MF_SPLITS2 = 128;  // row splits of the fused pass (8-16 tiles of columns each: >= 4 workgroups per CU)
    // batches of whole series: (padded copy | centred series) + the transform buffers (+ spectrum) <= ~4 GiB
    const long long per_series = fuse2 ? F * 8 + L * 8 : fuse ? F * 8 + 2 * L * 8 : 2 * L * 8 + K * 16;
    const long long nb_max = std::max<long long>(1, std::min<long long>(((long long)ctx->opt_lag_batch_mb << 20) / per_series, (1LL << 31) / K));
    const long long n_batches = (cols + nb_max - 1) / nb_max;
    const long long nb0 = (cols + n_batches - 1) / n_batches;

    MD_WS(d_mean, double, WS_AUX0, (size_t)(MF_SLABS + 1) * cols * 8);
    double *d_msum = d_mean + cols;
    MD_WS(d_pad, double, WS_AUX1, (size_t)nb0 * (fuse ? F : L) * 8);                 // fuse: the centred series [nb][F]
    MD_WS(d_spec, double2, WS_AUX2, fuse ? (size_t)nb0 * L * 8 : (size_t)nb0 * K * 16);  // fuse: the first transform buffer
    MD_WS(d_tmp, double2, WS_FFT_TMP, (size_t)(fuse2 ? S : std::max(nb0, S)) * L * 8 + 64);
    // Q [S][F] | P [S][K] | complex P [S][K] | correlations [S][L] | group offsets
    const size_t q_b = (size_t)S * F * 8, p_b = (size_t)S * K * 8, z_b = (size_t)S * K * 16, c_b = (size_t)S * L * 8;
    MD_WS(d_small, unsigned char, WS_AUX3, q_b + p_b + z_b + c_b + (size_t)(G + 1) * 8 + (size_t)G * 8 + 256);
    double *d_Q = reinterpret_cast<double *>(d_small);
    double *d_P = reinterpret_cast<double *>(d_small + q_b);
    double2 *d_Z = reinterpret_cast<double2 *>(d_small + q_b + p_b);
    double *d_corr = reinterpret_cast<double *>(d_small + q_b + p_b + z_b);
    long long *d_goff = reinterpret_cast<long long *>(d_small + q_b + p_b + z_b + c_b);
    double *d_ng = reinterpret_cast<double *>(d_goff + G + 1);  // entities per group (lag_finish_dd_kernel)
    MD_WS(d_part, double, WS_PART, (size_t)(fuse2 ? MF_SPLITS2 : MF_SPLITS) * K * 8);

    {
        // group offsets | entities per group: one pinned block, one copy on the launch stream
        MD_PIN(h_g, unsigned char, (size_t)(2 * G + 1) * 8);
        memcpy(h_g, group_off, (size_t)(G + 1) * 8);
        double *h_ng = reinterpret_cast<double *>(h_g + (size_t)(G + 1) * 8);
        for (long long g = 0; g < G; ++g) h_ng[g] = (double)(group_off[g + 1] - group_off[g]);
        const int rc0 = mdhip_copy_small(ctx, d_goff, h_g, (size_t)(2 * G + 1) * 8, hipMemcpyHostToDevice);
        if (rc0) return rc0;
    }
    MD_HIP(hipMemsetAsync(d_P, 0, p_b, ctx->stream));

    KernelTimer timer(ctx);
    const long long m_stride = lag_mean_stride(ctx, F);
    if (m_stride > 1)
        hipLaunchKernelGGL(col_sum_sample_kernel, dim3((unsigned)((cols + 255) / 256), MF_SLABS), dim3(256), 0, ctx->stream, d_r, F,
                           cols, m_stride, d_msum);
    else
        hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)((cols + 255) / 256), MF_SLABS), dim3(256), 0, ctx->stream, d_r, F, cols,
                           d_msum);
    hipLaunchKernelGGL(col_mean_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, d_msum, MF_SLABS,
                       m_stride > 1 ? sample_count(F, MF_SLABS, m_stride) : F, cols, scale, d_mean);
    hipLaunchKernelGGL(frame_sq_kernel, dim3((unsigned)F, 3), dim3(256), 0, ctx->stream, d_r, d_mean, E, scale,
                       d_goff, (int)G, F, d_Q);
    MD_HIP(hipGetLastError());

    for (long long c_first = 0; c_first < cols; c_first += nb0) {
        const long long nb = std::min(nb0, cols - c_first);
        if (fuse)
            hipLaunchKernelGGL(transpose_centre64_kernel, dim3((unsigned)((nb + 63) / 64), (unsigned)((F + 63) / 64)),
                               dim3(256), 0, ctx->stream, d_r, d_mean, F, cols, c_first, nb, scale, d_pad);
        else
            hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)((nb + 31) / 32), (unsigned)((L + 31) / 32)),
                               dim3(256), 0, ctx->stream, d_r, d_mean, F, cols, c_first, nb, L, scale, d_pad);
        MD_HIP(hipGetLastError());
        const double2 *d_Zp = nullptr;  // fuse: the packed transform of the batch
        int rc = fuse2  ? mdhip_fft_first_perm(ctx, d_pad, F, d_spec, L, (int)nb)
                 : fuse ? mdhip_fft_r2c_packed(ctx, d_pad, F, d_spec, d_tmp, L, (int)nb, &d_Zp)
                        : mdhip_fft_r2c(ctx, d_pad, d_tmp, d_spec, L, (int)nb);
        if (rc) return rc;
        // the (axis, group) segments this batch touches
        for (long long s = 0; s < S; ++s) {
            const long long a = s / G, g = s % G;
            const long long lo = std::max(c_first, a * E + (long long)group_off[g]);
            const long long hi = std::min(c_first + nb, a * E + (long long)group_off[g + 1]);
            if (lo >= hi) continue;
            const int splits = (int)std::min<long long>(fuse2 ? MF_SPLITS2 : MF_SPLITS, hi - lo);
            if (fuse2) {
                const int rcp = mdhip_fft_power_pass(ctx, d_spec, L, lo - c_first, hi - c_first, splits, d_part);
                if (rcp) return rcp;
            } else if (fuse) {
                const int rcp = mdhip_fft_power_rows(ctx, d_Zp, L, lo - c_first, hi - c_first, splits, d_part);
                if (rcp) return rcp;
            } else {
                hipLaunchKernelGGL(power_rows_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)splits), dim3(256), 0,
                                   ctx->stream, d_spec, K, lo - c_first, hi - c_first, d_part);
            }
            hipLaunchKernelGGL(power_fold_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream,
                               d_part, splits, K, d_P + (size_t)s * K);
        }
        MD_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(real_to_complex_kernel, dim3((unsigned)((S * K + 255) / 256)), dim3(256), 0, ctx->stream, d_P,
                       S * K, d_Z);
    int rc = mdhip_fft_c2r(ctx, d_Z, d_tmp, d_corr, L, (int)S);
    if (rc) return rc;
    timer.stop();
    ctx->last_kernel = "lag_msd_fft";

    // the finish, on the device: means into `d_fin`, from there to the caller's buffer on the stream; only the segments'
    // bounds come back for the completion step (as lag_msd_fft_fused)
    const size_t fin_b = (size_t)n_lags * G * 4 * 8;
    MD_WS(d_fin_ws, unsigned char, WS_OUT3, fin_b + lag_bound_words(S) * 8 + (size_t)S * (F + 1) * sizeof(DD) + 64);
    double *d_fin = reinterpret_cast<double *>(d_fin_ws), *d_bound = d_fin + (size_t)n_lags * G * 4;
    DD *d_pre = reinterpret_cast<DD *>(d_bound + lag_bound_words(S));
    const double eps_l = 4.0 * 2.220446049250313e-16 * std::log2((double)L);
    hipLaunchKernelGGL(lag_finish_dd_kernel, dim3((unsigned)S), dim3(256), 0, ctx->stream, d_Q, d_corr, L, 1.0 / (double)L, F,
                       n_lags, (int)G, d_ng, eps_l, d_pre, d_fin, d_bound);
    hipLaunchKernelGGL(lag_total_kernel, dim3((unsigned)((n_lags * G + 255) / 256)), dim3(256), 0, ctx->stream, d_fin,
                       n_lags * G, d_bound, (int)S, (const unsigned *)nullptr);
    ctx->lag_status_dev = d_bound + S;  // (mdhip_lag_msd_status_dev: valid until the next call that uses WS_OUT3)
    MD_HIP(hipGetLastError());
    {
        const int rcr = mdhip_result(cs, out, d_fin, fin_b, out_on_device);
        if (rcr) return rcr;
    }
    MD_PIN(h_bound, double, lag_bound_words(S) * 8);
    {
        const int rcc = mdhip_copy_small(ctx, h_bound, d_bound, lag_bound_words(S) * 8, hipMemcpyDeviceToHost);
        if (rcc) return rcc;
    }
    cs.defer([timer, res, h_bound, S, n_lags]() {
        timer.collect();
        lag_collect_bounds(res.get(), h_bound, S, n_lags);
        return MDHIP_OK;
    });
    return MDHIP_OK;
}
