// msd.hip — frame-pair displacement reductions (M1, M2) for gfx950.
//
// Replaces the pandas expressions at dynamical/diffusion.py:212-218 (single origin) and
// diffusion.py:225-237 (fixed lag) of the reference.
//
// HBM-bound streaming kernels: a frame pair reads 2 x 24 bytes per entity (the origin frame of the
// M1 pair list is the same for every pair and stays in L2 / Infinity Cache, so the algorithmic
// traffic is 24*E bytes per frame pair) and does 12 flops. Coordinates are [F][3][E] planes so that
// consecutive lanes read consecutive doubles. Reductions are fixed-order (wave shuffle tree, then
// LDS across the 4 waves, then an ordered pass over block partials): no float atomics, results are
// reproducible run to run.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "ctx.h"

namespace {

constexpr int MSD_THREADS = 256;
constexpr int MSD_PER_THREAD = 4;
constexpr int MSD_CHUNK = MSD_THREADS * MSD_PER_THREAD;

struct Chunk {
    long long e0, e1;
    int group;
    int pad;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// (Round 5, measured and not kept: a plain read of the same planes reaches 0.89 of 8 TB/s — tools/ubench_hbm.hip,
// profiles/r05_ubench_hbm.txt — this kernel 0.73-0.75, and it reads the origin's 24 bytes per entity again for every
// pair, from L2. A block that keeps the origin's values in registers and walks 8 pairs, or requests 4 pairs' frames at
// once, took 0.65 / 0.72 ms against 0.40 ms for 2000 pairs at C4 shape: what this kernel lives on is a quarter of a
// million small independent blocks in flight, which the longer blocks give up. msd_windows_kernel with two entities
// per lane and four windows' frames requested together: 0.119 against 0.117 ms, no change. The four wave sums through
// DPP lanes (row_shr 1-8, row_bcast15 / 31: no LDS instruction) instead of 48 ds_bpermute per wave: 0.405 against
// 0.408 ms — the epilogue is not what holds the kernel either.)
// grid (n_chunks, n_pairs). partial [n_pairs][n_chunks][4]. VEC2: every lane handles two consecutive
// entities with 16-byte loads (needs an even entity count and even chunk starts, checked by the host).
template <bool VEC2>
__global__ __launch_bounds__(MSD_THREADS) void msd_pairs_kernel(
    const double *__restrict__ r, long long n_ent, double scale, const int *__restrict__ pairs,
    const Chunk *__restrict__ chunks, int n_chunks, double *__restrict__ partial,
    double *__restrict__ per_entity, long long pe_stride, const double *__restrict__ origin)
{
    // per_entity: pe_stride == 0 -> rows [n_pairs][n_ent][4]; otherwise four columns, column k at k * pe_stride,
    // each [n_pairs][n_ent] (the column blocks a DataFrame is made of, stored without a host-side transpose)
    __shared__ double red[4][4];
    const int p = blockIdx.y;
    const Chunk ck = chunks[blockIdx.x];
    // t0 < 0: the separate origin frame (a frame shard of the single-origin MSD: the origin was broadcast by its owner)
    const int t0 = pairs[2 * p];
    const double *r0 = t0 < 0 ? origin : r + (size_t)t0 * 3 * n_ent;
    const double *r1 = r + (size_t)pairs[2 * p + 1] * 3 * n_ent;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    auto one = [&](long long e, double x1, double y1, double z1, double x0, double y0, double z0) {
        // diffusion.py:201-203 scales first, diffusion.py:214 differences, then squares
        const double dx = x1 * scale - x0 * scale;
        const double dy = y1 * scale - y0 * scale;
        const double dz = z1 * scale - z0 * scale;
        const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
        const double tot = (dx2 + dy2) + dz2;  // diffusion.py:215
        s0 += dx2;
        s1 += dy2;
        s2 += dz2;
        s3 += tot;
        if (per_entity) {
            if (pe_stride) {
                double *pe = per_entity + (size_t)p * n_ent + e;
                __builtin_nontemporal_store(dx2, pe);
                __builtin_nontemporal_store(dy2, pe + pe_stride);
                __builtin_nontemporal_store(dz2, pe + 2 * pe_stride);
                __builtin_nontemporal_store(tot, pe + 3 * pe_stride);
            } else {
                double4 *pe = reinterpret_cast<double4 *>(per_entity + ((size_t)p * n_ent + e) * 4);
                *pe = make_double4(dx2, dy2, dz2, tot);
            }
        }
    };
    if (VEC2) {
#pragma unroll
        for (int u = 0; u < MSD_PER_THREAD / 2; ++u) {
            const long long e = ck.e0 + 2 * ((long long)u * MSD_THREADS + threadIdx.x);
            if (e + 1 < ck.e1) {
                typedef double d2_t __attribute__((ext_vector_type(2)));
                const d2_t x1v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(r1 + e));
                const d2_t y1v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(r1 + n_ent + e));
                const d2_t z1v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(r1 + 2 * n_ent + e));
                const double2 x1 = make_double2(x1v[0], x1v[1]), y1 = make_double2(y1v[0], y1v[1]),
                              z1 = make_double2(z1v[0], z1v[1]);
                const double2 x0 = *reinterpret_cast<const double2 *>(r0 + e);
                const double2 y0 = *reinterpret_cast<const double2 *>(r0 + n_ent + e);
                const double2 z0 = *reinterpret_cast<const double2 *>(r0 + 2 * n_ent + e);
                one(e, x1.x, y1.x, z1.x, x0.x, y0.x, z0.x);
                one(e + 1, x1.y, y1.y, z1.y, x0.y, y0.y, z0.y);
            } else if (e < ck.e1) {
                one(e, r1[e], r1[n_ent + e], r1[2 * n_ent + e], r0[e], r0[n_ent + e], r0[2 * n_ent + e]);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < MSD_PER_THREAD; ++u) {
            const long long e = ck.e0 + (long long)u * MSD_THREADS + threadIdx.x;
            if (e < ck.e1)
                one(e, r1[e], r1[n_ent + e], r1[2 * n_ent + e], r0[e], r0[n_ent + e], r0[2 * n_ent + e]);
        }
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    s3 = wave_sum(s3);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        red[wave][0] = s0;
        red[wave][1] = s1;
        red[wave][2] = s2;
        red[wave][3] = s3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        partial[((size_t)p * n_chunks + blockIdx.x) * 4 + c] =
            ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    }
}

// sums [n_pairs][n_groups][4] from partial, chunks of one group are contiguous in the chunk list
__global__ void msd_group_sum_kernel(const double *__restrict__ partial,
                                     const int *__restrict__ group_chunk_off, int n_chunks,
                                     int n_groups, int n_pairs, double *__restrict__ sums)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pairs * n_groups * 4) return;
    const int c = idx & 3, g = (idx >> 2) % n_groups, p = (idx >> 2) / n_groups;
    double s = 0.0;
    int k = group_chunk_off[g];
    const int k1 = group_chunk_off[g + 1];
    for (; k + 8 <= k1; k += 8) {  // (eight partials requested together, added in order: same bits, a third of the time)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[((size_t)p * n_chunks + k + u) * 4 + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < k1; ++k) s += partial[((size_t)p * n_chunks + k) * 4 + c];
    sums[idx] = s;
}

// one lane per (entity, slab of windows): windows are walked in order with the previous kept frame in
// registers; blockIdx.y slabs give the chip enough lanes when there are few entities. part [slabs][E][4].
__global__ __launch_bounds__(256) void msd_windows_kernel(const double *__restrict__ r,
                                                          long long n_ent, long long n_kept,
                                                          double scale, int tao, int n_slabs,
                                                          double *__restrict__ part)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    // kept frames k = 0..n_kept-1 are frames k*tao; window k pairs kept frame k with k-1, k = 1..n_kept-1
    const long long n_win = n_kept - 1;
    const long long per = (n_win + n_slabs - 1) / n_slabs;
    const long long k0 = 1 + (long long)blockIdx.y * per;
    long long k1 = k0 + per;
    if (k1 > n_kept) k1 = n_kept;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (k0 < k1) {
        const double *rp = r + (size_t)(k0 - 1) * tao * 3 * n_ent;
        double px = rp[e] * scale, py = rp[n_ent + e] * scale, pz = rp[2 * n_ent + e] * scale;
        for (long long k = k0; k < k1; ++k) {
            const double *rt = r + (size_t)k * tao * 3 * n_ent;
            // (streamed once: nontemporal loads, +5 % on msd_pairs_kernel build against build, tools/ab_libs_msd.py)
            const double x = __builtin_nontemporal_load(rt + e) * scale, y = __builtin_nontemporal_load(rt + n_ent + e) * scale,
                         z = __builtin_nontemporal_load(rt + 2 * n_ent + e) * scale;
            const double dx = x - px, dy = y - py, dz = z - pz;  // diffusion.py:232
            const double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
            s0 += dx2;
            s1 += dy2;
            s2 += dz2;
            s3 += (dx2 + dy2) + dz2;  // diffusion.py:235
            px = x;
            py = y;
            pz = z;
        }
    }
    double4 *o = reinterpret_cast<double4 *>(part + ((size_t)blockIdx.y * n_ent + e) * 4);
    *o = make_double4(s0, s1, s2, s3);
}

__global__ void msd_windows_sum_kernel(const double *__restrict__ part, long long n_ent, int n_slabs,
                                       double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_ent * 4) return;
    double s = 0.0;
    for (int k = 0; k < n_slabs; ++k) s += part[(size_t)k * n_ent * 4 + i];
    out[i] = s;
}

// ---------------------------------------------------------------------------------------------
// full lag average (superset of the reference): msd[lag] = mean_{t0, e} |r_e(t0+lag) - r_e(t0)|^2
// ---------------------------------------------------------------------------------------------
// FP64-bound (12 flops per entity per frame pair, F^2/2 frame pairs). The trajectory is transposed
// once to time-major series x[c][e][t]; a block then owns (entity chunk, pair of lag tiles): for
// every entity and axis it stages the series through LDS (transposed [i mod 8][i div 8] so that a
// wave reads consecutive doubles) and every lane keeps 8 consecutive lags with a 16-deep sliding
// window in registers: 8 LDS reads feed 64 (sub, fma) pairs. Sums over the chunk's entities stay in
// registers; chunk partials are added in a fixed order afterwards.

constexpr int LG_LPT = 8;       // consecutive lags per lane
constexpr int LG_TT = 2048;     // time steps per LDS stage
constexpr int LG_ECHUNK = 64;   // at most this many entities are summed inside one block

__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ in,
                                                        double *__restrict__ out, long long rows,
                                                        long long cols, double scale)
{
    // out[col][row] = in[row][col] * scale, 32x32 tiles through LDS
    __shared__ double tile[32][33];
    const long long c0 = (long long)blockIdx.x * 32, r0 = (long long)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const long long rr = r0 + k, cc = c0 + tx;
        tile[k][tx] = (rr < rows && cc < cols) ? in[rr * cols + cc] * scale : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const long long cc = c0 + k, rr = r0 + tx;
        if (rr < rows && cc < cols) out[cc * rows + rr] = tile[tx][k];
    }
}

// x: [3][E][F] time-major series. partial: [n_chunks][max_lag+1][4]. LG_THREADS lanes x 8 lags = one lag
// tile; short trajectories use fewer lanes per block so that no lane owns only lags >= n.
template <int LG_THREADS>
__global__ __launch_bounds__(LG_THREADS) void lag_msd_kernel(
    const double *__restrict__ x, long long n_ent, long long n, long long n_lags,
    const Chunk *__restrict__ chunks, int n_tiles, double *__restrict__ partial)
{
    constexpr int LG_KT = LG_THREADS * LG_LPT;
    constexpr int LG_AW = LG_TT + LG_KT + 8;
    constexpr int LG_ROW = LG_AW / 8 + 1;
    __shared__ double s_a[8 * LG_ROW];
    __shared__ __attribute__((aligned(16))) double s_b[LG_TT];
    const int tid = threadIdx.x;
    const Chunk ck = chunks[blockIdx.y];
    const int pair_id = blockIdx.x;
    for (int half = 0; half < 2; ++half) {
        const int tile = half == 0 ? pair_id : n_tiles - 1 - pair_id;
        if (half == 1 && tile == pair_id) break;
        const long long K0 = (long long)tile * LG_KT;
        if (K0 >= n_lags) continue;
        const long long kb = K0 + (long long)tid * LG_LPT;
        const long long lim = n - kb;  // pair (t, kb+m) is valid iff t + m < lim
        const long long t_total = n - K0;
        double acc[3][LG_LPT];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int m = 0; m < LG_LPT; ++m) acc[c][m] = 0.0;

        for (long long e = ck.e0; e < ck.e1; ++e) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double *xs = x + ((size_t)c * n_ent + e) * n;
                for (long long T0 = 0; T0 < t_total; T0 += LG_TT) {
                    __syncthreads();
                    for (int i = tid; i < LG_TT; i += LG_THREADS) {
                        const long long t = T0 + i;
                        s_b[i] = t < n ? xs[t] : 0.0;
                    }
                    for (int i = tid; i < LG_AW; i += LG_THREADS) {
                        const long long g = T0 + K0 + i;
                        s_a[(i & 7) * LG_ROW + (i >> 3)] = g < n ? xs[g] : 0.0;
                    }
                    __syncthreads();
                    double w[16];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = s_a[j * LG_ROW + tid];
                    // time steps of this stage that can still pair with the tile's smallest lag
                    const long long left = t_total - T0;
                    const int tt_end = left < LG_TT ? (int)((left + 7) & ~7LL) : LG_TT;
                    for (int tt = 0; tt < tt_end; tt += 8) {
                        const int col = tid + (tt >> 3) + 1;
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[8 + j] = s_a[j * LG_ROW + col];
                        const long long s = lim - (T0 + tt);
                        if (s >= 15) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double bt = s_b[tt + u];
#pragma unroll
                                for (int m = 0; m < LG_LPT; ++m) {
                                    const double d = w[u + m] - bt;
                                    acc[c][m] = __builtin_fma(d, d, acc[c][m]);
                                }
                            }
                        } else if (s > 0) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double bt = s_b[tt + u];
#pragma unroll
                                for (int m = 0; m < LG_LPT; ++m) {
                                    const double d = (u + m) < s ? w[u + m] - bt : 0.0;
                                    acc[c][m] = __builtin_fma(d, d, acc[c][m]);
                                }
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) w[j] = w[8 + j];
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < LG_LPT; ++m) {
            const long long k = kb + m;
            if (k < n_lags) {
                double *p = partial + ((size_t)blockIdx.y * n_lags + k) * 4;
                p[0] = acc[0][m];
                p[1] = acc[1][m];
                p[2] = acc[2][m];
                p[3] = (acc[0][m] + acc[1][m]) + acc[2][m];
            }
        }
    }
}

// Series-resident variant (default whenever one series plus its padding fits LDS, n <~ 19 000 frames).
// The whole series of one (entity, axis) sits in LDS (padded natural order, see LX below), and is
// both the broadcast operand x[t] and the per-lane sliding window x[t + lag]. A WAVE owns a pair of
// lag tiles of 512 lags, (j, nT-1-j), and walks each only as far as that tile's own longest valid
// origin range (n - K0), so the idle lanes of the lag x origin triangle are confined to the last 512
// origins of every tile and every wave of the block does the same amount of work (n + 512 origins per
// series). Axes are the outer loop: 2 tiles x 8 lags of accumulators per lane instead of 3 x 8 per tile.
template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void lag_msd_lds_kernel(
    const double *__restrict__ x, long long n_ent, int n, int n_lags, const Chunk *__restrict__ chunks,
    int n_tiles, int row, double *__restrict__ partial)
{
    // the series, zero beyond n, one pad double behind every 8 entries: entry i at LX(i) = i + (i >> 3). The lanes of a
    // wave read entries 8 apart (9 doubles apart in LDS: no bank conflicts) and a lane's own entries follow each other
    // at offsets that are compile-time constants from ONE address register (kb and the trip's t are multiples of 8);
    // the round-1 layout [i mod 8][i div 8] needed a register per row — the row length depends on the series — and
    // the compiler spilled them
    extern __shared__ double s_x[];
#define LX(i) ((i) + ((i) >> 3))
    constexpr int KT = 64 * LG_LPT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Chunk ck = chunks[blockIdx.y];
    const int pair_id = blockIdx.x * NW + wave;
    // this wave's tiles (tile[1] < 0: none — the middle tile of an odd count is walked once)
    int tile[2] = {pair_id, n_tiles - 1 - pair_id};
    if (tile[1] <= tile[0]) tile[1] = -1;
    if (tile[0] > n_tiles - 1 - pair_id) tile[0] = -1;
    const int n_pad = 8 * (row - 1);  // entries the host sized the stage for
    for (int c = 0; c < 3; ++c) {
        double acc[2][LG_LPT];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < LG_LPT; ++m) acc[h][m] = 0.0;
        for (long long e = ck.e0; e < ck.e1; ++e) {
            const double *xs = x + ((size_t)c * n_ent + e) * n;
            __syncthreads();  // the previous series has been consumed
            for (int base = 0; base < n_pad; base += NW * 64 * 4) {  // four loads in flight per lane
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = base + q * NW * 64 + tid;
                    v[q] = i < n ? xs[i] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = base + q * NW * 64 + tid;
                    if (i < n_pad) s_x[LX(i)] = v[q];
                }
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (tile[h] < 0) continue;  // wave-uniform
                const int K0 = tile[h] * KT;
                const int kb = K0 + lane * LG_LPT;
                // pair (t, kb + m) is valid iff t + m < lim; lanes whose lags nobody asked for do nothing
                const int lim = kb < n_lags ? n - kb : 0;
                // window x[t + kb + j], j = 0..15, as two halves that swap roles every 8 origins (no moves)
                double w[16];  // the ring; wa / wb name its halves for the finishing steps (t is a multiple of 16 there)
#define wa (w)
#define wb (w + 8)
#pragma unroll
                for (int j = 0; j < 8; ++j) wa[j] = s_x[LX(kb) + j];  // (kb & 7) == 0
// One step of 8 origins x 8 lags for the lanes whose whole 8 x 8 block is valid (the others are masked off:
// their few remaining pairs are swept up after the loop). A holds x[T+kb+0..7], B is loaded with x[T+kb+8..15].
#define LG_STEP(A, B, T, BC, BN)                                                        \
    {                                                                                   \
        const double *next_ = s_x + LX((T) + kb) + 9;  /* entry T + kb + 8 */          \
        /* x[T+8 .. T+15] for the NEXT step: wave-uniform scalar loads, in flight during this step */ \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) BN[u] = xs[(T) + 8 + u];          \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) B[j] = next_[j];                  \
        if (lim - (T) >= 15) {                                                          \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                               \
            {                                                                           \
                /* differences first, then the FMAs: a dependent pair back to back leaves the FP64 pipe idle */ \
                double d_[LG_LPT];                                                      \
                _Pragma("unroll") for (int m = 0; m < LG_LPT; ++m)                      \
                    d_[m] = ((u + m) < 8 ? A[(u + m) & 7] : B[(u + m) & 7]) - BC[u];    \
                __builtin_amdgcn_sched_barrier(0);                                      \
                _Pragma("unroll") for (int m = 0; m < LG_LPT; ++m)                      \
                    acc[h][m] = __builtin_fma(d_[m], d_[m], acc[h][m]);                 \
                __builtin_amdgcn_sched_barrier(0);                                      \
            }                                                                           \
        }                                                                               \
    }
                const int t_wave = n - K0 - 15;  // lane 0 (smallest lag) has a whole block up to here
                double b0[8], b1[8];             // x[t .. t+7] of the current / next step, SGPRs (ping-pong)
#pragma unroll
                for (int u = 0; u < 8; ++u) b0[u] = xs[u];
                int t = 0;
                // Fast trips (round 2; cf. xcorr_direct_kernel). The window is a ring of 16 registers, w[i & 15] =
                // x[kb + i]: origin o of a trip uses w[o .. o+7], after which w[o] is dead and takes the entry 16
                // further on (one address register, immediate offsets) — 8 origins, ~550 cycles, before
                // its first use. The operations of a half trip's FIRST origin come before its scalar request (b of
                // the next half): they need this half's b, whose scalar load can only be waited for with lgkmcnt(0),
                // and that wait then meets requests that are 8 origins old plus the ring reads of the origin before.
                // (The round-1 step requested b and its 8 window entries at the top and needed them within the same
                // step: every step paid a scalar-load latency. Three rotating window groups, as in xcorr_direct,
                // spill here: 32 accumulators are live.)
// (the ring loads are asm: written as plain loads the compiler gathers a half's eight reads behind its first origin
// and waits for them on the spot. The wait for them is the one at the top of the next half — the wait asm names the
// registers, so no use of a new value can be moved above it.)
#define LG_WAIT8(B0)                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                 \
                 : "+v"(w[(B0) + 0]), "+v"(w[(B0) + 1]), "+v"(w[(B0) + 2]), "+v"(w[(B0) + 3]), "+v"(w[(B0) + 4]), \
                   "+v"(w[(B0) + 5]), "+v"(w[(B0) + 6]), "+v"(w[(B0) + 7]))
#define LG_HALF(O0, BC, BN, T, BNOFF)                                                   \
    {                                                                                   \
        LG_WAIT8(8 - (O0)); /* the reads issued during the previous half */             \
        _Pragma("unroll") for (int o = (O0); o < (O0) + 8; ++o)                         \
        {                                                                               \
            double d_[LG_LPT];                                                          \
            _Pragma("unroll") for (int m = 0; m < LG_LPT; ++m) d_[m] = w[(o + m) & 15] - BC[o - (O0)]; \
            __builtin_amdgcn_sched_barrier(0);                                          \
            _Pragma("unroll") for (int m = 0; m < LG_LPT; ++m)                          \
                acc[h][m] = __builtin_fma(d_[m], d_[m], acc[h][m]);                     \
            __builtin_amdgcn_sched_barrier(0);                                          \
            if (o == (O0)) {                                                            \
                _Pragma("unroll") for (int u = 0; u < 8; ++u) BN[u] = xs[(T) + (BNOFF) + u]; \
            }                                                                           \
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[o]) : "v"(wt), "n"(LX(o + 16) * 8)); \
            __builtin_amdgcn_sched_barrier(0);                                          \
        }                                                                               \
    }
                {
                    const double *wrow = s_x + LX(kb);  // LX(kb + e) = LX(kb) + LX(e): kb is a multiple of 8
#pragma unroll
                    for (int j = 0; j < 8; ++j) wb[j] = wrow[9 + j];
                    // while EVERY lane of the wave has whole 8 x 8 blocks (the lane with the largest lag runs out
                    // first) the trips run without exec masking — the lanes whose lags nobody asked for compute on the
                    // zero padding and are never stored; the masked steps below walk the remaining diagonal band
                    const int t_all = n - (K0 + 63 * LG_LPT) - 15;
                    const unsigned wrow_lds = (unsigned)(unsigned long long)wrow;  // LDS byte address
                    for (; t + 8 <= t_all; t += 16) {
                        const unsigned wt = wrow_lds + 144u * (unsigned)(t >> 4);  // + LX(t) * 8: t is a multiple of 16
                        LG_HALF(0, b0, b1, t, 8)
                        LG_HALF(8, b1, b0, t, 16)
                    }
                    LG_WAIT8(0);
                    LG_WAIT8(8);
                }
#undef LG_WAIT8
#undef LG_HALF
                for (; t + 8 <= t_wave; t += 16) {
                    LG_STEP(wa, wb, t, b0, b1)
                    LG_STEP(wb, wa, t + 8, b1, b0)
                }
                if (t <= t_wave) LG_STEP(wa, wb, t, b0, b1)
#undef LG_STEP
#undef wa
#undef wb
                // the pairs of the last, partial blocks of every lag: < 22 origins per lane
                {
                    const int t_stop = lim >= 15 ? ((lim - 15) / 8 + 1) * 8 : 0;
                    for (int tt = t_stop; tt < lim; ++tt) {
                        const double bt = s_x[LX(tt)];
#pragma unroll
                        for (int m = 0; m < LG_LPT; ++m) {
                            const int i = tt + kb + m;
                            const double d = s_x[LX(i)] - bt;
                            if (tt + m < lim) acc[h][m] = __builtin_fma(d, d, acc[h][m]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (tile[h] < 0) continue;
#pragma unroll
            for (int m = 0; m < LG_LPT; ++m) {
                const long long k = (long long)tile[h] * KT + lane * LG_LPT + m;
                if (k < n_lags) partial[((size_t)blockIdx.y * n_lags + k) * 4 + c] = acc[h][m];
            }
        }
    }
}

#undef LX

// resident blocks per CU as the runtime computes it (registers, LDS, wave slots)
int lag_lds_blocks_per_cu(int nw, size_t lds_b)
{
    int n = 0;
#define MD_OCC(NW)                                                                                               \
    case NW:                                                                                                     \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lag_msd_lds_kernel<NW>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);                       \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(                                                      \
            &n, reinterpret_cast<const void *>(lag_msd_lds_kernel<NW>), NW * 64, lds_b);                         \
        break;
    switch (nw) {
        MD_OCC(1) MD_OCC(2) MD_OCC(3) MD_OCC(4) MD_OCC(5) MD_OCC(6) MD_OCC(7) MD_OCC(8)
    }
#undef MD_OCC
    return n;
}

// out[lag][g][c] = sum over the group's chunks / ((n - lag) * group size)
__global__ void lag_msd_finish_kernel(const double *__restrict__ partial,
                                      const int *__restrict__ group_chunk_off,
                                      const long long *__restrict__ group_off, int n_groups,
                                      long long n, long long n_lags, int axes_only, double *__restrict__ out)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_lags * n_groups * 4) return;
    const int c = (int)(idx & 3);
    const int g = (int)((idx >> 2) % n_groups);
    const long long k = (idx >> 2) / n_groups;
    double s = 0.0;
    for (int q = group_chunk_off[g]; q < group_chunk_off[g + 1]; ++q) {
        const double *p = partial + ((size_t)q * n_lags + k) * 4;
        s += (c < 3 || !axes_only) ? p[c] : (p[0] + p[1]) + p[2];  // total column: same sum as the staged kernel
    }
    const double cnt = (double)(n - k) * (double)(group_off[g + 1] - group_off[g]);
    out[idx] = cnt > 0.0 ? s / cnt : 0.0;
}

int build_chunks(mdhip_ctx *ctx, int64_t n_ent, int n_groups, const int64_t *group_off,
                 std::vector<Chunk> &chunks, std::vector<int> &gco, int64_t chunk_len = MSD_CHUNK)
{
    MD_REQUIRE(n_groups >= 1 && group_off, "need at least one entity group");
    MD_REQUIRE(group_off[0] >= 0 && group_off[n_groups] <= n_ent, "group_off outside [0, n_ent]");
    gco.assign(n_groups + 1, 0);
    for (int g = 0; g < n_groups; ++g) {
        MD_REQUIRE(group_off[g] <= group_off[g + 1], "group_off must be non-decreasing");
        gco[g] = (int)chunks.size();
        for (int64_t e = group_off[g]; e < group_off[g + 1]; e += chunk_len)
            chunks.push_back({(long long)e, (long long)std::min<int64_t>(e + chunk_len, group_off[g + 1]), g, 0});
    }
    gco[n_groups] = (int)chunks.size();
    return MDHIP_OK;
}

// per_entity layouts: col_stride == 0 -> rows [n_pairs][n_ent][4]; > 0 -> four columns [n_pairs * n_ent] that lie
// col_stride doubles apart at the destination (packed on the device, copied column by column)
int msd_pairs_impl(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device, double scale,
                   int n_pairs, const int32_t *pairs, int n_groups, const int64_t *group_off, double *sums,
                   double *per_entity, int pe_on_device, int64_t col_stride, int sums_on_device = 0,
                   const double *origin = nullptr, int origin_on_device = 0)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n_frames >= 0 && n_ent >= 0 && n_pairs >= 0, "negative sizes");
    MD_REQUIRE(n_pairs == 0 || (pairs && sums), "NULL pairs/sums");
    for (int p = 0; p < 2 * n_pairs; ++p)
        MD_REQUIRE((pairs[p] >= 0 && pairs[p] < n_frames) || (origin && p % 2 == 0 && pairs[p] == -1),
                   "pair index %d out of range", pairs[p]);
    std::vector<Chunk> chunks;
    std::vector<int> gco;
    int rc = build_chunks(ctx, n_ent, n_groups, group_off, chunks, gco);
    if (rc) return rc;
    MD_HIP(hipSetDevice(ctx->device));
    rc = mdhip_zero_result(ctx, sums, (size_t)n_pairs * n_groups * 4 * 8, sums_on_device);
    if (rc) return rc;
    if (n_pairs == 0 || chunks.empty()) return cs.end();
    MD_REQUIRE(r != nullptr, "r is NULL");
    MD_REQUIRE(n_pairs <= 65535, "at most 65535 frame pairs per call");
    const double *d_r =
        (const double *)mdhip_stage(ctx, WS_XYZ_I, r, (size_t)n_frames * 3 * n_ent * 8, on_device, &rc);
    if (rc) return rc;
    const double *d_origin = nullptr;
    if (origin) {
        d_origin = (const double *)mdhip_stage(ctx, WS_XYZ_J, origin, (size_t)3 * n_ent * 8, origin_on_device, &rc);
        if (rc) return rc;
    }
    const int n_chunks = (int)chunks.size();
    // chunks | pairs | group -> chunk offsets: one pinned staging block, one copy
    const size_t ch_b = chunks.size() * sizeof(Chunk), pr_b = (size_t)n_pairs * 8, gc_b = gco.size() * 4;
    const size_t tab_b = ch_b + pr_b + gc_b;
    MD_WS(d_tab, unsigned char, WS_TABLES, tab_b + 64);
    MD_PIN(h_tab, unsigned char, tab_b);
    memcpy(h_tab, chunks.data(), ch_b);
    memcpy(h_tab + ch_b, pairs, pr_b);
    memcpy(h_tab + ch_b + pr_b, gco.data(), gc_b);
    rc = mdhip_copy_small(ctx, d_tab, h_tab, tab_b, hipMemcpyHostToDevice);
    if (rc) return rc;
    Chunk *d_chunks = reinterpret_cast<Chunk *>(d_tab);
    int *d_pairs = reinterpret_cast<int *>(d_tab + ch_b);
    int *d_gco = d_pairs + 2 * n_pairs;
    MD_WS(d_partial, double, WS_PART, (size_t)n_pairs * n_chunks * 4 * 8);
    const size_t sums_b = (size_t)n_pairs * n_groups * 4 * 8;
    double *d_sums = sums;
    if (!sums_on_device) {
        d_sums = (double *)mdhip_ws(ctx, WS_OUT, sums_b);
        if (!d_sums) return MDHIP_ENOMEM;
    }
    double *d_pe = nullptr;
    const size_t pe_b = (size_t)n_pairs * n_ent * 4 * 8;
    if (per_entity) {
        d_pe = pe_on_device ? per_entity : (double *)mdhip_ws(ctx, WS_AUX0, pe_b);
        if (!d_pe) return MDHIP_ENOMEM;
    }
    const long long col_len = (long long)n_pairs * n_ent;
    const long long d_stride = col_stride ? (pe_on_device ? (long long)col_stride : col_len) : 0;
    KernelTimer timer(ctx);
    ctx->last_kernel = "msd_pairs_kernel";
    // 16-byte loads need 16-byte aligned planes and even chunk starts
    bool vec2 = (n_ent % 2 == 0) && ((reinterpret_cast<uintptr_t>(d_r) & 15) == 0) &&
                ((reinterpret_cast<uintptr_t>(d_origin) & 15) == 0);
    for (const Chunk &c : chunks) vec2 = vec2 && (c.e0 % 2 == 0);
    if (vec2)
        hipLaunchKernelGGL(msd_pairs_kernel<true>, dim3((unsigned)n_chunks, (unsigned)n_pairs),
                           dim3(MSD_THREADS), 0, ctx->stream, d_r, (long long)n_ent, scale, d_pairs,
                           d_chunks, n_chunks, d_partial, d_pe, d_stride, d_origin);
    else
        hipLaunchKernelGGL(msd_pairs_kernel<false>, dim3((unsigned)n_chunks, (unsigned)n_pairs),
                           dim3(MSD_THREADS), 0, ctx->stream, d_r, (long long)n_ent, scale, d_pairs,
                           d_chunks, n_chunks, d_partial, d_pe, d_stride, d_origin);
    timer.stop();
    MD_HIP(hipGetLastError());
    const int tot = n_pairs * n_groups * 4;
    hipLaunchKernelGGL(msd_group_sum_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_partial, d_gco, n_chunks, n_groups, n_pairs, d_sums);
    MD_HIP(hipGetLastError());
    if (!sums_on_device) {
        rc = mdhip_result(cs, sums, d_sums, sums_b, 0);
        if (rc) return rc;
    }
    if (per_entity && !pe_on_device) {
        // (F x E x 4 values: straight into the caller's memory, see mdhip_result)
        if (!col_stride || col_stride == col_len) {
            MD_HIP(hipMemcpyAsync(per_entity, d_pe, pe_b, hipMemcpyDeviceToHost, ctx->stream));
        } else {
            for (int k = 0; k < 4; ++k)
                MD_HIP(hipMemcpyAsync(per_entity + (size_t)k * col_stride, d_pe + (size_t)k * col_len,
                                      (size_t)col_len * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

}  // namespace

extern "C" {

int mdhip_msd_pairs(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r,
                    int on_device, double scale, int n_pairs, const int32_t *pairs, int n_groups,
                    const int64_t *group_off, double *sums, double *per_entity, int pe_on_device)
{
    return msd_pairs_impl(ctx, n_frames, n_ent, r, on_device, scale, n_pairs, pairs, n_groups, group_off, sums,
                          per_entity, pe_on_device, 0);
}

int mdhip_msd_pairs_cols(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r,
                         int on_device, double scale, int n_pairs, const int32_t *pairs, int n_groups,
                         const int64_t *group_off, double *sums, double *cols, int64_t col_stride,
                         int cols_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(cols != nullptr || (int64_t)n_pairs * n_ent == 0, "cols is NULL");
    MD_REQUIRE(col_stride >= (int64_t)n_pairs * n_ent, "col_stride is shorter than one column");
    return msd_pairs_impl(ctx, n_frames, n_ent, r, on_device, scale, n_pairs, pairs, n_groups, group_off, sums, cols,
                          cols_on_device, col_stride ? col_stride : 1);
}

static int msd_windows_impl(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                            double scale, int tao, double *win_sums, int out_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    MD_REQUIRE(n_frames >= 0 && n_ent >= 0 && tao >= 1, "bad sizes");
    MD_REQUIRE(n_ent == 0 || win_sums, "win_sums is NULL");
    if (n_ent == 0) return cs.end();
    MD_HIP(hipSetDevice(ctx->device));
    int rc;
    if (n_frames == 0) {
        rc = mdhip_zero_result(ctx, win_sums, (size_t)n_ent * 4 * 8, out_on_device);
        return rc ? rc : cs.end();
    }
    MD_REQUIRE(r != nullptr, "r is NULL");
    const double *d_r =
        (const double *)mdhip_stage(ctx, WS_XYZ_I, r, (size_t)n_frames * 3 * n_ent * 8, on_device, &rc);
    if (rc) return rc;
    double *d_out = win_sums;
    if (!out_on_device) {
        d_out = (double *)mdhip_ws(ctx, WS_OUT, (size_t)n_ent * 4 * 8);
        if (!d_out) return MDHIP_ENOMEM;
    }
    const long long n_kept = (n_frames + tao - 1) / tao;
    const long long n_blocks_e = (n_ent + 255) / 256;
    long long n_slabs = ((long long)ctx->cu_count * 8 + n_blocks_e - 1) / n_blocks_e;
    if (n_slabs > n_kept - 1) n_slabs = n_kept - 1;
    if (n_slabs > 1024) n_slabs = 1024;
    if (n_slabs < 1) n_slabs = 1;
    MD_WS(d_part, double, WS_PART, (size_t)n_slabs * n_ent * 4 * 8);
    KernelTimer timer(ctx);
    ctx->last_kernel = "msd_windows_kernel";
    hipLaunchKernelGGL(msd_windows_kernel, dim3((unsigned)n_blocks_e, (unsigned)n_slabs), dim3(256), 0,
                       ctx->stream, d_r, (long long)n_ent, n_kept, scale, tao, (int)n_slabs, d_part);
    timer.stop();  // (the reported time is the dominant kernel's, as for msd_pairs_kernel: the 5 us sum of the slabs and the
                   // hand-over to it are not part of what the roofline prices)
    hipLaunchKernelGGL(msd_windows_sum_kernel, dim3((unsigned)((n_ent * 4 + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_part, (long long)n_ent, (int)n_slabs, d_out);
    MD_HIP(hipGetLastError());
    if (!out_on_device) {
        rc = mdhip_result(cs, win_sums, d_out, (size_t)n_ent * 4 * 8, 0);
        if (rc) return rc;
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

int mdhip_msd_windows(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r,
                      int on_device, double scale, int tao, double *win_sums)
{
    return msd_windows_impl(ctx, n_frames, n_ent, r, on_device, scale, tao, win_sums, 0);
}

int mdhip_msd_windows_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                          double scale, int tao, double *win_sums_dev)
{
    return msd_windows_impl(ctx, n_frames, n_ent, r, on_device, scale, tao, win_sums_dev, 1);
}

int mdhip_msd_origin(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                     const double *origin, int origin_on_device, double scale, int n_groups,
                     const int64_t *group_off, double *sums, int sums_on_device, double *cols, int64_t col_stride,
                     int cols_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(n_frames >= 0 && n_frames <= 65535, "1..65535 frames per call");
    MD_REQUIRE(origin != nullptr || n_ent == 0 || n_frames == 0, "origin is NULL");
    MD_REQUIRE(sums != nullptr || n_frames == 0, "sums is NULL");
    MD_REQUIRE(!cols || col_stride >= n_frames * n_ent, "col_stride is shorter than one column");
    std::vector<int32_t> pairs((size_t)2 * n_frames);
    for (int64_t t = 0; t < n_frames; ++t) {
        pairs[2 * t] = -1;
        pairs[2 * t + 1] = (int32_t)t;
    }
    return msd_pairs_impl(ctx, n_frames, n_ent, r, on_device, scale, (int)n_frames, pairs.data(), n_groups, group_off,
                          sums, cols, cols_on_device, cols ? (col_stride ? col_stride : 1) : 0, sums_on_device, origin,
                          origin_on_device);
}

int mdhip_msd_pairs_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device, double scale,
                        int n_pairs, const int32_t *pairs, int n_groups, const int64_t *group_off, double *sums_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(sums_dev != nullptr || n_pairs == 0, "sums_dev is NULL");
    return msd_pairs_impl(ctx, n_frames, n_ent, r, on_device, scale, n_pairs, pairs, n_groups, group_off, sums_dev,
                          nullptr, 0, 0, 1);
}

// A host result [n_lags][G][4] that is ready in `src` goes to the caller: host memory at once, device memory through a
// copy of its own (a small call inside the completion step that runs this).
static int deliver_host_values(mdhip_ctx *ctx, const double *src, size_t bytes, double *dst, int dst_on_device)
{
    if (!dst_on_device) {
        memcpy(dst, src, bytes);
        return MDHIP_OK;
    }
    return mdhip_deliver_to_device(ctx, dst, src, bytes);
}

// ---- Round 6: the exact difference form for a FEW lags at the two ends of the lag range -------------------------------
// The spectral path's rounding error is the same absolute amount at every lag; relative to the MSD it is largest where
// (origins x MSD) is smallest: the first lags (small displacement) and the last ones (few origins). A diffusive or ballistic
// trajectory a little too long for the 1e-10 bound misses it at a handful of lags only — the whole call then used to go to
// the difference kernel, O(F^2 E): seconds where the spectral path takes milliseconds. The completion step now asks
// msd_fft.hip which lags miss the bound (lag_finish_dd_kernel: none beyond [1, k_lo] and [k_hi, max_lag]) and, when those
// are at most LAG_ENDS_MAX (24) per end, recomputes just them from the trajectory itself and writes them over the spectral
// values: one pass over the trajectory for the low lags (a window of k_lo + 1 frames in registers per series), a few rows
// for the high ones.
constexpr int LAG_ENDS_MAX = 24;
constexpr int LAG_ENDS_SLABS = 16;

// part[(slab * kl + k - 1) * cols + c] = sum over the slab's origins t (t + k < F) of (r[t + k][c] - r[t][c])^2 * scale^2,
// k = 1 .. kl; c = axis * E + entity. grid (ceil(cols / 256), LAG_ENDS_SLABS)
__global__ __launch_bounds__(256) void lag_low_lags_kernel(const double *__restrict__ r, long long F, long long cols, double scale,
                                                           int kl, double *__restrict__ part)
{
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const long long t0 = F * blockIdx.y / gridDim.y, t1 = F * (blockIdx.y + 1) / gridDim.y;
    double w[LAG_ENDS_MAX + 1], acc[LAG_ENDS_MAX];
#pragma unroll
    for (int k = 0; k < LAG_ENDS_MAX; ++k) acc[k] = 0.0;
#pragma unroll
    for (int k = 0; k <= LAG_ENDS_MAX; ++k) w[k] = (k <= kl && t0 + k < F) ? r[(t0 + k) * cols + c] * scale : 0.0;
    for (long long t = t0; t < t1; ++t) {
#pragma unroll
        for (int k = 1; k <= LAG_ENDS_MAX; ++k)
            if (k <= kl && t + k < F) {
                const double d = w[k] - w[0];
                acc[k - 1] += d * d;
            }
#pragma unroll
        for (int k = 0; k < LAG_ENDS_MAX; ++k) w[k] = w[k + 1];
        // (the window's new last frame: t + 1 + kl; the slots above kl are never read)
#pragma unroll
        for (int k = 1; k <= LAG_ENDS_MAX; ++k)
            if (k == kl) w[k] = t + 1 + kl < F ? r[(t + 1 + kl) * cols + c] * scale : 0.0;
    }
#pragma unroll
    for (int k = 1; k <= LAG_ENDS_MAX; ++k)
        if (k <= kl) part[((size_t)blockIdx.y * kl + (k - 1)) * cols + c] = acc[k - 1];
}

// part[j * cols + c] = sum over the F - k origins of lag k = k_hi + j (at most LAG_ENDS_MAX + ... origins each)
__global__ __launch_bounds__(256) void lag_high_lags_kernel(const double *__restrict__ r, long long F, long long cols, double scale,
                                                            long long k_hi, int n_hi, double *__restrict__ part)
{
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    for (int j = 0; j < n_hi; ++j) {
        const long long k = k_hi + j;
        double acc = 0.0;
        for (long long t = 0; t + k < F; ++t) {
            const double d = r[(t + k) * cols + c] * scale - r[t * cols + c] * scale;
            acc += d * d;
        }
        part[(size_t)j * cols + c] = acc;
    }
}

// rows[(i * G + g) * 4 + a] = (sum over the group's entities and the `slabs` partial rows of lag i) / ((F - k_i) n_g);
// k_i = k0 + i. One block per (lag i, axis a, group g); the sum in a fixed order (per-lane strided, then a tree).
__global__ __launch_bounds__(256) void lag_ends_fold_kernel(const double *__restrict__ part, int slabs, int n_rows, long long cols,
                                                            long long E, const long long *__restrict__ goff, int G, long long F,
                                                            long long k0, double *__restrict__ rows)
{
    __shared__ double red[256];
    const int i = blockIdx.x, a = blockIdx.y, g = blockIdx.z;
    const long long lo = goff[g], hi = goff[g + 1];
    double s = 0.0;
    for (int sl = 0; sl < slabs; ++sl) {
        const double *p = part + ((size_t)sl * n_rows + i) * cols + (size_t)a * E;
        for (long long e = lo + threadIdx.x; e < hi; e += 256) s += p[e];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double cnt = (double)(F - (k0 + i)) * (double)(hi - lo);
        rows[((size_t)i * G + g) * 4 + a] = cnt > 0.0 ? red[0] / cnt : 0.0;
    }
}

__global__ void lag_ends_total_kernel(double *__restrict__ rows, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[4 * i + 3] = (rows[4 * i] + rows[4 * i + 1]) + rows[4 * i + 2];
}

// The lags 1 .. k_lo and k_hi .. max_lag of `out` [max_lag + 1][G][4] (host or device memory) from the difference form.
// A synchronous call of its own (it runs inside the completion step of the spectral call, like the full fallback).
static int lag_exact_ends(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device, double scale,
                          int max_lag, int n_groups, const int64_t *group_off, long long k_lo, long long k_hi, double *out,
                          int out_on_device)
{
    CallScope cs(ctx);
    const long long F = n_frames, E = n_ent, G = n_groups, cols = 3 * E;
    const int kl = (int)k_lo, n_hi = (int)((long long)max_lag + 1 - k_hi);
    MD_REQUIRE(kl >= 0 && kl <= LAG_ENDS_MAX && n_hi >= 0 && n_hi <= LAG_ENDS_MAX, "internal: %d + %d lags to recompute", kl, n_hi);
    MD_HIP(hipSetDevice(ctx->device));
    int rc = MDHIP_OK;
    const size_t r_b = (size_t)F * 3 * E * 8;
    const double *d_r = (const double *)mdhip_stage(ctx, WS_XYZ_I, r, r_b, on_device, &rc);
    if (rc) return rc;
    const size_t part_b = (size_t)std::max(kl * LAG_ENDS_SLABS, n_hi) * cols * 8, go_b = (size_t)(G + 1) * 8;
    const size_t rows_b = (size_t)(kl + n_hi) * G * 4 * 8;
    MD_WS(d_part, double, WS_PART, part_b + 256);
    MD_WS(d_tab, unsigned char, WS_TABLES, go_b + rows_b + 64);
    long long *d_goff = reinterpret_cast<long long *>(d_tab);
    double *d_rows = reinterpret_cast<double *>(d_tab + go_b);
    MD_PIN(h_goff, unsigned char, go_b);
    memcpy(h_goff, group_off, go_b);
    rc = mdhip_copy_small(ctx, d_goff, h_goff, go_b, hipMemcpyHostToDevice);
    if (rc) return rc;
    KernelTimer timer(ctx);
    if (kl > 0) {
        hipLaunchKernelGGL(lag_low_lags_kernel, dim3((unsigned)((cols + 255) / 256), LAG_ENDS_SLABS), dim3(256), 0, ctx->stream, d_r, F,
                           cols, scale, kl, d_part);
        hipLaunchKernelGGL(lag_ends_fold_kernel, dim3((unsigned)kl, 3, (unsigned)G), dim3(256), 0, ctx->stream, d_part,
                           LAG_ENDS_SLABS, kl, cols, E, d_goff, (int)G, F, 1LL, d_rows);
    }
    if (n_hi > 0) {
        hipLaunchKernelGGL(lag_high_lags_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, d_r, F, cols, scale,
                           k_hi, n_hi, d_part);
        hipLaunchKernelGGL(lag_ends_fold_kernel, dim3((unsigned)n_hi, 3, (unsigned)G), dim3(256), 0, ctx->stream, d_part, 1, n_hi,
                           cols, E, d_goff, (int)G, F, k_hi, d_rows + (size_t)kl * G * 4);
    }
    hipLaunchKernelGGL(lag_ends_total_kernel, dim3((unsigned)(((kl + n_hi) * G + 255) / 256)), dim3(256), 0, ctx->stream, d_rows,
                       (long long)(kl + n_hi) * G);
    MD_HIP(hipGetLastError());
    timer.stop();
    const size_t row_b = (size_t)G * 4 * 8;
    if (kl > 0) {
        rc = mdhip_result(cs, out + (size_t)1 * G * 4, d_rows, kl * row_b, out_on_device);
        if (rc) return rc;
    }
    if (n_hi > 0) {
        rc = mdhip_result(cs, out + (size_t)k_hi * G * 4, d_rows + (size_t)kl * G * 4, n_hi * row_b, out_on_device);
        if (rc) return rc;
    }
    cs.defer([timer]() {
        timer.collect();
        return MDHIP_OK;
    });
    return cs.end();
}

// force_variant >= 0: that lag_variant instead of the context's option (the fallback of the spectral path)
static int lag_msd_impl(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                        double scale, int max_lag, int n_groups, const int64_t *group_off, double *out,
                        int out_on_device, int force_variant = -1)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    const int variant = force_variant >= 0 ? force_variant : ctx->opt_lag_variant;
    MD_REQUIRE(n_frames >= 0 && n_ent >= 0 && max_lag >= 0, "negative sizes");
    MD_REQUIRE(max_lag < n_frames || n_frames == 0, "max_lag must be < n_frames");
    MD_REQUIRE(out != nullptr, "out is NULL");
    std::vector<Chunk> chunks;
    std::vector<int> gco;
    const long long n_lags = (long long)max_lag + 1;
    // series-resident kernel: one series (+ 536 zeros so that the windows of the last tile stay inside) in LDS
    int lds_row = (int)((n_frames + 536 + 7) / 8) + 1;
    lds_row |= 1;  // odd row length: the 8 rows of the transposed layout start in different banks
    const size_t lds_b = (size_t)(9 * lds_row + 8) * 8;  // 8 lds_row entries + one pad double per 8
    const bool resident = variant != 0 && lds_b <= ctx->lds_max - 1024 && n_frames < (1 << 30);
    const int r_tiles = (int)((n_lags + 64 * LG_LPT - 1) / (64 * LG_LPT));
    const int r_pairs = (r_tiles + 1) / 2;
    const int r_gx = (r_pairs + 7) / 8;                 // blocks per entity chunk
    const int r_nw = (r_pairs + r_gx - 1) / r_gx;       // waves (tile pairs) per block, <= 8
    // entities per block: enough blocks to fill the chip (>= ~8 per CU), at most LG_ECHUNK
    int64_t echunk = LG_ECHUNK;
    if (resident) {
        // blocks that fit the chip at once (128 VGPRs: 4 waves per SIMD; LDS: one series per block), then the
        // smallest whole number of rounds R with <= LG_ECHUNK entities per block: blocks ~ R x capacity
        int64_t per_cu = lag_lds_blocks_per_cu(r_nw, lds_b);
        if (per_cu < 1) per_cu = 1;
        const int64_t capacity = std::max<int64_t>(1, per_cu * ctx->cu_count / r_gx);
        echunk = LG_ECHUNK;
        for (int64_t R = 1; R <= 64; ++R) {
            const int64_t ec = (n_ent + capacity * R - 1) / (capacity * R);
            if (ec <= LG_ECHUNK) {
                echunk = ec;
                break;
            }
        }
        if (echunk < 1) echunk = 1;
    } else {
        const int thr = n_lags <= 512 ? 64 : n_lags <= 1024 ? 128 : 256;
        const int64_t tile_pairs = ((n_lags + thr * LG_LPT - 1) / (thr * LG_LPT) + 1) / 2;
        const int64_t want = (int64_t)ctx->cu_count * 8;
        echunk = n_ent * tile_pairs / want;
        if (echunk < 1) echunk = 1;
        if (echunk > LG_ECHUNK) echunk = LG_ECHUNK;
    }
    int rc = build_chunks(ctx, n_ent, n_groups, group_off, chunks, gco, echunk);
    if (rc) return rc;
    MD_HIP(hipSetDevice(ctx->device));
    const size_t res_b = (size_t)n_lags * n_groups * 4 * 8;
    rc = mdhip_zero_result(ctx, out, res_b, out_on_device);
    if (rc) return rc;
    if (n_frames == 0 || chunks.empty()) return cs.end();
    MD_REQUIRE(r != nullptr, "r is NULL");
    MD_REQUIRE(chunks.size() <= 65535, "too many entity chunks (%zu)", chunks.size());
    const size_t r_b = (size_t)n_frames * 3 * n_ent * 8;
    const double *d_r = (const double *)mdhip_stage(ctx, WS_XYZ_I, r, r_b, on_device, &rc);
    if (rc) return rc;
    ctx->last_rel_bound = 0.0;
    ctx->lag_status_dev = nullptr;
    if (variant >= 2 && variant <= 4) {
        // the spectral path: the fused kernels finish on the device (round 5: prefix sums of the squares, S1 - 2 S2, division
        // by the counts in double-double arithmetic, msd_fft.hip) and the means are on their way to `out` when this
        // returns (round 6: the batched path for long series as well). The completion step decides
        // whether the bound is good enough — if not, the exact-difference kernel answers, from inside the step
        auto res = std::make_shared<LagFftResult>();
        rc = mdhip_lag_msd_fft(cs, n_frames, n_ent, d_r, scale, max_lag, n_groups, group_off, res, out, out_on_device);
        if (rc) return rc;
        cs.defer([=]() {
            ctx->last_rel_bound = res->bound;
            if (variant != 3 || res->bound <= 1e-10)
                return res->delivered ? MDHIP_OK : deliver_host_values(ctx, res->out.data(), res_b, out, out_on_device);
            // (round 6) too loose at a few lags of the two ends only: those from the difference form, the rest stands
            if (ctx->opt_lag_ends != 0 && res->ends_valid && res->delivered && res->k_lo <= LAG_ENDS_MAX &&
                (long long)max_lag + 1 - res->k_hi <= LAG_ENDS_MAX && res->bound_ok <= 1e-10) {
                const int rc2 = lag_exact_ends(ctx, n_frames, n_ent, r, on_device, scale, max_lag, n_groups, res->group_off.data(),
                                               res->k_lo, res->k_hi, out, out_on_device);
                ctx->last_rel_bound = res->bound_ok;
                ctx->last_kernel = "spectral + lag_low_lags_kernel (the lags that missed the bound from the difference form)";
                return rc2;
            }
            // bound too loose for this data (r and group_off are the caller's: valid until the call has completed)
            const int rc2 = lag_msd_impl(ctx, n_frames, n_ent, r, on_device, scale, max_lag, n_groups,
                                         res->group_off.data(), out, out_on_device, 1);
            ctx->last_rel_bound = 0.0;
            return rc2;
        });
        return cs.end();
    }
    MD_WS(d_x, double, WS_XYZ_J, r_b + 256);  // the scalar prefetch of the resident kernel reads <= 16 doubles past a series
    const int n_chunks = (int)chunks.size();
    const size_t ch_b = chunks.size() * sizeof(Chunk), go_b = (size_t)(n_groups + 1) * 8, gc_b = gco.size() * 4;
    const size_t tab_b = ch_b + go_b + gc_b;
    MD_WS(d_tab, unsigned char, WS_TABLES, tab_b + 64);
    MD_PIN(h_tab, unsigned char, tab_b);
    memcpy(h_tab, chunks.data(), ch_b);
    memcpy(h_tab + ch_b, group_off, go_b);
    memcpy(h_tab + ch_b + go_b, gco.data(), gc_b);
    rc = mdhip_copy_small(ctx, d_tab, h_tab, tab_b, hipMemcpyHostToDevice);
    if (rc) return rc;
    Chunk *d_chunks = reinterpret_cast<Chunk *>(d_tab);
    long long *d_goff = reinterpret_cast<long long *>(d_tab + ch_b);
    int *d_gco = reinterpret_cast<int *>(d_goff + n_groups + 1);
    MD_WS(d_partial, double, WS_PART, (size_t)n_chunks * n_lags * 4 * 8);
    const size_t out_b = (size_t)n_lags * n_groups * 4 * 8;
    double *d_out = out;
    if (!out_on_device) {
        d_out = (double *)mdhip_ws(ctx, WS_OUT, out_b);
        if (!d_out) return MDHIP_ENOMEM;
    }
    const long long cols = 3 * n_ent;
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((n_frames + 31) / 32)),
                       dim3(256), 0, ctx->stream, d_r, d_x, (long long)n_frames, cols, scale);
    MD_HIP(hipGetLastError());
    KernelTimer timer(ctx);
    if (resident) {
        const dim3 grid((unsigned)r_gx, (unsigned)n_chunks);
#define MD_LAG_CASE(NW)                                                                                          \
    case NW:                                                                                                     \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lag_msd_lds_kernel<NW>),                       \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));                     \
        hipLaunchKernelGGL(lag_msd_lds_kernel<NW>, grid, dim3(NW * 64), lds_b, ctx->stream, d_x,                 \
                           (long long)n_ent, (int)n_frames, (int)n_lags, d_chunks, r_tiles, lds_row, d_partial); \
        break;
        switch (r_nw) {
            MD_LAG_CASE(1)
            MD_LAG_CASE(2)
            MD_LAG_CASE(3)
            MD_LAG_CASE(4)
            MD_LAG_CASE(5)
            MD_LAG_CASE(6)
            MD_LAG_CASE(7)
            MD_LAG_CASE(8)
        }
#undef MD_LAG_CASE
        ctx->last_kernel = "lag_msd_lds_kernel";
    } else {
        const int threads = n_lags <= 512 ? 64 : n_lags <= 1024 ? 128 : 256;
        const int kt = threads * LG_LPT;
        const int n_tiles = (int)((n_lags + kt - 1) / kt);
        const dim3 grid((unsigned)((n_tiles + 1) / 2), (unsigned)n_chunks);
        if (threads == 64)
            hipLaunchKernelGGL(lag_msd_kernel<64>, grid, dim3(64), 0, ctx->stream, d_x, (long long)n_ent,
                               (long long)n_frames, n_lags, d_chunks, n_tiles, d_partial);
        else if (threads == 128)
            hipLaunchKernelGGL(lag_msd_kernel<128>, grid, dim3(128), 0, ctx->stream, d_x, (long long)n_ent,
                               (long long)n_frames, n_lags, d_chunks, n_tiles, d_partial);
        else
            hipLaunchKernelGGL(lag_msd_kernel<256>, grid, dim3(256), 0, ctx->stream, d_x, (long long)n_ent,
                               (long long)n_frames, n_lags, d_chunks, n_tiles, d_partial);
        ctx->last_kernel = "lag_msd_kernel";
    }
    timer.stop();
    MD_HIP(hipGetLastError());
    const long long tot = n_lags * n_groups * 4;
    hipLaunchKernelGGL(lag_msd_finish_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_partial, d_gco, d_goff, n_groups, (long long)n_frames, n_lags,
                       resident ? 1 : 0, d_out);
    MD_HIP(hipGetLastError());
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

int mdhip_lag_msd(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                  double scale, int max_lag, int n_groups, const int64_t *group_off, double *out)
{
    return lag_msd_impl(ctx, n_frames, n_ent, r, on_device, scale, max_lag, n_groups, group_off, out, 0);
}

int mdhip_lag_msd_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                      double scale, int max_lag, int n_groups, const int64_t *group_off, double *out_dev)
{
    return lag_msd_impl(ctx, n_frames, n_ent, r, on_device, scale, max_lag, n_groups, group_off, out_dev, 1);
}

/* ---- asynchronous twins: the work is queued on the context's stream, results are complete after mdhip_sync / mdhip_wait ---- */
int mdhip_msd_origin_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                           const double *origin, int origin_on_device, double scale, int n_groups,
                           const int64_t *group_off, double *sums, int sums_on_device, double *cols, int64_t col_stride,
                           int cols_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_msd_origin(ctx, n_frames, n_ent, r, on_device, origin, origin_on_device, scale, n_groups, group_off, sums,
                            sums_on_device, cols, col_stride, cols_on_device);
}

int mdhip_msd_pairs_dev_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                              double scale, int n_pairs, const int32_t *pairs, int n_groups, const int64_t *group_off,
                              double *sums_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_msd_pairs_dev(ctx, n_frames, n_ent, r, on_device, scale, n_pairs, pairs, n_groups, group_off, sums_dev);
}

int mdhip_msd_windows_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device,
                            double scale, int tao, double *win_sums, int out_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return msd_windows_impl(ctx, n_frames, n_ent, r, on_device, scale, tao, win_sums, out_on_device ? 1 : 0);
}

int mdhip_lag_msd_status_dev(mdhip_ctx *ctx, double *dst_dev)
{
    if (!ctx || !dst_dev) return MDHIP_EINVAL;
    MD_HIP(hipSetDevice(ctx->device));
    if (ctx->lag_status_dev)
        return mdhip_copy_small(ctx, dst_dev, ctx->lag_status_dev, 8, hipMemcpyDeviceToDevice);
    else
        MD_HIP(hipMemsetAsync(dst_dev, 0, 8, ctx->stream));
    return MDHIP_OK;
}

int mdhip_lag_msd_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_ent, const double *r, int on_device, double scale,
                        int max_lag, int n_groups, const int64_t *group_off, double *out, int out_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return lag_msd_impl(ctx, n_frames, n_ent, r, on_device, scale, max_lag, n_groups, group_off, out,
                        out_on_device ? 1 : 0);
}

}  // extern "C"
