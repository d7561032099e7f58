// residence.hip — neighbour-shell residence autocorrelation.
//
// Replaces the two loops of dynamical/residence_time.py:70-146 of the reference
// (ResidenceTime.calc_auto_correlation): per frame an indicator h_ij(t) = 1 when atom j sits in the shell
// (lo, hi] around central atom i, then the mean over all (i, j) of the unbiased autocovariance of h_ij.
// The numerators are integers,
//
//     counts[k] = sum_{i,j} sum_t h_ij(t) h_ij(t+k)
//
// and are computed exactly; the reference's FFT estimator returns them with ~1e-16 relative noise.
//
//  1. shell_pairs_kernel (twice: count, then fill): all central x shell atoms of every frame, the same
//     exact single-wrap rsq as the pair histograms (rdf_cn.py:44-57, contraction off), shell atoms staged
//     through LDS. Every hit appends one record (pair key << frame bits | frame).
//  2. a 64-bit radix sort of the records (hipCUB) brings each pair's frames together, in time order.
//  3. residence_lag_kernel: a wave per run of equal pair keys builds the pair's presence bit mask over the
//     frames in LDS and adds popcount(mask & (mask >> k)) to its lag table for every lag k up to the run's
//     span; tables are merged with 64-bit global atomics (integers: order-independent).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "ctx.h"

#pragma clang fp contract(off)

namespace {

constexpr int RT_TILE = 256;

__device__ __forceinline__ double rt_wrap_abs(double d, double L)
{
    // |d - sign(d) L| if |d| > L/2 else |d|, as min(|d|, ||d| - L|): see pair_hist.hip for the equivalence
    const double a = __builtin_fabs(d);
    return __builtin_fmin(a, __builtin_fabs(a - L));
}

// grid (ceil(n_lane / 256), F): a lane holds one atom of the LARGER set (round 6: with the central atoms always on the
// lanes, 315 Mg against 11 280 O filled 315 of 512 lanes), the atoms of the other set are staged through LDS 256 at a time
// and read as broadcasts. SWAP: the lanes hold the shell atoms j, the loop walks the central atoms i (the record key is
// i * n_j + j either way). Records are appended to the list; a block collects its hits in LDS and reserves space in the
// global list once per staged tile (one global atomic per flush instead of one per hit on a single counter, which
// serialises: 10^7 hits took 100 ms that way). *n_rec counts every hit, also those beyond `cap` (the host sweeps again
// with the exact size then).
constexpr int RT_STAGE = 2048;
template <bool SWAP>
__global__ __launch_bounds__(RT_TILE) void shell_pairs_kernel(
    const double *__restrict__ xi, long long n_i, const double *__restrict__ xj, long long n_j,
    const double *__restrict__ box, double lo2, double hi2, int exclude_diagonal, int frame_bits,
    unsigned long long *__restrict__ n_rec, unsigned long long *__restrict__ rec, unsigned long long cap)
{
    __shared__ double s_b[3][RT_TILE];
    __shared__ unsigned long long s_rec[RT_STAGE];
    __shared__ unsigned s_n;
    __shared__ unsigned long long s_base;
    const int f = blockIdx.y, tid = threadIdx.x;
    // lane set a, looped set b
    const long long n_a = SWAP ? n_j : n_i, n_b = SWAP ? n_i : n_j;
    const double *pa = (SWAP ? xj : xi) + (size_t)f * 3 * n_a, *pb = (SWAP ? xi : xj) + (size_t)f * 3 * n_b;
    const long long la = (long long)blockIdx.x * RT_TILE + tid;
    const double Lx = box[3 * f], Ly = box[3 * f + 1], Lz = box[3 * f + 2];
    double x = 0.0, y = 0.0, z = 0.0;
    if (la < n_a) {
        x = pa[la];
        y = pa[n_a + la];
        z = pa[2 * n_a + la];
    }
    if (tid == 0) s_n = 0u;
    for (long long b0 = 0; b0 < n_b; b0 += RT_TILE) {
        __syncthreads();
        const long long bl = b0 + tid;
        s_b[0][tid] = bl < n_b ? pb[bl] : 0.0;
        s_b[1][tid] = bl < n_b ? pb[n_b + bl] : 0.0;
        s_b[2][tid] = bl < n_b ? pb[2 * n_b + bl] : 0.0;
        __syncthreads();
        const int cnt = (int)((n_b - b0) < RT_TILE ? (n_b - b0) : RT_TILE);
        if (la < n_a) {
            for (int bb = 0; bb < cnt; ++bb) {
                // head - other (rdf_cn.py:44-57): the central atom is the head row; |d| is what enters, so the order of
                // the subtraction is immaterial to the bits (rounding is sign-symmetric)
                const double ax = rt_wrap_abs(x - s_b[0][bb], Lx);
                const double ay = rt_wrap_abs(y - s_b[1][bb], Ly);
                const double az = rt_wrap_abs(z - s_b[2][bb], Lz);
                const double rsq = (ax * ax + ay * ay) + az * az;
                const long long i = SWAP ? b0 + bb : la, j = SWAP ? la : b0 + bb;
                if (rsq > lo2 && rsq <= hi2 && !(exclude_diagonal && j == i)) {  // residence_time.py:102-104
                    const unsigned long long r =
                        (((unsigned long long)i * (unsigned long long)n_j + (unsigned long long)j) << frame_bits) |
                        (unsigned long long)f;
                    const unsigned k = atomicAdd(&s_n, 1u);
                    if (k < (unsigned)RT_STAGE) {
                        s_rec[k] = r;
                    } else {  // stage full (a very dense shell): straight to the global list
                        const unsigned long long pos = atomicAdd(n_rec, 1ull);
                        if (pos < cap) rec[pos] = r;
                    }
                }
            }
        }
        // flush the stage
        __syncthreads();
        const unsigned staged = s_n < (unsigned)RT_STAGE ? s_n : (unsigned)RT_STAGE;
        if (tid == 0 && staged) s_base = atomicAdd(n_rec, (unsigned long long)staged);
        __syncthreads();
        for (unsigned k = tid; k < staged; k += RT_TILE)
            if (s_base + k < cap) rec[s_base + k] = s_rec[k];
        __syncthreads();
        if (tid == 0) s_n = 0u;
    }
}

// starts[] = indices where a new pair key begins in the sorted records (any order); a block reserves its
// slots with ONE global atomic
__global__ __launch_bounds__(256) void run_starts_kernel(const unsigned long long *__restrict__ rec,
                                                         unsigned long long n, int frame_bits,
                                                         unsigned long long *__restrict__ n_runs,
                                                         unsigned long long *__restrict__ starts)
{
    __shared__ unsigned s_wave[4];
    __shared__ unsigned long long s_base;
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool is_start = k < n && (k == 0 || (rec[k] >> frame_bits) != (rec[k - 1] >> frame_bits));
    const unsigned long long m = __builtin_amdgcn_ballot_w64(is_start);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wave[wave] = (unsigned)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = tot ? atomicAdd(n_runs, (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (is_start) {
        unsigned off = (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        starts[s_base + off] = k;
    }
}

// One wave per run (grid-stride over the runs). LDS: presence mask [words] + lag table [n_frames] (u64).
__global__ __launch_bounds__(64) void residence_lag_kernel(
    const unsigned long long *__restrict__ rec, unsigned long long n, int frame_bits,
    const unsigned long long *__restrict__ starts, unsigned long long n_runs, int n_frames, int words,
    unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned long long s_mem[];
    unsigned long long *mask = s_mem;            // [words + 1] (one zero word behind the end)
    unsigned long long *table = s_mem + words + 1;  // [n_frames]
    const int lane = threadIdx.x;
    const unsigned long long fmask = (1ull << frame_bits) - 1ull;
    for (int k = lane; k < n_frames; k += 64) table[k] = 0ull;
    for (unsigned long long r = blockIdx.x; r < n_runs; r += gridDim.x) {
        for (int w = lane; w <= words; w += 64) mask[w] = 0ull;
        __syncthreads();
        const unsigned long long s0 = starts[r];
        const unsigned long long key = rec[s0] >> frame_bits;
        // records of a run are sorted by frame: walk them 64 at a time
        int t_first = n_frames, t_last = -1;
        for (unsigned long long p = s0 + lane;; p += 64) {
            const bool in = p < n && (rec[p] >> frame_bits) == key;
            if (in) {
                const int t = (int)(rec[p] & fmask);
                atomicOr(&mask[t >> 6], 1ull << (t & 63));
                t_first = t < t_first ? t : t_first;
                t_last = t > t_last ? t : t_last;
            }
            if (!__builtin_amdgcn_ballot_w64(in) || __builtin_amdgcn_ballot_w64(!in)) break;  // run ended in this batch
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int a = __shfl_xor(t_first, off, 64), b = __shfl_xor(t_last, off, 64);
            t_first = a < t_first ? a : t_first;
            t_last = b > t_last ? b : t_last;
        }
        __syncthreads();
        const int span = t_last - t_first;  // lags beyond the span see no overlap
        const int w0 = t_first >> 6, w1 = t_last >> 6;
        for (int lag = lane; lag <= span; lag += 64) {
            const int q = lag >> 6, sh = lag & 63;
            unsigned long long c = 0;
            for (int w = w0; w + q <= w1; ++w) {
                const unsigned long long lo = mask[w + q], hi = mask[w + q + 1];
                const unsigned long long shifted = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                c += (unsigned long long)__builtin_popcountll(mask[w] & shifted);
            }
            table[lag] += c;  // a lag belongs to one lane: no conflict
        }
        __syncthreads();
    }
    for (int k = lane; k < n_frames; k += 64)
        if (table[k]) atomicAdd(&counts[k], table[k]);
}

}  // namespace

extern "C" {

int mdhip_shell_residence(mdhip_ctx *ctx, int64_t n_frames, int64_t n_i, const double *xi, int xi_on_device,
                          int64_t n_j, const double *xj, int xj_on_device, const double *box, double r_lo_sq,
                          double r_hi_sq, int exclude_diagonal, uint64_t *counts, uint64_t *n_records)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);  // (synchronous throughout: the record and run counts steer the launches that follow)
    MD_REQUIRE(n_frames >= 0 && n_i >= 0 && n_j >= 0, "negative sizes");
    MD_REQUIRE(n_frames == 0 || counts, "counts is NULL");
    MD_REQUIRE(!exclude_diagonal || n_i == n_j, "exclude_diagonal needs identical sets");
    if (n_records) *n_records = 0;
    std::fill(counts, counts + n_frames, (uint64_t)0);
    if (n_frames == 0 || n_i == 0 || n_j == 0) return cs.end();
    MD_REQUIRE(xi && xj && box, "NULL input array");
    int frame_bits = 1;
    while ((1LL << frame_bits) < n_frames) ++frame_bits;
    MD_REQUIRE(frame_bits <= 24, "at most 2^24 frames");
    MD_REQUIRE((double)n_i * (double)n_j < (double)(1ull << (63 - frame_bits)), "pair key does not fit 64 bits");
    MD_REQUIRE(n_frames <= 65535, "at most 65535 frames per call");
    const int words = (int)((n_frames + 63) / 64);
    const size_t lds_b = ((size_t)words + 1 + (size_t)n_frames) * 8;
    MD_REQUIRE(lds_b <= ctx->lds_max - 512, "%lld frames exceed the LDS lag table", (long long)n_frames);
    MD_HIP(hipSetDevice(ctx->device));
    int rc;
    const double *d_xi = (const double *)mdhip_stage(ctx, WS_XYZ_I, xi, (size_t)n_frames * 3 * n_i * 8, xi_on_device, &rc);
    if (rc) return rc;
    const double *d_xj = d_xi;
    if (xj != xi || xj_on_device != xi_on_device) {
        d_xj = (const double *)mdhip_stage(ctx, WS_XYZ_J, xj, (size_t)n_frames * 3 * n_j * 8, xj_on_device, &rc);
        if (rc) return rc;
    }
    MD_WS(d_box, double, WS_BOX, (size_t)n_frames * 3 * 8);
    MD_PIN(h_box, double, (size_t)n_frames * 3 * 8);
    memcpy(h_box, box, (size_t)n_frames * 3 * 8);
    MD_HIP(hipMemcpyAsync(d_box, h_box, (size_t)n_frames * 3 * 8, hipMemcpyHostToDevice, ctx->stream));
    MD_WS(d_misc, unsigned long long, WS_MISC, 64);
    MD_HIP(hipMemsetAsync(d_misc, 0, 64, ctx->stream));
    MD_WS(d_counts, unsigned long long, WS_OUT, (size_t)n_frames * 8);
    MD_HIP(hipMemsetAsync(d_counts, 0, (size_t)n_frames * 8, ctx->stream));
    MD_PIN(h_out, unsigned long long, ((size_t)n_frames + 8) * 8);

    const bool swap = n_j > n_i;  // the larger set on the lanes
    const dim3 grid((unsigned)(((swap ? n_j : n_i) + RT_TILE - 1) / RT_TILE), (unsigned)n_frames);
    KernelTimer timer(ctx);
    ctx->last_kernel = "shell_pairs_kernel";
    // ONE sweep over the pairs in the common case (round 6; rounds 1-5 swept twice, count then fill): the record list is
    // sized from the shell's share of the box — expected records x 1.5 + slack — and the fill sweep counts every hit
    // whether it fits or not; only a call whose shells are denser than that (a clustered system) sweeps again with
    // the exact size. The sweep is the call's cost (n_i x n_j x F exact f64 distance chains), the list a few MB.
    unsigned long long cap;
    {
        const double pi43 = 4.18879020478639;
        const double vol = box[0] * box[1] * box[2];
        const double r_hi = std::sqrt(std::max(r_hi_sq, 0.0)), r_lo = std::sqrt(std::max(r_lo_sq, 0.0));
        const double share = vol > 0.0 ? std::min(1.0, pi43 * (r_hi * r_hi * r_hi - r_lo * r_lo * r_lo) / vol) : 1.0;
        const double est = (double)n_i * (double)n_j * (double)n_frames * share * 1.5 + 262144.0;
        cap = (unsigned long long)std::min(est, 2.5e8);  // (<= 4 GB for the list and its sorted copy)
        if (ctx->opt_residence_cap > 0) cap = (unsigned long long)ctx->opt_residence_cap;  // (tests: force the second sweep)
    }
    unsigned long long n_rec = 0;
    unsigned long long *d_rec = nullptr;
    for (int sweep = 0; sweep < 2; ++sweep) {
        d_rec = (unsigned long long *)mdhip_ws(ctx, WS_AUX0, (size_t)cap * 8);
        if (!d_rec) return MDHIP_ENOMEM;
        if (swap)
            hipLaunchKernelGGL(shell_pairs_kernel<true>, grid, dim3(RT_TILE), 0, ctx->stream, d_xi, (long long)n_i, d_xj,
                               (long long)n_j, d_box, r_lo_sq, r_hi_sq, exclude_diagonal, frame_bits, d_misc + sweep, d_rec, cap);
        else
            hipLaunchKernelGGL(shell_pairs_kernel<false>, grid, dim3(RT_TILE), 0, ctx->stream, d_xi, (long long)n_i, d_xj,
                               (long long)n_j, d_box, r_lo_sq, r_hi_sq, exclude_diagonal, frame_bits, d_misc + sweep, d_rec, cap);
        MD_HIP(hipGetLastError());
        MD_HIP(hipMemcpyAsync(h_out, d_misc + sweep, 8, hipMemcpyDeviceToHost, ctx->stream));
        MD_HIP(mdhip_stream_wait(ctx));
        n_rec = h_out[0];
        if (n_rec <= cap) break;
        MD_REQUIRE(sweep == 0, "residence: the record count changed between two sweeps of the same frames");
        cap = n_rec;  // exact now
    }
    if (n_records) *n_records = n_rec;
    if (n_rec == 0) {
        timer.stop();
        MD_HIP(mdhip_stream_wait(ctx));
        const double ms0 = timer.collect();
        cs.defer([ctx, ms0]() {
            ctx->last_ms = ms0;
            return MDHIP_OK;
        });
        return cs.end();
    }
    MD_WS(d_srt, unsigned long long, WS_AUX1, (size_t)n_rec * 8);
    size_t tmp_b = 0;
    MD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_b, d_rec, d_srt, (size_t)n_rec, 0, 64, ctx->stream));
    MD_WS(d_tmp, unsigned char, WS_AUX2, tmp_b + 256);
    MD_HIP(hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_b, d_rec, d_srt, (size_t)n_rec, 0, 64, ctx->stream));
    // run starts (unordered list) reuse the unsorted buffer
    unsigned long long *d_starts = d_rec;
    hipLaunchKernelGGL(run_starts_kernel, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, ctx->stream, d_srt,
                       n_rec, frame_bits, d_misc + 2, d_starts);
    MD_HIP(hipGetLastError());
    MD_HIP(hipMemcpyAsync(h_out, d_misc + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
    MD_HIP(mdhip_stream_wait(ctx));
    const unsigned long long n_runs = h_out[0];
    if (lds_b > 65536)
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(residence_lag_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
    const unsigned lag_grid = (unsigned)std::min<unsigned long long>(n_runs, (unsigned long long)ctx->cu_count * 16);
    hipLaunchKernelGGL(residence_lag_kernel, dim3(lag_grid), dim3(64), lds_b, ctx->stream, d_srt, n_rec, frame_bits,
                       d_starts, n_runs, (int)n_frames, words, d_counts);
    timer.stop();
    MD_HIP(hipGetLastError());
    MD_HIP(hipMemcpyAsync(h_out, d_counts, (size_t)n_frames * 8, hipMemcpyDeviceToHost, ctx->stream));
    MD_HIP(mdhip_stream_wait(ctx));
    const double ms = timer.collect();
    memcpy(counts, h_out, (size_t)n_frames * 8);
    cs.defer([ctx, ms]() {
        ctx->last_ms = ms;
        return MDHIP_OK;
    });
    return cs.end();
}

}  // extern "C"
