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
//  1. shell_pairs_kernel: all central x shell atoms of every frame, the same exact single-wrap rsq as the pair
//     histograms (rdf_cn.py:44-57, contraction off), the smaller set staged through LDS. Every hit sets bit t of the
//     PRESENCE MASK of its pair (i, j) — a row of ceil(F / 64) words in a hash table keyed by i * n_j + j (open
//     addressing, linear probing; a slot is claimed with one compare-and-swap, a bit set with one atomic OR).
//     Round 6: rounds 1-5 appended a record per hit and brought a pair's frames together with a 64-bit radix sort of
//     the records (hipCUB — the one vendor-library call of the product); only the GROUPING was ever needed — the
//     masks are sets — and the table gives it without a sort, a record list or a run search.
//  2. residence_lag_kernel: a wave per occupied slot copies the pair's mask to LDS and adds
//     popcount(mask & (mask >> k)) to its lag table for every lag k up to the pair's span; tables are merged with
//     64-bit global atomics (integers: order-independent, so is the table's slot order).

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

constexpr unsigned long long RT_EMPTY = ~0ull;  // (a pair key is < 2^63)
constexpr unsigned RT_NONE = ~0u;

__device__ __forceinline__ unsigned long long rt_mix(unsigned long long x)
{
    // splitmix64's finaliser: consecutive keys (neighbouring j of one i) land far apart
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// the slot of `key` in the table (claimed if new); RT_NONE when `max_probe` slots in a row were other pairs' (the table
// is too small: the host sweeps again with one sized from the hit count)
__device__ __forceinline__ unsigned rt_slot(unsigned long long *__restrict__ keys, unsigned slot_mask, unsigned max_probe,
                                            unsigned long long key)
{
    unsigned h = (unsigned)rt_mix(key) & slot_mask;
    for (unsigned probe = 0; probe < max_probe; ++probe) {
        const unsigned long long seen = atomicCAS(&keys[h], RT_EMPTY, key);
        if (seen == RT_EMPTY || seen == key) return h;
        h = (h + 1u) & slot_mask;
    }
    return RT_NONE;
}

// grid (ceil(n_lane / 256), F): a lane holds one atom of the LARGER set (round 6: with the central atoms always on the
// lanes, 315 Mg against 11 280 O filled 315 of 512 lanes), the atoms of the other set are staged through LDS 256 at a time
// and read as broadcasts. SWAP: the lanes hold the shell atoms j, the loop walks the central atoms i (the pair key is
// i * n_j + j either way). stat[0] += hits, stat[1] = 1 when a hit found no slot.
template <bool SWAP>
__global__ __launch_bounds__(RT_TILE) void shell_pairs_kernel(
    const double *__restrict__ xi, long long n_i, const double *__restrict__ xj, long long n_j,
    const double *__restrict__ box, double lo2, double hi2, int exclude_diagonal, int words,
    unsigned long long *__restrict__ keys, unsigned long long *__restrict__ masks, unsigned slot_mask, unsigned max_probe,
    unsigned long long *__restrict__ stat)
{
    __shared__ double s_b[3][RT_TILE];
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
    const unsigned long long bit = 1ull << (f & 63);
    const size_t word = (size_t)(f >> 6);
    unsigned long long mine = 0;
    bool lost = false;
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
                    ++mine;
                    const unsigned slot = rt_slot(keys, slot_mask, max_probe, (unsigned long long)i * (unsigned long long)n_j + (unsigned long long)j);
                    if (slot == RT_NONE) lost = true;
                    else atomicOr(&masks[(size_t)slot * (size_t)words + word], bit);
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((tid & 63) == 0 && mine) atomicAdd(stat, mine);
    if (lost) atomicOr(stat + 1, 1ull);
}

// One wave per occupied slot (grid-stride over the table). LDS: presence mask [words + 1] + lag table [n_frames] (u64).
__global__ __launch_bounds__(64) void residence_lag_kernel(
    const unsigned long long *__restrict__ keys, const unsigned long long *__restrict__ masks, unsigned long long n_slots,
    int n_frames, int words, unsigned long long *__restrict__ counts)
{
    extern __shared__ unsigned long long s_mem[];
    unsigned long long *mask = s_mem;               // [words + 1] (one zero word behind the end)
    unsigned long long *table = s_mem + words + 1;  // [n_frames]
    const int lane = threadIdx.x;
    for (int k = lane; k < n_frames; k += 64) table[k] = 0ull;
    if (lane == 0) mask[words] = 0ull;
    for (unsigned long long r = blockIdx.x; r < n_slots; r += gridDim.x) {
        if (keys[r] == RT_EMPTY) continue;  // (wave-uniform)
        __syncthreads();  // (the previous pair's lags have read the mask)
        int t_first = n_frames, t_last = -1;
        for (int w = lane; w < words; w += 64) {
            const unsigned long long m = masks[(size_t)r * (size_t)words + w];
            mask[w] = m;
            if (m) {
                const int lo = 64 * w + __builtin_ctzll(m), hi = 64 * w + 63 - __builtin_clzll(m);
                t_first = lo < t_first ? lo : t_first;
                t_last = hi > t_last ? hi : t_last;
            }
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
    }
    __syncthreads();
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
    MD_REQUIRE((double)n_i * (double)n_j < 9.0e18, "pair key does not fit 63 bits");
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
    MD_WS(d_counts, unsigned long long, WS_OUT, (size_t)n_frames * 8);
    MD_HIP(hipMemsetAsync(d_counts, 0, (size_t)n_frames * 8, ctx->stream));
    MD_PIN(h_out, unsigned long long, ((size_t)n_frames + 8) * 8);

    const bool swap = n_j > n_i;  // the larger set on the lanes
    const dim3 grid((unsigned)(((swap ? n_j : n_i) + RT_TILE - 1) / RT_TILE), (unsigned)n_frames);
    KernelTimer timer(ctx);
    ctx->last_kernel = "shell_pairs_kernel";
    // The table: one slot per PAIR that is ever in the shell (key 8 B + a mask of `words` words), at most half full.
    // ONE sweep in the common case: the slot count comes from the expected number of HITS (the shell's share of the box x
    // pairs x frames x 1.5 + slack: far more than distinct pairs — a pair stays for many frames); only a call whose table
    // fills up (a clustered system) sweeps again, with slots from the hit count the first sweep returned (>= its pairs).
    const size_t slot_b = 8 + (size_t)words * 8;
    const size_t mem_cap = (size_t)12 << 30;  // (the table's bytes; beyond that the call fails cleanly)
    auto pow2_at_least = [](double v, unsigned long long lo = 1024) {
        unsigned long long p = lo;
        while ((double)p < v && p < (1ull << 31)) p <<= 1;
        return p;
    };
    unsigned long long slots;
    {
        const double pi43 = 4.18879020478639;
        const double vol = box[0] * box[1] * box[2];
        const double r_hi = std::sqrt(std::max(r_hi_sq, 0.0)), r_lo = std::sqrt(std::max(r_lo_sq, 0.0));
        const double share = vol > 0.0 ? std::min(1.0, pi43 * (r_hi * r_hi * r_hi - r_lo * r_lo * r_lo) / vol) : 1.0;
        const double est = (double)n_i * (double)n_j * (double)n_frames * share * 1.5 + 262144.0;
        slots = pow2_at_least(2.0 * std::min(est, (double)n_i * (double)n_j));  // (never more pairs than there are)
        if (ctx->opt_residence_cap > 0) slots = pow2_at_least((double)ctx->opt_residence_cap, 4);  // (tests: force the re-sweep)
        while (slots > 1024 && slots * slot_b > mem_cap) slots >>= 1;
    }
    unsigned long long n_hit = 0;
    unsigned long long *d_keys = nullptr, *d_masks = nullptr;
    for (int sweep = 0; sweep < 2; ++sweep) {
        unsigned char *d_tab = (unsigned char *)mdhip_ws(ctx, WS_AUX0, (size_t)slots * slot_b);
        if (!d_tab) return MDHIP_ENOMEM;
        d_keys = reinterpret_cast<unsigned long long *>(d_tab);
        d_masks = d_keys + slots;
        MD_HIP(hipMemsetAsync(d_keys, 0xFF, (size_t)slots * 8, ctx->stream));
        MD_HIP(hipMemsetAsync(d_masks, 0, (size_t)slots * (size_t)words * 8, ctx->stream));
        MD_HIP(hipMemsetAsync(d_misc, 0, 64, ctx->stream));
        // (the second sweep's table holds every pair at half load: its probes may walk as far as they must)
        const unsigned max_probe = sweep == 0 ? 512u : (unsigned)std::min<unsigned long long>(slots, 0xFFFFFFFFull);
        if (swap)
            hipLaunchKernelGGL(shell_pairs_kernel<true>, grid, dim3(RT_TILE), 0, ctx->stream, d_xi, (long long)n_i, d_xj,
                               (long long)n_j, d_box, r_lo_sq, r_hi_sq, exclude_diagonal, words, d_keys, d_masks,
                               (unsigned)(slots - 1), max_probe, d_misc);
        else
            hipLaunchKernelGGL(shell_pairs_kernel<false>, grid, dim3(RT_TILE), 0, ctx->stream, d_xi, (long long)n_i, d_xj,
                               (long long)n_j, d_box, r_lo_sq, r_hi_sq, exclude_diagonal, words, d_keys, d_masks,
                               (unsigned)(slots - 1), max_probe, d_misc);
        MD_HIP(hipGetLastError());
        MD_HIP(hipMemcpyAsync(h_out, d_misc, 16, hipMemcpyDeviceToHost, ctx->stream));
        MD_HIP(mdhip_stream_wait(ctx));
        n_hit = h_out[0];
        if (h_out[1] == 0) break;
        MD_REQUIRE(sweep == 0, "residence: the pair table filled up although it was sized from the hit count");
        slots = pow2_at_least(2.0 * (double)n_hit);
        if (slots * slot_b > mem_cap)
            return mdhip_fail(ctx, MDHIP_ELIMIT, "residence: %llu in-shell records over %lld frames need a pair table of %.1f GB",
                              n_hit, (long long)n_frames, (double)(slots * slot_b) / 1073741824.0);
    }
    if (n_records) *n_records = n_hit;
    if (n_hit > 0) {
        if (lds_b > 65536)
            MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(residence_lag_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
        const unsigned lag_grid = (unsigned)std::min<unsigned long long>(slots, (unsigned long long)ctx->cu_count * 16);
        hipLaunchKernelGGL(residence_lag_kernel, dim3(lag_grid), dim3(64), lds_b, ctx->stream, d_keys, d_masks, slots,
                           (int)n_frames, words, d_counts);
        MD_HIP(hipGetLastError());
    }
    timer.stop();
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
