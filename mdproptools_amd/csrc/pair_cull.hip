// pair_cull.hip — pre-pass of the spatially culled pair sweep: Hilbert sort of every frame, bounding boxes,
// neighbour-tile lists (SURVEY.md 8f "cell-list variant").
#include "pair_common.h"

#ifndef SORT_EXP
#define SORT_EXP 0  // timing experiments of cull_sort_reg_kernel: 1 no division, 2 no Hilbert arithmetic (a plain key)
#endif
namespace mdpair {
namespace {

// ------------------------------------------------------------------------------------------------
// Spatial culling for r_cut << L (SURVEY.md §8f "cell-list variant"): atoms are re-ordered along a Hilbert
// curve over a 32^3 grid of the periodic cell (laid from the frame's smallest coordinates), so that every
// tile of 256 consecutive atoms is a compact blob; a tile pair whose axis-aligned bounding boxes are
// farther apart than the cutoff cannot contain an in-cutoff pair and is never swept. The pair kernel's
// arithmetic is unchanged — the SAME exact rsq decides every pair that is swept — so the integer
// histograms are identical to the dense path's. Conservative by construction:
//  * boxes hold the coordinates AS GIVEN (no wrapping). The reference's per-axis distance after its single
//    wrap is min(|d|, ||d| - L|) = dist(d, {0, +L, -L}); over all d = a - b with a, b in two boxes its
//    minimum is the gap between the d interval and the nearest of those three points (interval_gap / gapf),
//    for any coordinates, inside the cell or box lengths away from it;
//  * f32 boxes are widened outward beyond their rounding; the tile test carries its own slack.
// The order of atoms inside a cell depends on atomic arrival order; only sums of integers depend on it.
// ------------------------------------------------------------------------------------------------


__device__ __forceinline__ double wrapped_frac(double x, double L)
{
    const double s = x / L;
    double f = s - __builtin_floor(s);
    return f < 1.0 ? f : 0.0;
}

// Hilbert index of a cell on the 32^3 grid (Skilling's transpose algorithm): consecutive indices are
// face-adjacent cells, so a run of consecutive atoms is a compact blob (a Morton run can jump across the
// box; measured on a uniform 100k-atom frame, 1.3x more tile pairs survive the culling with Morton order).
__device__ __forceinline__ unsigned hilbert3(unsigned cx, unsigned cy, unsigned cz)
{
    unsigned X[3] = {cx, cy, cz};
    const unsigned M = 1u << (MORTON_BITS - 1);
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const unsigned t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t;
    X[1] ^= t;
    X[2] ^= t;
    unsigned key = 0;
    for (int b = MORTON_BITS - 1; b >= 0; --b)
#pragma unroll
        for (int i = 0; i < 3; ++i) key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
}

// origin[f][3] ~ smallest x, y, z of the frame, from 1024 atoms spread over the id range (one block per
// frame). The grid of the spatial sort is laid from there, so that a cell [lo, lo+L) with any lo is cut at
// its own faces and not somewhere inside. Only the quality of the sort depends on it (an origin a little
// inside the cell sends a thin slice of atoms to the far end of the curve), never a result.
__global__ __launch_bounds__(256) void cull_origin_kernel(const double *__restrict__ xyz, long long n,
                                                          double *__restrict__ origin)
{
    __shared__ double red[3][4];
    const int f = blockIdx.x;
    const double *x = xyz + (size_t)f * 3 * n;
    double lo[3] = {1e300, 1e300, 1e300};
    const long long stride = n > 1024 ? n / 1024 : 1;
    for (int k = threadIdx.x; k < 1024; k += 256) {
        const long long i = (long long)k * stride;
        if (i < n)
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) lo[ax] = __builtin_fmin(lo[ax], x[ax * n + i]);
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lo[ax] = __builtin_fmin(lo[ax], __shfl_down(lo[ax], off, 64));
        if ((threadIdx.x & 63) == 0) red[ax][threadIdx.x >> 6] = lo[ax];
    }
    __syncthreads();
    if (threadIdx.x < 3)
        origin[3 * f + threadIdx.x] = __builtin_fmin(__builtin_fmin(red[threadIdx.x][0], red[threadIdx.x][1]),
                                                     __builtin_fmin(red[threadIdx.x][2], red[threadIdx.x][3]));
}

// keys[f][n] and cell populations cells[f][key]
__global__ void cull_keys_kernel(const double *__restrict__ xyz, const double *__restrict__ box, long long n,
                                 const double *__restrict__ origin,
                                 unsigned short *__restrict__ keys, unsigned *__restrict__ cells)
{
    const int f = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *x = xyz + (size_t)f * 3 * n;
    const double G = (double)(1 << MORTON_BITS);
    unsigned c[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        // the origin is a sampled minimum: an atom a little below it belongs to the first cell, not (wrapped)
        // to the last one, where it would blow up the bounding boxes of the tiles it lands in
        double d = x[ax * n + i] - origin[3 * f + ax];
        if (d < 0.0 && d >= -box[3 * f + ax] * (1.0 / 64.0)) d = 0.0;
        int v = (int)(wrapped_frac(d, box[3 * f + ax]) * G);
        c[ax] = (unsigned)(v < 0 ? 0 : v > (1 << MORTON_BITS) - 1 ? (1 << MORTON_BITS) - 1 : v);
    }
    const unsigned key = hilbert3(c[0], c[1], c[2]);
    keys[(size_t)f * n + i] = (unsigned short)key;
    atomicAdd(&cells[(size_t)f * MORTON_CELLS + key], 1u);
}

// exclusive scan of the 32768 cell populations of one frame (one block per frame)
__global__ __launch_bounds__(256) void cull_scan_kernel(unsigned *__restrict__ cells)
{
    __shared__ unsigned part[256];
    unsigned *c = cells + (size_t)blockIdx.x * MORTON_CELLS;
    constexpr int PER = MORTON_CELLS / 256;
    const int base = threadIdx.x * PER;
    unsigned sum = 0;
    for (int k = 0; k < PER; ++k) sum += c[base + k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned add = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (int k = 0; k < PER; ++k) {
        const unsigned v = c[base + k];
        c[base + k] = run;
        run += v;
    }
}

// Whole counting sort of one frame in ONE block (many-frame workloads: as many blocks as frames): exact
// grid origin, keys, cell populations, scan and scatter with the 32768 cell counters in LDS — no global
// atomics (device-scope atomics on counters spread over HBM cost several times what the arithmetic costs).
constexpr int SORT_THREADS = 1024;
// LDS word of cell k: one pad word per 32 cells. The scan hands every lane a run of 32 consecutive cells; without the pad
// all 64 lanes of a wave walk the same bank (32-way conflicts on both scan loops).
__device__ __forceinline__ unsigned sort_cell(unsigned k) { return k + (k >> 5); }
constexpr size_t SORT_CELL_WORDS = MORTON_CELLS + MORTON_CELLS / 32;
__global__ __launch_bounds__(SORT_THREADS) void cull_sort_lds_kernel(
    const double *__restrict__ xyz, const int *__restrict__ type, long long type_fs,
    const double *__restrict__ box, long long n, unsigned short *__restrict__ keys, double *__restrict__ sxyz,
    int *__restrict__ stype, double4 *__restrict__ aos, long long n_pad, int n_ti, float near, int row_len, RowDisp disp)
{
    extern __shared__ unsigned s_cells[];  // [SORT_CELL_WORDS] + 3 x 16 doubles of scratch behind it
    double *s_red = reinterpret_cast<double *>(s_cells + SORT_CELL_WORDS);
    __shared__ unsigned s_part[SORT_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *x = xyz + (size_t)f * 3 * n;
    const double L[3] = {box[3 * f], box[3 * f + 1], box[3 * f + 2]};
    for (int k = tid; k < (int)SORT_CELL_WORDS; k += SORT_THREADS) s_cells[k] = 0u;
    // ---- origin = exact minimum of every axis ----
    double lo[3] = {1e300, 1e300, 1e300};
    for (long long i = tid; i < n; i += SORT_THREADS)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) lo[ax] = __builtin_fmin(lo[ax], x[ax * n + i]);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lo[ax] = __builtin_fmin(lo[ax], __shfl_down(lo[ax], off, 64));
        if ((tid & 63) == 0) s_red[ax * 16 + (tid >> 6)] = lo[ax];
    }
    __syncthreads();
    double org[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        double m = s_red[ax * 16];
        for (int w = 1; w < SORT_THREADS / 64; ++w) m = __builtin_fmin(m, s_red[ax * 16 + w]);
        org[ax] = m;
    }
    // ---- keys + cell populations ----
    const double G = (double)(1 << MORTON_BITS);
    unsigned short *kf = keys + (size_t)f * n;
    for (long long i = tid; i < n; i += SORT_THREADS) {
        unsigned c[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            int v = (int)(wrapped_frac(x[ax * n + i] - org[ax], L[ax]) * G);
            c[ax] = (unsigned)(v < 0 ? 0 : v > (1 << MORTON_BITS) - 1 ? (1 << MORTON_BITS) - 1 : v);
        }
        const unsigned key = hilbert3(c[0], c[1], c[2]);
        kf[i] = (unsigned short)key;
        atomicAdd(&s_cells[sort_cell(key)], 1u);
    }
    __syncthreads();
    // ---- exclusive scan of the populations: 32 consecutive cells per lane (stride 33 words between lanes), the lane
    // totals through wave shuffles, the 16 wave totals in order (round 3: two barriers instead of twenty-one) ----
    constexpr int PER = MORTON_CELLS / SORT_THREADS;
    static_assert(PER == 32, "one pad word per lane run");
    const int base = tid * (PER + 1);
    unsigned sum = 0;
    for (int k = 0; k < PER; ++k) sum += s_cells[base + k];
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int w = 0; w < wv; ++w) run += s_part[w];
    for (int k = 0; k < PER; ++k) {
        const unsigned v = s_cells[base + k];
        s_cells[base + k] = run;
        run += v;
    }
    __syncthreads();
    // ---- scatter ----
    for (long long i = tid; i < n; i += SORT_THREADS) {
        const unsigned pos = atomicAdd(&s_cells[sort_cell(kf[i])], 1u);
        const double px = x[i], py = x[n + i], pz = x[2 * n + i];
        const int t = type[(size_t)f * type_fs + i];
        if (sxyz) {
            double *o = sxyz + (size_t)f * 3 * n;
            o[pos] = px;
            o[n + pos] = py;
            o[2 * n + pos] = pz;
            stype[(size_t)f * n + pos] = t;
        }
        aos[(size_t)f * n_pad + pos] = make_double4(px, py, pz, pack_w(t, n_ti, near, row_len, disp));
    }
    for (long long i = n + tid; i < n_pad; i += SORT_THREADS)
        aos[(size_t)f * n_pad + i] = make_double4(PAD_J, PAD_J, PAD_J, __longlong_as_double(0LL));
}

// The same sort for frames of up to ITEMS x 1024 atoms, every atom read ONCE (round 5): the kernel above walks the frame
// three times (minimum, keys, scatter) in loops whose every trip waits for its own loads — one block per CU has nothing
// else to run meanwhile — and took 69 us for the 10k atoms of a C2 frame, a third of it arithmetic. Here a thread's
// ITEMS atoms (coordinates and type) are requested at the top, all at once, and stay in registers through the three
// phases; the keys never leave the registers either. 59 us (pre-pass of a C2 step 101 -> 92 us). What is left is not
// arithmetic (timing builds, -DSORT_EXP: a multiplication for the division of wrapped_frac 0 us, a plain key instead
// of the Hilbert arithmetic -8 us): the 200 blocks move 56 MB in and 64 MB out in lockstep — all read, all compute,
// all write — so the memory system idles through the middle of every block's life. (Non-temporal stores of the
// scattered records: 92 -> 204 us, they lose the L2's write combining; of cull_boxes_kernel's f32 records: the
// pre-pass the same, the sweep that reads them +17 us. Neither kept.)
template <int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void cull_sort_reg_kernel(
    const double *__restrict__ xyz, const int *__restrict__ type, long long type_fs, const double *__restrict__ box,
    long long n, double *__restrict__ sxyz, int *__restrict__ stype, double4 *__restrict__ aos, long long n_pad, int n_ti,
    float near, int row_len, RowDisp disp)
{
    extern __shared__ unsigned s_cells[];  // [SORT_CELL_WORDS] + 3 x 16 doubles of scratch behind it
    double *s_red = reinterpret_cast<double *>(s_cells + SORT_CELL_WORDS);
    __shared__ unsigned s_part[SORT_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *x = xyz + (size_t)f * 3 * n;
    const int *tf = type + (size_t)f * type_fs;
    double px[ITEMS], py[ITEMS], pz[ITEMS];
    int tp[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = tid + (long long)k * SORT_THREADS;
        px[k] = py[k] = pz[k] = 1e300;
        tp[k] = 0;
        if (i < n) {
            px[k] = x[i];
            py[k] = x[n + i];
            pz[k] = x[2 * n + i];
            tp[k] = tf[i];
        }
    }
    const double L[3] = {box[3 * f], box[3 * f + 1], box[3 * f + 2]};
    for (int k = tid; k < (int)SORT_CELL_WORDS; k += SORT_THREADS) s_cells[k] = 0u;
    // ---- origin = exact minimum of every axis ----
    double lo[3] = {1e300, 1e300, 1e300};
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        lo[0] = __builtin_fmin(lo[0], px[k]);
        lo[1] = __builtin_fmin(lo[1], py[k]);
        lo[2] = __builtin_fmin(lo[2], pz[k]);
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lo[ax] = __builtin_fmin(lo[ax], __shfl_down(lo[ax], off, 64));
        if (lane == 0) s_red[ax * 16 + wv] = lo[ax];
    }
    __syncthreads();
    double org[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        double m = s_red[ax * 16];
        for (int w = 1; w < SORT_THREADS / 64; ++w) m = __builtin_fmin(m, s_red[ax * 16 + w]);
        org[ax] = m;
    }
    // ---- keys + cell populations ----
    const double G = (double)(1 << MORTON_BITS);
    unsigned key[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = tid + (long long)k * SORT_THREADS;
        key[k] = 0u;
        if (i < n) {
            const double p[3] = {px[k], py[k], pz[k]};
            unsigned c[3];
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
#if SORT_EXP & 1
                const double sfr = (p[ax] - org[ax]) * (1.0 / L[ax]);
                double fr = sfr - __builtin_floor(sfr);
                fr = fr < 1.0 ? fr : 0.0;
                int v = (int)(fr * G);
#else
                int v = (int)(wrapped_frac(p[ax] - org[ax], L[ax]) * G);
#endif
                c[ax] = (unsigned)(v < 0 ? 0 : v > (1 << MORTON_BITS) - 1 ? (1 << MORTON_BITS) - 1 : v);
            }
#if SORT_EXP & 2
            key[k] = (c[0] | (c[1] << 5) | (c[2] << 10)) & 0xFFFFu;
#else
            key[k] = hilbert3(c[0], c[1], c[2]) & 0xFFFFu;  // (the other kernel keeps its keys in 16 bits)
#endif
            atomicAdd(&s_cells[sort_cell(key[k])], 1u);
        }
    }
    __syncthreads();
    // ---- exclusive scan of the populations (as above) ----
    constexpr int PER = MORTON_CELLS / SORT_THREADS;
    static_assert(PER == 32, "one pad word per lane run");
    const int base = tid * (PER + 1);
    unsigned sum = 0;
    for (int k = 0; k < PER; ++k) sum += s_cells[base + k];
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int w = 0; w < wv; ++w) run += s_part[w];
    for (int k = 0; k < PER; ++k) {
        const unsigned v = s_cells[base + k];
        s_cells[base + k] = run;
        run += v;
    }
    __syncthreads();
    // ---- scatter ----
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const long long i = tid + (long long)k * SORT_THREADS;
        if (i < n) {
            const unsigned pos = atomicAdd(&s_cells[sort_cell(key[k])], 1u);
            if (sxyz) {
                double *o = sxyz + (size_t)f * 3 * n;
                o[pos] = px[k];
                o[n + pos] = py[k];
                o[2 * n + pos] = pz[k];
                stype[(size_t)f * n + pos] = tp[k];
            }
            aos[(size_t)f * n_pad + pos] = make_double4(px[k], py[k], pz[k], pack_w(tp[k], n_ti, near, row_len, disp));
        }
    }
    for (long long i = n + tid; i < n_pad; i += SORT_THREADS)
        aos[(size_t)f * n_pad + i] = make_double4(PAD_J, PAD_J, PAD_J, __longlong_as_double(0LL));
}

// scatter atoms to their sorted position (cells[] holds running offsets)
__global__ void cull_scatter_kernel(const double *__restrict__ xyz, const int *__restrict__ type,
                                    long long type_fs, long long n, const unsigned short *__restrict__ keys,
                                    unsigned *__restrict__ cells, double *__restrict__ sxyz,
                                    int *__restrict__ stype, double4 *__restrict__ aos, long long n_pad,
                                    int n_ti, float near, int row_len, RowDisp disp)
{
    const int f = blockIdx.y;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned key = keys[(size_t)f * n + i];
    const unsigned pos = atomicAdd(&cells[(size_t)f * MORTON_CELLS + key], 1u);
    const double *x = xyz + (size_t)f * 3 * n;
    const double px = x[i], py = x[n + i], pz = x[2 * n + i];
    const int t = type[(size_t)f * type_fs + i];
    if (sxyz) {  // SoA copy: only the LDS-tile kernel reads it
        double *o = sxyz + (size_t)f * 3 * n;
        o[pos] = px;
        o[n + pos] = py;
        o[2 * n + pos] = pz;
        stype[(size_t)f * n + pos] = t;
    }
    aos[(size_t)f * n_pad + pos] = make_double4(px, py, pz, pack_w(t, n_ti, near, row_len, disp));
    // the pad records behind the last atom (never in cutoff: rsq overflows to +inf)
    if (i < n_pad - n) aos[(size_t)f * n_pad + n + i] = make_double4(PAD_J, PAD_J, PAD_J, __longlong_as_double(0LL));
}

// Bounding boxes of one tile of the sorted records, coordinates as given (not wrapped), all three levels
// in one pass: bbox[f][tile][6] (doubles, min xyz / max xyz, for the tile-pair lists), and in f32, widened
// so that rounding can only make them larger, the boxes of every 4 and every 8 consecutive atoms (culling groups
// of the scalar-j and of the LDS-tile kernel) and of every 64 (the i atoms of one wave): boxes[2*g] = (lo.xyz, 1), boxes[2*g+1] = (hi.xyz, 1);
// groups without atoms get w = 0 (never within reach).
__global__ __launch_bounds__(TILE) void cull_boxes_kernel(const double4 *__restrict__ aos,
                                                          const double *__restrict__ box, long long n, int nT,
                                                          double *__restrict__ bbox, float4 *__restrict__ gboxes,
                                                          float4 *__restrict__ wboxes, float4 *__restrict__ g4boxes,
                                                          double *__restrict__ cen, float *__restrict__ rel,
                                                          int rel_w_type, int cbox)
{
    __shared__ double red[6][TILE / 64];
    const int f = blockIdx.y, T = blockIdx.x, tid = threadIdx.x;
    const long long i = (long long)T * TILE + tid;
    const double4 me = aos[((size_t)f * nT + T) * TILE + tid];
    const bool real = i < n;
    double lo[3] = {real ? me.x : 1e300, real ? me.y : 1e300, real ? me.z : 1e300};
    double hi[3] = {real ? me.x : -1e300, real ? me.y : -1e300, real ? me.z : -1e300};
    // >> the f32 rounding (6e-8 relative) of a bound, whatever its magnitude
    const double pad0 = 1e-5 * (box[3 * f] + box[3 * f + 1] + box[3 * f + 2]) + 1e-6;
    auto widened = [&](float4 &lo4, float4 &hi4) {
        float l[3], h[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double pad = pad0 + 2.5e-7 * __builtin_fmax(__builtin_fabs(lo[ax]), __builtin_fabs(hi[ax]));
            l[ax] = (float)(lo[ax] - pad);
            h[ax] = (float)(hi[ax] + pad);
        }
        const float w = hi[0] >= lo[0] ? 1.f : 0.f;  // no atom: lo = 1e300 > hi
        lo4 = w > 0.f ? make_float4(l[0], l[1], l[2], 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        hi4 = w > 0.f ? make_float4(h[0], h[1], h[2], 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // cbox (packed-f32 sweep): the 4-atom and 64-atom boxes as (centre, half extents) instead — the sweep's group test
    // is then |nearest image of (c_wave - c_group)| - (h_wave + h_group) per axis, and its plain-difference test
    // |c_wave - n L - c_group| + (h_wave + h_group): a third of the instructions of the same tests on (lo, hi) pairs.
    // The centre is rounded to f32 first and the half extents taken around THAT value, widened as above (the padding
    // also covers the roundings of the sweep's own arithmetic: it grows with the coordinates like they do). A box
    // without atoms gets half extents of -1e18: its gap to anything is 1e18.
    auto centred = [&](float4 &c4, float4 &h4) {
        float cc[3], hh[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const double pad = pad0 + 2.5e-7 * __builtin_fmax(__builtin_fabs(lo[ax]), __builtin_fabs(hi[ax]));
            cc[ax] = (float)(0.5 * (lo[ax] + hi[ax]));
            hh[ax] = (float)(__builtin_fmax(hi[ax] - (double)cc[ax], (double)cc[ax] - lo[ax]) + pad);
        }
        const bool any = hi[0] >= lo[0];
        c4 = any ? make_float4(cc[0], cc[1], cc[2], 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        h4 = any ? make_float4(hh[0], hh[1], hh[2], 1.f) : make_float4(-1.0e18f, -1.0e18f, -1.0e18f, 0.f);
    };
    auto fold = [&](int m) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            lo[ax] = __builtin_fmin(lo[ax], __shfl_xor(lo[ax], m, 64));
            hi[ax] = __builtin_fmax(hi[ax], __shfl_xor(hi[ax], m, 64));
        }
    };
    fold(1);
    fold(2);
    if ((tid & 3) == 0) {  // every 4 atoms: the culling groups of the scalar-j kernel
        float4 l4, h4;
        if (cbox) centred(l4, h4);
        else widened(l4, h4);
        const size_t g = ((size_t)f * nT + T) * (TILE / 4) + (tid >> 2);
        g4boxes[2 * g] = l4;
        g4boxes[2 * g + 1] = h4;
    }
    fold(4);
    if ((tid & 7) == 0) {
        float4 l4, h4;
        widened(l4, h4);
        const size_t g = ((size_t)f * nT + T) * (TILE / 8) + (tid >> 3);
        gboxes[2 * g] = l4;
        gboxes[2 * g + 1] = h4;
    }
    fold(8);
    fold(16);
    fold(32);
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        float4 l4, h4;
        if (cbox) centred(l4, h4);
        else widened(l4, h4);
        const size_t w = ((size_t)f * nT + T) * (TILE / 64) + wave;
        wboxes[2 * w] = l4;
        wboxes[2 * w + 1] = h4;
        for (int ax = 0; ax < 3; ++ax) {
            red[ax][wave] = lo[ax];
            red[3 + ax][wave] = hi[ax];
        }
    }
    __syncthreads();
    if (tid < 6) {
        double v = red[tid][0];
        for (int w = 1; w < TILE / 64; ++w)
            v = tid < 3 ? __builtin_fmin(v, red[tid][w]) : __builtin_fmax(v, red[tid][w]);
        bbox[((size_t)f * nT + T) * 6 + tid] = v;
    }
    if (rel) {
        // packed-f32 sweep (pair_sj.hip MODE 3): the atoms relative to the centre c of their tile's box, rounded to f32
        // — |x - c| <= half extent, so the rounding error is 2^-24 of a few Angstrom instead of 2^-24 of the coordinate.
        // c and the half extents go to cen[f][T][8]; every thread of the block derives the same doubles.
        double c[3], hext[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            double l = red[ax][0], h = red[3 + ax][0];
            for (int w = 1; w < TILE / 64; ++w) {
                l = __builtin_fmin(l, red[ax][w]);
                h = __builtin_fmax(h, red[3 + ax][w]);
            }
            const bool any = h >= l;  // a tile of pad atoms only: lo = 1e300 > hi
            c[ax] = any ? 0.5 * (l + h) : 0.0;
            hext[ax] = any ? __builtin_fmax(h - c[ax], c[ax] - l) : 0.0;
        }
        if (tid < 3) {
            double *o = cen + ((size_t)f * nT + T) * 8;
            o[tid] = c[tid];
            o[3 + tid] = hext[tid];
        }
        // two atoms per 32-byte record: (x0, x1, y0, y1, z0, z1, w0, w1); pad atoms sit 1e18 away (rsq32 = 3e36:
        // finite, never in cutoff)
        float *o = rel + (((size_t)f * nT + T) * (TILE / 2) + (tid >> 1)) * 8 + (tid & 1);
        o[0] = real ? (float)(me.x - c[0]) : 1.0e18f;
        o[2] = real ? (float)(me.y - c[1]) : 1.0e18f;
        o[4] = real ? (float)(me.z - c[2]) : 1.0e18f;
        // w: the bin-guess addend near + type * row_len (ordered rows) or the row-table offset type * n_ti (class rows)
        o[6] = __int_as_float(rel_w_type ? __double2loint(me.w) : __double2hiint(me.w));
    }
}

// Lower bound of the reference's per-axis distance min(|d|, ||d| - L|) = dist(d, {0, +L, -L}) over all
// d = a - b with a in [a0,a1], b in [b0,b1]: the gap between the d interval and the nearest of those points.
__device__ __forceinline__ double interval_gap(double a0, double a1, double b0, double b1, double L)
{
    double g = __builtin_fmax(b0 - a1, a0 - b1);                   // to d = 0
    const double g1 = __builtin_fmax(b0 + L - a1, a0 - (b1 + L));  // to d = +L
    const double g2 = __builtin_fmax(b0 - L - a1, a0 - (b1 - L));  // to d = -L
    g = __builtin_fmin(g, __builtin_fmin(g1, g2));
    return g > 0.0 ? g : 0.0;
}

// list[f][I][*] = half-shell tiles J whose boxes come within the cutoff of tile I's box; cnt[f][I]
// TRI: atom-atom, half-shell candidates of the one tile set. !TRI: every tile of the j set (bbox_j, nTj tiles).
template <bool TRI>
__global__ __launch_bounds__(256) void cull_list_kernel(const double *__restrict__ bbox,
                                                        const double *__restrict__ bbox_j, int nTj,
                                                        const double *__restrict__ box, int nT, double rc2_test,
                                                        unsigned short *__restrict__ list, int *__restrict__ cnt)
{
    __shared__ int s_n;
    const int f = blockIdx.y, I = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const double *bi = bbox + ((size_t)f * nT + I) * 6;
    const double Lx = box[3 * f], Ly = box[3 * f + 1], Lz = box[3 * f + 2];
    // slack for the roundings of the box arithmetic (the pair kernel decides every listed pair exactly)
    const double sl = 1e-12 * (__builtin_fabs(bi[0]) + __builtin_fabs(bi[3]) + __builtin_fabs(bi[1]) +
                               __builtin_fabs(bi[4]) + __builtin_fabs(bi[2]) + __builtin_fabs(bi[5]) + Lx + Ly + Lz);
    unsigned short *row = list + ((size_t)f * nT + I) * (TRI ? nT : nTj);
    // TRI candidates = the half shell J = I, I+1, ..., I+S-1 (mod nT): every unordered tile pair belongs to
    // exactly one row and all rows have about the same length (a plain J >= I scan would be triangular)
    const int S = TRI ? tri_shifts(nT, I) : nTj;
    for (int sft = threadIdx.x; sft < S; sft += 256) {
        int J = TRI ? I + sft : sft;
        if (TRI) J = J >= nT ? J - nT : J;
        const double *bj = (TRI ? bbox : bbox_j) + ((size_t)f * (TRI ? nT : nTj) + J) * 6;
        double gx = interval_gap(bi[0], bi[3], bj[0], bj[3], Lx) - sl;
        double gy = interval_gap(bi[1], bi[4], bj[1], bj[4], Ly) - sl;
        double gz = interval_gap(bi[2], bi[5], bj[2], bj[5], Lz) - sl;
        gx = gx > 0.0 ? gx : 0.0;
        gy = gy > 0.0 ? gy : 0.0;
        gz = gz > 0.0 ? gz : 0.0;
        if (gx * gx + gy * gy + gz * gz <= rc2_test) row[atomicAdd(&s_n, 1)] = (unsigned short)J;
    }
    __syncthreads();
    if (threadIdx.x == 0) cnt[(size_t)f * nT + I] = s_n;
}

}  // namespace

int cull_prepare_set(mdhip_ctx *ctx, int64_t F, const double *d_x, const int *d_t, long long t_fs,
                     const double *d_box, long long N, int nT, int n_ti, float near, int row_len, RowDisp disp,
                     bool want_soa, int want_rel_i, int rel_w_type, int cbox, const int slot[5], SortedSet &out)
{
    const bool want_rel = want_rel_i != 0;
    MD_WS(d_rel, float, WS_REL, want_rel ? (size_t)F * nT * TILE * 16 : 64);
    MD_WS(d_cen, double, WS_CEN, want_rel ? (size_t)F * nT * 64 : 64);
    MD_WS(d_sx, double, WS_SORT_XYZ, want_soa ? (size_t)F * 3 * N * 8 : 64);
    MD_WS(d_st, int, WS_SORT_TYPE, want_soa ? (size_t)F * N * 4 : 64);
    MD_WS(d_keys, unsigned short, WS_KEYS, (size_t)F * N * 2);
    MD_WS(d_cells, unsigned, WS_CELLS, (size_t)F * MORTON_CELLS * 4);
    MD_WS(d_ao, double4, slot[0], (size_t)F * nT * TILE * sizeof(double4));
    MD_WS(d_bbox, double, slot[1], (size_t)F * nT * 6 * 8);
    MD_WS(d_gs, float4, slot[2], (size_t)F * nT * (TILE / 8) * 2 * sizeof(float4));
    MD_WS(d_ws, float4, slot[3], (size_t)F * nT * (TILE / 64) * 2 * sizeof(float4));
    MD_WS(d_g4, float4, slot[4], (size_t)F * nT * (TILE / 4) * 2 * sizeof(float4));
    const dim3 ga((unsigned)((N + 255) / 256), (unsigned)F);
    // one block per frame with the cell counters in LDS when there are frames enough to fill the chip (or the
    // frames are small); the multi-block path with global counters otherwise
    const size_t sort_lds = SORT_CELL_WORDS * 4 + 3 * 16 * 8;
    const bool lds_sort = ctx->opt_rdf_sort != 0 && sort_lds + 8192 <= ctx->lds_max && N <= 262144 &&
                          (ctx->opt_rdf_sort == 1 || F >= ctx->cu_count / 4 || N <= 16384);
    const int sort_items = (int)((N + SORT_THREADS - 1) / SORT_THREADS);
    if (lds_sort && sort_items <= 12 && ctx->opt_rdf_sort != 3) {
        // (frames of up to 12288 atoms: every atom read once, kept in registers — cull_sort_reg_kernel)
#define MD_SORT_REG(I)                                                                                              \
    {                                                                                                               \
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cull_sort_reg_kernel<I>),                         \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));                     \
        hipLaunchKernelGGL(cull_sort_reg_kernel<I>, dim3((unsigned)F), dim3(SORT_THREADS), sort_lds, ctx->stream,   \
                           d_x, d_t, t_fs, d_box, N, want_soa ? d_sx : (double *)nullptr, d_st, d_ao,               \
                           (long long)nT * TILE, n_ti, near, row_len, disp);                                        \
    }
        if (sort_items <= 2) MD_SORT_REG(2)
        else if (sort_items <= 4) MD_SORT_REG(4)
        else if (sort_items <= 6) MD_SORT_REG(6)
        else if (sort_items <= 8) MD_SORT_REG(8)
        else if (sort_items <= 10) MD_SORT_REG(10)
        else MD_SORT_REG(12)
#undef MD_SORT_REG
    } else if (lds_sort) {
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cull_sort_lds_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        hipLaunchKernelGGL(cull_sort_lds_kernel, dim3((unsigned)F), dim3(SORT_THREADS), sort_lds, ctx->stream, d_x,
                           d_t, t_fs, d_box, N, d_keys, want_soa ? d_sx : (double *)nullptr, d_st, d_ao,
                           (long long)nT * TILE, n_ti, near, row_len, disp);
    } else {
        MD_HIP(hipMemsetAsync(d_cells, 0, (size_t)F * MORTON_CELLS * 4, ctx->stream));
        MD_WS(d_org, double, WS_ORIGIN, (size_t)F * 3 * 8);
        hipLaunchKernelGGL(cull_origin_kernel, dim3((unsigned)F), dim3(256), 0, ctx->stream, d_x, N, d_org);
        hipLaunchKernelGGL(cull_keys_kernel, ga, dim3(256), 0, ctx->stream, d_x, d_box, N, d_org, d_keys, d_cells);
        hipLaunchKernelGGL(cull_scan_kernel, dim3((unsigned)F), dim3(256), 0, ctx->stream, d_cells);
        hipLaunchKernelGGL(cull_scatter_kernel, ga, dim3(256), 0, ctx->stream, d_x, d_t, t_fs, N, d_keys, d_cells,
                           want_soa ? d_sx : (double *)nullptr, d_st, d_ao, (long long)nT * TILE, n_ti, near,
                           row_len, disp);
    }
    hipLaunchKernelGGL(cull_boxes_kernel, dim3((unsigned)nT, (unsigned)F), dim3(TILE), 0, ctx->stream, d_ao, d_box, N,
                       nT, d_bbox, d_gs, d_ws, d_g4, want_rel ? d_cen : (double *)nullptr,
                       want_rel ? d_rel : (float *)nullptr, rel_w_type, cbox);
    MD_HIP(hipGetLastError());
    out.rel = want_rel ? d_rel : nullptr;
    out.cen = want_rel ? d_cen : nullptr;
    out.aos = d_ao;
    out.bbox = d_bbox;
    out.gs = d_gs;
    out.ws = d_ws;
    out.gs4 = d_g4;
    out.sx = d_sx;
    out.st = d_st;
    return MDHIP_OK;
}

void launch_cull_lists(hipStream_t stream, bool tri, int64_t F, const double *bbox_i, const double *bbox_j, int nTi,
                       int nTj, const double *d_box, double rc2_test, unsigned short *list, int *cnt)
{
    if (tri)
        hipLaunchKernelGGL(cull_list_kernel<true>, dim3((unsigned)nTi, (unsigned)F), dim3(256), 0, stream, bbox_i,
                           bbox_i, nTi, d_box, nTi, rc2_test, list, cnt);
    else
        hipLaunchKernelGGL(cull_list_kernel<false>, dim3((unsigned)nTi, (unsigned)F), dim3(256), 0, stream, bbox_i,
                           bbox_j, nTj, d_box, nTi, rc2_test, list, cnt);
}

}  // namespace mdpair
