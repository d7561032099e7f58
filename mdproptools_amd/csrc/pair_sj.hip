// pair_sj.hip — the scalar-j pair kernel (default of the culled sweep) and the merge of its per-block histograms.
// Formulation, exactness argument and binning: pair_hist.hip.
#include "pair_common.h"

#pragma clang fp contract(off)

namespace mdpair {
namespace {

// ------------------------------------------------------------------------------------------------
// Scalar-j kernel (culled path, rdf_variant = 1 default): the four waves of a block run independently.
// A wave keeps one i atom per lane; the j atoms of a group are the same for all lanes, so they are read
// with SCALAR loads (s_load_dwordx8 from the sorted record array, through the scalar cache) and used as
// scalar operands of the rsq chain — no LDS staging of tiles, no barrier per tile, so a wave that culls
// more groups than its neighbours never waits for them. LDS holds only the class histograms (shared by
// the block's waves), the row table and, for CN, the few edges. Binning and flush are the fast kernel's.
//
// Wrap decisions hoisted out of the pair loop. The wave knows the bounding box of its 64 i atoms and of
// every 8-atom j group (coordinates as given), hence the interval [dlo, dhi] that contains every
// d = xi - xj of the 512 pairs, per axis. The reference wraps d iff d > L/2 or d < -L/2, so
//   dlo >= -L/2 + m and dhi <= L/2 - m : no pair wraps            -> d' = d            (VAR 2: all three axes)
//   dlo >=  L/2 + m                    : every pair takes d - L   -> d' = d + s, s = -L (VAR 1: every axis is
//   dhi <= -L/2 - m                    : every pair takes d + L   -> d' = d + s, s = +L  one of the three)
//   otherwise                          : per-pair decision        -> min(|d|, ||d| - L|) (that axis only)
// d + (-L) is the reference's d - sign(d)*L operation and d + 0 is d, so the doubles entering rsq are the
// same in all three variants; m = 1e-4 * L/2 dwarfs the f32 rounding of the (outward widened) boxes.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

// Four consecutive 32-byte records (x, y, z, w) through the scalar cache into 4 x 8 SGPRs; `p` must be
// wave-uniform. The loads and the wait for them are ONE asm statement: hipcc knows nothing about the
// latency of an inline-asm load and would otherwise schedule uses of the outputs in front of a separate
// s_waitcnt (cdna_hip_programming.md §5.7). The record array is written by an earlier launch and only
// read here, so the scalar cache is coherent.
__device__ __forceinline__ void sload_records4(const double4 *p, u32x8 &r0, u32x8 &r1, u32x8 &r2, u32x8 &r3)
{
    asm volatile(
        "s_load_dwordx8 %0, %4, 0x0\n\t"
        "s_load_dwordx8 %1, %4, 0x20\n\t"
        "s_load_dwordx8 %2, %4, 0x40\n\t"
        "s_load_dwordx8 %3, %4, 0x60\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3)
        : "s"(p)
        : "memory");
}

struct AxisL {
    double Lx, Ly, Lz;  // box lengths (VAR 0)
    double sx, sy, sz;  // wave-uniform shifts in {-L, 0, +L} (VAR 1)
};

// VAR of sweep_group_sj: bits 0..2 = axes (x, y, z) that need the per-pair wrap decision, the other axes add
// their wave-uniform shift; VAR = 8: no axis wraps at all.
template <int VAR, int AXIS>
__device__ __forceinline__ double axis_abs(double d, double L, double sft)
{
    if (VAR == 8) return d;
    if (VAR & (1 << AXIS)) return wrap_abs(d, L);
    return d + sft;
}

template <bool DIAG, int MODE, int VAR>
__device__ __forceinline__ void sweep_group_sj(const double4 *__restrict__ grp, int local0, double xi, double yi,
                                               double zi, const AxisL &L, double rc2, const FastCtx &c,
                                               int lane_in_tile)
{
    constexpr int U = 4;  // records per batch of scalar loads (4 x 8 SGPRs)
#pragma unroll
    for (int h = 0; h < SJ_GROUP / U; ++h) {
        u32x8 rec[U];
        sload_records4(grp + h * U, rec[0], rec[1], rec[2], rec[3]);
        double rsq[U];
        unsigned row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double xj = __hiloint2double((int)rec[u][1], (int)rec[u][0]);
            const double yj = __hiloint2double((int)rec[u][3], (int)rec[u][2]);
            const double zj = __hiloint2double((int)rec[u][5], (int)rec[u][4]);
            const double ax = axis_abs<VAR, 0>(xi - xj, L.Lx, L.sx);
            const double ay = axis_abs<VAR, 1>(yi - yj, L.Ly, L.sy);
            const double az = axis_abs<VAR, 2>(zi - zj, L.Lz, L.sz);
            rsq[u] = (ax * ax + ay * ay) + az * az;
            if (MODE != 2) row[u] = c.rowtab_me[(int)rec[u][6]];  // low word of w = type * n_ti
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool in = rsq[u] < rc2;
            if (DIAG) in = in && (local0 + h * U + u > lane_in_tile);
            if (in) {
                int k;
                if (MODE == 2) {
                    // ordered-pair rows: the word offset of row (., tj) rides in the addend of the bin guess
                    // (high word of w = float(near + tj * row_len)), so no row lookup at all:
                    // trunc(g1) = tj * row_len + bin, and fract(g1) is the same guard-band test as in MODE 0
                    const float nearoff = __uint_as_float(rec[u][7]);
                    const float g1 = __builtin_fmaf(__builtin_amdgcn_sqrtf((float)rsq[u]), c.gscale, nearoff);
                    k = (int)g1;
                    if (__builtin_amdgcn_fractf(g1) < c.near2) {
                        // g1 is within 2*near above an integer: the true bin is that integer or the one below
                        // (|error| < near), and the exact edge of that integer decides
                        const int koff = (int)nearoff;  // near < 1: truncation gives tj * row_len back
                        int kk = k - koff;
                        kk = kk > c.nbins ? c.nbins : (kk < 0 ? 0 : kk);
                        k = koff + (rsq[u] < c.edges[kk] ? kk - 1 : kk);
                    }
                    const unsigned addr2 = ((unsigned)k << 2) + c.rowbase_me;
                    asm volatile("ds_add_u32 %0, %1" ::"v"(addr2), "v"(1u) : "memory");
                    continue;
                }
                if (MODE == 0) {
                    const float g1 = __builtin_fmaf(__builtin_amdgcn_sqrtf((float)rsq[u]), c.gscale, c.near);
                    k = (int)g1;
                    if (__builtin_amdgcn_fractf(g1) < c.near2) {
                        // g1 is within 2*near above the integer k: the true bin is k or k - 1 (|error| < near),
                        // and the exact edge of k decides
                        k = k > c.nbins ? c.nbins : k;
                        k = rsq[u] < c.edges[k] ? k - 1 : k;
                    }
                } else {
                    k = 0;
                    for (int e = 1; e <= c.nbins; ++e) k += rsq[u] >= c.edges[e] ? 1 : 0;
                }
                const unsigned addr = ((unsigned)k << 2) + row[u];
                asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
            }
        }
    }
}

// per-axis wrap class of a (wave box, group box) pair: bit 0 = every pair takes d - L, bit 1 = every pair
// takes d + L, bit 2 = undecided (per-pair decision needed); 0 = no pair wraps
__device__ __forceinline__ unsigned wrap_class(float wlo, float whi, float glo, float ghi, float L)
{
    const float dlo = wlo - ghi, dhi = whi - glo;
    const float h = 0.5f * L, m = 1.0e-4f * h;
    if (dlo >= -(h - m) && dhi <= h - m) return 0u;
    if (dlo >= h + m) return 1u;
    if (dhi <= -(h + m)) return 2u;
    return 4u;
}

// One work item of the scalar-j sweep: the 64 i atoms of wave `wq` of tile I of frame f against slice
// `split` of the tile's neighbour list.
template <int MODE>
__device__ __forceinline__ void sj_item(const PairArgs &a, FastCtx &c, const unsigned *s_row, int f, int I, int wq,
                                        int split, int lane)
{
    const long long n_pad = (long long)a.nTi * TILE, n_pad_j = (long long)a.nTj * TILE;
    const long long rowid = (long long)f * a.nTi + I;
    const int cnt = a.list_cnt[rowid];
    const unsigned short *row_list = a.list + rowid * a.nTj;  // (nTj == nTi for atom-atom)
    const int t_begin = (int)((long long)split * cnt / a.jsplit);
    const int t_end = (int)((long long)(split + 1) * cnt / a.jsplit);
    if (t_begin >= t_end) return;
    AxisL L;
    L.Lx = a.box[3 * f];
    L.Ly = a.box[3 * f + 1];
    L.Lz = a.box[3 * f + 2];
    L.sx = L.sy = L.sz = 0.0;
    const double4 *ats = a.aos + (long long)f * n_pad;
    const int lane_in_tile = wq * 64 + lane;
    const long long ig = (long long)I * TILE + lane_in_tile;
    double4 me = ats[ig];
    if (ig >= a.ni) me = make_double4(PAD_I, PAD_I, PAD_I, __longlong_as_double(0LL));
    {
        const int ti_me = (int)((unsigned)__double_as_longlong(me.w)) / a.n_ti;  // low word of w = type * n_ti
        c.rowtab_me = s_row + ti_me;
        c.rowbase_me = c.lds_base + (unsigned)ti_me * (unsigned)a.n_tj * (unsigned)(a.nbins + 1) * 4u;
    }
    const long long w = ((long long)f * a.nTi + I) * (TILE / 64) + wq;
    const float4 wlo = a.wsph[2 * w], whi = a.wsph[2 * w + 1];
    const float4 *gb_f = a.gsph4 + (long long)f * a.nTj * (TILE / SJ_GROUP) * 2;  // group boxes of the j set
    const double4 *ats_j = a.aos_j + (long long)f * n_pad_j;
    const float fLx = (float)L.Lx, fLy = (float)L.Ly, fLz = (float)L.Lz;
    for (int t = t_begin; t < t_end; ++t) {
        const int J = __builtin_amdgcn_readfirstlane((int)row_list[t]);
        // every lane tests one 4-atom group box of the tile against this wave's box
        const float4 glo = gb_f[((long long)J * (TILE / SJ_GROUP) + lane) * 2];
        const float4 ghi = gb_f[((long long)J * (TILE / SJ_GROUP) + lane) * 2 + 1];
        const float gx = gapf(wlo.x, whi.x, glo.x, ghi.x, fLx);
        const float gy = gapf(wlo.y, whi.y, glo.y, ghi.y, fLy);
        const float gz = gapf(wlo.z, whi.z, glo.z, ghi.z, fLz);
        const bool keep = wlo.w > 0.f && glo.w > 0.f && gx * gx + gy * gy + gz * gz < a.reach * a.reach;
        const double4 *tile = ats_j + (long long)J * TILE;
        if (a.tri && J == I) {
            unsigned long long mask = __builtin_amdgcn_ballot_w64(keep);
            while (mask) {
                const int g = __builtin_ctzll(mask);
                mask &= mask - 1;
                sweep_group_sj<true, MODE, 7>(tile + g * SJ_GROUP, g * SJ_GROUP, me.x, me.y, me.z, L, a.rc2, c, lane_in_tile);
            }
            continue;
        }
        const unsigned cx = wrap_class(wlo.x, whi.x, glo.x, ghi.x, fLx);
        const unsigned cy = wrap_class(wlo.y, whi.y, glo.y, ghi.y, fLy);
        const unsigned cz = wrap_class(wlo.z, whi.z, glo.z, ghi.z, fLz);
        // groups by the set of axes that still need the per-pair decision (bit k = axis k); the decided axes add
        // their wave-uniform shift; groups where nothing wraps at all take the shortest chain
        const unsigned amb = (cx >> 2) | ((cy >> 2) << 1) | ((cz >> 2) << 2);
        const bool none = !(cx | cy | cz);
        unsigned long long m8 = __builtin_amdgcn_ballot_w64(keep && none);
        while (m8) {
            const int g = __builtin_ctzll(m8);
            m8 &= m8 - 1;
            sweep_group_sj<false, MODE, 8>(tile + g * SJ_GROUP, g * SJ_GROUP, me.x, me.y, me.z, L, a.rc2, c, lane_in_tile);
        }
        if (!__builtin_amdgcn_ballot_w64(keep && !none)) continue;
        const unsigned long long xm = __builtin_amdgcn_ballot_w64(cx == 1u), xp = __builtin_amdgcn_ballot_w64(cx == 2u);
        const unsigned long long ym = __builtin_amdgcn_ballot_w64(cy == 1u), yp = __builtin_amdgcn_ballot_w64(cy == 2u);
        const unsigned long long zm = __builtin_amdgcn_ballot_w64(cz == 1u), zp = __builtin_amdgcn_ballot_w64(cz == 2u);
#define SJ_VARIANT(A)                                                                                        \
    {                                                                                                        \
        unsigned long long mk = __builtin_amdgcn_ballot_w64(keep && !none && amb == (A));                    \
        while (mk) {                                                                                         \
            const int g = __builtin_ctzll(mk);                                                               \
            mk &= mk - 1;                                                                                    \
            AxisL S = L;                                                                                     \
            S.sx = ((xm >> g) & 1ull) ? -L.Lx : ((xp >> g) & 1ull) ? L.Lx : 0.0;                                 \
            S.sy = ((ym >> g) & 1ull) ? -L.Ly : ((yp >> g) & 1ull) ? L.Ly : 0.0;                                 \
            S.sz = ((zm >> g) & 1ull) ? -L.Lz : ((zp >> g) & 1ull) ? L.Lz : 0.0;                                 \
            sweep_group_sj<false, MODE, (A)>(tile + g * SJ_GROUP, g * SJ_GROUP, me.x, me.y, me.z, S, a.rc2, c, lane_in_tile); \
        }                                                                                                    \
    }
        SJ_VARIANT(0)
        SJ_VARIANT(1)
        SJ_VARIANT(2)
        SJ_VARIANT(3)
        SJ_VARIANT(4)
        SJ_VARIANT(5)
        SJ_VARIANT(6)
        SJ_VARIANT(7)
#undef SJ_VARIANT
    }
}

// PERSIST = true (frame-summed output): the grid is one resident set of blocks; every WAVE draws items
// (frame, tile, wave, list slice) from its XCD's counter — frames stay dealt to XCDs (f % 8) so a frame's
// records live in one L2 — and the block flushes its LDS histograms once, when its four waves have run
// dry. Every wave leaves the loop as soon as the counter passes the item count.
// PERSIST = false (per-frame output): block = (frame, tile, list slice), one flush per block.
template <int MODE, bool PERSIST>
__global__ __launch_bounds__(TILE) void pair_hist_sj_kernel(const PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const long long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);

    // ---- LDS: hist | (CN edges) | row table ----
    // MODE 2: one row per ORDERED type pair (ti, tj), addressed without a table (see sweep_group_sj)
    const int row_len = a.nbins + 1;
    const int hist_words = (MODE == 2 ? a.n_ti * a.n_tj : a.n_cls + 1) * row_len;
    unsigned *s_hist = reinterpret_cast<unsigned *>(smem);
    size_t off = ((size_t)hist_words * 4 + 15) & ~size_t(15);
    double *s_edges = reinterpret_cast<double *>(smem + off);
    off += MODE == 1 ? (((size_t)(a.nbins + 2) * 8 + 15) & ~size_t(15)) : 0;
    unsigned *s_row = reinterpret_cast<unsigned *>(smem + off);
    const unsigned lds_base =
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)smem;
    for (int k = tid; k < hist_words; k += TILE) s_hist[k] = 0u;
    if (MODE != 2)
        for (int k = tid; k < a.n_ti * a.n_tj; k += TILE) {
            const int ti = k % a.n_ti, tj = k / a.n_ti;
            const unsigned cl = a.cls[ti * a.n_tj + tj];
            s_row[k] = lds_base + (cl == 0xFFu ? (unsigned)a.n_cls : cl) * (unsigned)row_len * 4u;
        }
    FastCtx c;
    c.hist = s_hist;
    c.edges = a.edges;
    if (MODE == 1) {
        for (int k = tid; k <= a.nbins + 1; k += TILE) s_edges[k] = a.edges[k];
        c.edges = s_edges;
    }
    c.gscale = a.gscale;
    if (MODE == 2) {
        // the bin guess is fma(sqrt, gscale, addend) with the addend in an SGPR (it belongs to the j atom): a VOP3
        // may read one SGPR, so gscale has to live in a VGPR or every guess pays a v_mov
        float gs;
        asm volatile("v_mov_b32 %0, %1" : "=v"(gs) : "s"(a.gscale));
        c.gscale = gs;
    }
    c.near = MODE == 2 ? a.near : (float)a.nbins * 1.0e-6f + 1.0e-5f;
    c.near2 = 2.0f * c.near;
    c.nbins = a.nbins;
    c.lds_base = lds_base;
    c.rowbase_me = lds_base;
    __syncthreads();  // tables ready; from here on the waves do not synchronise until the flush

    const int lane = tid & 63;
    int f_out = 0;
    if (PERSIST) {
        const int nfx = a.n_frames > xcd ? (a.n_frames - xcd + 7) / 8 : 0;  // frames of this XCD
        const int ipf = a.nTi * (TILE / 64) * a.jsplit;                       // items per frame
        const long long n_items = (long long)nfx * ipf;
        for (;;) {
            unsigned it = 0;
            if (lane == 0) it = atomicAdd(&a.work[xcd], 1u);
            it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);
            if ((long long)it >= n_items) break;
            const int fx = (int)(it / (unsigned)ipf), r = (int)(it % (unsigned)ipf);
            const int split = r % a.jsplit, wI = r / a.jsplit;
            sj_item<MODE>(a, c, s_row, fx * 8 + xcd, wI >> 2, wI & 3, split, lane);
        }
    } else {
        // a.blocks_per_frame blocks share one frame and flush once each into the frame's row
        // (frames stay dealt to XCDs, f % 8 = XCD, so that a frame's records are fetched into one L2)
        const int f = (int)((bid >> 3) / a.blocks_per_frame) * 8 + xcd;
        f_out = f < a.n_frames ? f : 0;
        const unsigned ipf = f < a.n_frames ? (unsigned)(a.nTi * (TILE / 64) * a.jsplit) : 0u;
        for (;;) {  // the frame's blocks draw its wave items from the frame's counter (integer sums: any order)
            unsigned it = 0;
            if (ipf == 0u) break;
            if (lane == 0) it = atomicAdd(&a.work[f], 1u);
            it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);
            if (it >= ipf) break;
            const int split = (int)(it % (unsigned)a.jsplit), wI = (int)(it / (unsigned)a.jsplit);
            sj_item<MODE>(a, c, s_row, f, wI >> 2, wI & 3, split, lane);
        }
    }

    // ---- flush: the block's LDS histogram goes to its own slice with plain coalesced stores (device-scope
    // atomics on rows spread over HBM cost ~40 ps each: 10^7 of them per launch were 6 % of the kernel);
    // merge_slices_kernel adds the slices up afterwards ----
    (void)f_out;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned *slice = a.slices + (size_t)bid * (size_t)hist_words;
    for (int w = tid; w < hist_words; w += TILE) slice[w] = s_hist[w];
}

// rows[o][w] = sum of slice word w over the blocks of output o: per-frame output o = frame f, whose blocks are
// ((f / 8) * bpf + sub) * 8 + f % 8, sub < bpf; frame-summed output: all blocks, split over gridDim.y chunks
// whose partial sums are added with (few) 64-bit atomics into the zeroed row buffer.
__global__ void merge_slices_kernel(const unsigned *__restrict__ slices, int hist_words, long long n_blocks,
                                    int per_frame, int bpf, unsigned long long *__restrict__ rows)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= hist_words) return;
    unsigned long long sum = 0;
    if (per_frame) {
        const long long f = blockIdx.y;
        const long long b0 = ((f >> 3) * bpf) * 8 + (f & 7);
        for (int sub = 0; sub < bpf; ++sub) sum += slices[(size_t)(b0 + 8LL * sub) * hist_words + w];
        rows[(size_t)f * hist_words + w] = sum;
    } else {
        const long long per = (n_blocks + gridDim.y - 1) / gridDim.y;
        const long long b1 = std::min<long long>(n_blocks, (blockIdx.y + 1) * per);
        for (long long b = blockIdx.y * per; b < b1; ++b) sum += slices[(size_t)b * hist_words + w];
        if (sum) atomicAdd(&rows[w], sum);
    }
}

}  // namespace

size_t lds_bytes_sj_ordered(int nbins, int n_ti, int n_tj)
{
    return (((size_t)n_ti * n_tj * (nbins + 1) * 4 + 15) & ~size_t(15)) + 16;
}

size_t lds_bytes_sj(int nbins, int n_cls, int n_ti, int n_tj, bool mode_cn)
{
    size_t off = ((size_t)(n_cls + 1) * (nbins + 1) * 4 + 15) & ~size_t(15);
    off += mode_cn ? (((size_t)(nbins + 2) * 8 + 15) & ~size_t(15)) : 0;
    off += (size_t)n_ti * n_tj * 4;
    return (off + 15) & ~size_t(15);
}

PairKernel sj_kernel(int mode, bool persist, const char **name)
{
#define MD_PICK(...) (*name = #__VA_ARGS__, __VA_ARGS__)
    if (mode == 2) return persist ? MD_PICK(pair_hist_sj_kernel<2, true>) : MD_PICK(pair_hist_sj_kernel<2, false>);
    if (mode == 1) return persist ? MD_PICK(pair_hist_sj_kernel<1, true>) : MD_PICK(pair_hist_sj_kernel<1, false>);
    return persist ? MD_PICK(pair_hist_sj_kernel<0, true>) : MD_PICK(pair_hist_sj_kernel<0, false>);
#undef MD_PICK
}

void launch_merge_slices(hipStream_t stream, const unsigned *slices, int hist_words, long long n_blocks, int per_frame,
                         int bpf, unsigned grid_y, unsigned long long *rows)
{
    hipLaunchKernelGGL(merge_slices_kernel, dim3((unsigned)((hist_words + 255) / 256), grid_y), dim3(256), 0, stream,
                       slices, hist_words, n_blocks, per_frame, bpf, rows);
}

}  // namespace mdpair
