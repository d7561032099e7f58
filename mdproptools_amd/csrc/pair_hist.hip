// pair_hist.hip — all-pairs minimum-image distance histogramming for gfx950 (R1-R5).
//
// Replaces structural/rdf_cn.py:35-162 of the reference (_calc_rsq, _remove_outliers, _rdf_loop,
// _cn_loop, _rdf_mol_loop, _cn_mol_loop). One kernel family serves all four loops:
//
//   H[class(type_i, type_j)][bin(rsq)] += 1      for every in-cutoff pair, exactly once
//
// where `bin` is a comparison of rsq against a host-made table of exact edges and `class` is a small
// table lookup. The reference's outputs are integer-linear in H (see derive_* below), so the kernel
// never sees the relation list. RDF uses the edges of trunc(sqrt(rsq)/ddr); CN uses the sorted
// distinct cutoffs^2 as edges.
//
// Bit-exactness: rsq is built from exactly-rounded IEEE double ops in the reference's order,
// with contraction off for this whole translation unit:
//   d = head - other; if |d| > L/2: d -= sign(d)*L; rsq = (dx*dx + dy*dy) + dz*dz
// The wrap is evaluated as  a = |d|;  a' = min(a, |a - L|)  which selects the same double:
//   |d - sign(d)L| == | |d| - L |  (rounding is sign-symmetric), and for a > L/2 the real value
//   |a - L| < a while for a <= L/2 it is >= a; rounding to nearest is monotone and `a` is itself a
//   double, so the comparison of the rounded value against `a` falls the same way (ties give equal
//   values). Only a^2 enters rsq, so the lost sign is irrelevant.
//
// Work decomposition: a block owns one tile of 256 "i" atoms of one frame (one atom per lane, in
// registers) and sweeps a list of 256-atom "j" tiles staged through LDS (double-buffered, one
// barrier per tile). For the triangular (atom-atom) case the j list is the half shell
// J = I, I+1, ..., I+nT/2 (mod nT), which covers every unordered tile pair once with equal work per
// block; only the J == I tile needs the i<j mask. Class histograms are LDS-private per block
// (ds_add_u32) and flushed once with 64-bit global atomics into one of `slots` replicas.
#include <algorithm>
#include <cmath>
#include <limits>

#include "ctx.h"

#pragma clang fp contract(off)

namespace {

constexpr int TILE = 256;
constexpr double PAD_I = -1.0e300;  // padding atoms: rsq overflows to +inf, never in cutoff, never NaN
constexpr double PAD_J = 1.0e300;

struct __attribute__((aligned(16))) JAtom {
    double x, y, z;
    int t;
    int pad;
};

struct PairArgs {
    const double *xi;  // [F][3][ni]
    const double *xj;  // [F][3][nj]
    const int *ti;     // compact type index of i atoms
    const int *tj;
    const double *box;           // [F][3]
    const unsigned char *cls;    // [n_ti][n_tj] -> class, 0xFF = not counted in this pass
    const double *edges;         // [nbins+1]
    unsigned long long *hist;    // [slots | F][n_cls][nbins]
    unsigned long long *overflow;
    long long ni, nj, ti_fs, tj_fs;
    double rc2;
    float gscale;  // 1/bin_size as float for the sqrt guess; 0 -> scan up from bin 0 (CN edges)
    int n_ti, n_tj, n_cls, nbins;
    int n_frames, nTi, nTj, jsplit, blocks_per_frame;
    int per_frame, slots;
    int fpb;  // frames swept per block before the flush (fast kernel, frame-summed output only)
    // culled path: per (frame, i-tile) the j-tiles (>= I) whose bounding boxes come within the cutoff
    const unsigned short *list;  // [F][nTi][nTi]
    const int *list_cnt;         // [F][nTi]
    const float4 *gsph;          // [F][nTi*32][2] bounding box (lo, hi; w = 1 if non-empty) of every 8 sorted atoms
    const float4 *wsph;          // [F][nTi*4][2]  bounding box of every 64 sorted atoms (one wave's i atoms)
    float reach;                 // r_cut rounded up, plus slack for the f32 box test
    const double4 *aos;          // [F][nTi*256] sorted atoms (x, y, z, bits = type * n_ti), padded with +1e300
    unsigned *work;              // work counters of the scalar-j kernel, zeroed per launch: [8] per XCD
                                 // (frame-summed output) or [F] per frame (per-frame output)
    float near;                  // MODE 2: guard band half-width (also folded into the records' row offsets)
    unsigned *slices;            // scalar-j kernels: [blocks][LDS histogram words], every block stores its own copy
    const double4 *aos_j;        // sorted records of the j set (== aos for atom-atom), [F][nTj*256]
    int tri;                     // 1: atom-atom (i < j inside the diagonal tile), 0: atoms x sites
};

__device__ __forceinline__ double wrap_abs(double d, double L)
{
    double a = __builtin_fabs(d);
    double w = __builtin_fabs(a - L);
    return __builtin_fmin(a, w);
}

// periodic gap of two intervals inside [0,L) (f32, for the in-kernel group test)
__device__ __forceinline__ float gapf(float alo, float ahi, float blo, float bhi, float L)
{
    float g = __builtin_fmaxf(blo - ahi, alo - bhi);
    const float g1 = __builtin_fmaxf(blo + L - ahi, alo - (bhi + L));
    const float g2 = __builtin_fmaxf(blo - L - ahi, alo - (bhi - L));
    g = __builtin_fminf(g, __builtin_fminf(g1, g2));
    return g > 0.f ? g : 0.f;
}

struct BinCtx {
    const double *edges;   // LDS, nbins+2 entries, last = +inf
    unsigned *hist;        // LDS
    unsigned *ovf;         // LDS
    const unsigned char *cls_row;  // LDS row of this lane's i type
    float gscale;
    int nbins;
};

__device__ __forceinline__ void count_pair(const BinCtx &b, double rsq, int tj)
{
    int k = 0;
    if (b.gscale > 0.f) {
        k = (int)(__builtin_amdgcn_sqrtf((float)rsq) * b.gscale);
        k = k > b.nbins ? b.nbins : k;
        while (rsq < b.edges[k]) --k;  // edges[0] == 0 stops it
    }
    while (rsq >= b.edges[k + 1]) ++k;  // edges[nbins+1] == +inf stops it
    if (k < b.nbins) {
        unsigned c = b.cls_row[tj];
        if (c != 0xFFu) atomicAdd(&b.hist[c * b.nbins + k], 1u);
    } else {
        atomicAdd(b.ovf, 1u);
    }
}

template <bool DIAG>
__device__ __forceinline__ void sweep_tile(const JAtom *__restrict__ tile, double xi, double yi,
                                           double zi, double Lx, double Ly, double Lz, double rc2,
                                           const BinCtx &b, int lane_id)
{
#pragma unroll 4
    for (int jj = 0; jj < TILE; ++jj) {
        const JAtom pj = tile[jj];
        const double ax = wrap_abs(xi - pj.x, Lx);
        const double ay = wrap_abs(yi - pj.y, Ly);
        const double az = wrap_abs(zi - pj.z, Lz);
        const double rsq = (ax * ax + ay * ay) + az * az;
        bool in = rsq < rc2;
        if (DIAG) in = in && (jj > lane_id);
        if (in) count_pair(b, rsq, pj.t);
    }
}

__device__ __forceinline__ JAtom load_atom(const double *__restrict__ xyz, const int *__restrict__ t,
                                           long long n, long long g, double pad)
{
    JAtom a;
    if (g < n) {
        a.x = xyz[g];
        a.y = xyz[n + g];
        a.z = xyz[2 * n + g];
        a.t = t[g];
    } else {
        a.x = a.y = a.z = pad;
        a.t = 0;
    }
    a.pad = 0;
    return a;
}

// number of half-shell shifts owned by i-tile I when there are nT tiles
__device__ __host__ __forceinline__ int tri_shifts(int nT, int I)
{
    return (nT & 1) ? (nT + 1) / 2 : nT / 2 + (I < nT / 2 ? 1 : 0);
}

template <bool TRI>
__global__ __launch_bounds__(TILE) void pair_hist_kernel(const PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;

    // ---- which frame / i-tile / slice of the j list (XCD-aware: frames are dealt to XCDs) ----
    const long long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long long q = bid >> 3;
    const int f = (int)(q / a.blocks_per_frame) * 8 + xcd;
    if (f >= a.n_frames) return;
    const int within = (int)(q % a.blocks_per_frame);
    const int I = within % a.nTi;
    const int split = within / a.nTi;

    int t_begin, t_end;  // range in the block's j list
    if (TRI) {
        const int S = tri_shifts(a.nTi, I);
        t_begin = (int)((long long)split * S / a.jsplit);
        t_end = (int)((long long)(split + 1) * S / a.jsplit);
    } else {
        t_begin = (int)((long long)split * a.nTj / a.jsplit);
        t_end = (int)((long long)(split + 1) * a.nTj / a.jsplit);
    }
    if (t_begin >= t_end) return;

    // ---- LDS carve-up ----
    double *s_edges = reinterpret_cast<double *>(smem);
    size_t off = (((size_t)(a.nbins + 2) * 8) + 15) & ~size_t(15);
    JAtom *s_tile = reinterpret_cast<JAtom *>(smem + off);
    off += sizeof(JAtom) * 2 * TILE;
    unsigned *s_hist = reinterpret_cast<unsigned *>(smem + off);
    const int hist_words = a.n_cls * a.nbins;
    off += (size_t)hist_words * 4;
    unsigned *s_ovf = reinterpret_cast<unsigned *>(smem + off);
    off += 16;
    unsigned char *s_cls = smem + off;

    for (int k = tid; k <= a.nbins; k += TILE) s_edges[k] = a.edges[k];
    if (tid == 0) {
        s_edges[a.nbins + 1] = __builtin_inf();
        *s_ovf = 0u;
    }
    for (int k = tid; k < hist_words; k += TILE) s_hist[k] = 0u;
    for (int k = tid; k < a.n_ti * a.n_tj; k += TILE) s_cls[k] = a.cls[k];

    // ---- this lane's i atom ----
    const double *xi_f = a.xi + (long long)f * 3 * a.ni;
    const double *xj_f = a.xj + (long long)f * 3 * a.nj;
    const int *ti_f = a.ti + (long long)f * a.ti_fs;
    const int *tj_f = a.tj + (long long)f * a.tj_fs;
    const double Lx = a.box[3 * f], Ly = a.box[3 * f + 1], Lz = a.box[3 * f + 2];
    const JAtom me = load_atom(xi_f, ti_f, a.ni, (long long)I * TILE + tid, PAD_I);

    BinCtx b;
    b.edges = s_edges;
    b.hist = s_hist;
    b.ovf = s_ovf;
    b.cls_row = s_cls + me.t * a.n_tj;
    b.gscale = a.gscale;
    b.nbins = a.nbins;
    auto tile_of = [&](int t) -> int {
        if (TRI) {
            int J = I + t;
            return J >= a.nTi ? J - a.nTi : J;
        }
        return t;
    };

    // ---- sweep the j list, staging tiles through two LDS buffers ----
    JAtom nxt = load_atom(xj_f, tj_f, a.nj, (long long)tile_of(t_begin) * TILE + tid, PAD_J);
    s_tile[tid] = nxt;
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end)
            nxt = load_atom(xj_f, tj_f, a.nj, (long long)tile_of(t + 1) * TILE + tid, PAD_J);
        if (TRI && t == 0)
            sweep_tile<true>(s_tile + buf * TILE, me.x, me.y, me.z, Lx, Ly, Lz, a.rc2, b, tid);
        else
            sweep_tile<false>(s_tile + buf * TILE, me.x, me.y, me.z, Lx, Ly, Lz, a.rc2, b, tid);
        if (t + 1 < t_end) s_tile[(buf ^ 1) * TILE + tid] = nxt;
        __syncthreads();
    }

    // ---- flush: one 64-bit global atomic per non-empty LDS word ----
    unsigned long long *g =
        a.hist + (size_t)(a.per_frame ? f : (int)(bid % a.slots)) * (size_t)hist_words;
    for (int k = tid; k < hist_words; k += TILE) {
        const unsigned v = s_hist[k];
        if (v) atomicAdd(&g[k], (unsigned long long)v);
    }
    if (tid == 0 && *s_ovf) atomicAdd(a.overflow, (unsigned long long)*s_ovf);
}

// ------------------------------------------------------------------------------------------------
// Fast variant (rdf_variant = 1, RDF edge tables only): same arithmetic for rsq, cheaper bookkeeping.
//  * LDS: the class histograms sit at offset 0 as (n_cls+1) rows of (nbins+1) words. Word nbins of a
//    row counts that class's overflow pairs (bin index == nbins), row n_cls is a bin for pairs no
//    relation asks for in this pass — so the hot path has no "skip" and no "overflow" branch.
//  * the byte offset of row class(ti,tj) comes from a u32 table in LDS indexed [tj][ti]: tj is
//    wave-uniform (kept in a scalar register), so the lookup is one v_add + one ds_read_b32 and works
//    for any number of types.
//  * binning: table-free guess with an exact guard band (see sweep_fast below).
//  * tiles are read as two 16-byte LDS loads per j atom (type in the 4th double).
// ------------------------------------------------------------------------------------------------

struct FastCtx {
    unsigned *hist;               // LDS offset 0
    const double *edges;          // global memory, nbins+2 entries, last = +inf
    const unsigned *rowtab_me;    // LDS: &rowtab[0][ti] of the table [n_tj][n_ti] -> LDS byte address of the class row
    float gscale, near, near2;    // guard band half-width and its double
    unsigned rowbase_me;          // MODE 2: LDS byte address of row (ti, 0) of the ordered-pair histogram
    unsigned lds_base;            // LDS byte address of the histogram
    int nbins;
};

template <bool DIAG, int U, int MODE>
__device__ __forceinline__ void sweep_group(const double4 *__restrict__ tile, int j0, double xi, double yi,
                                            double zi, double Lx, double Ly, double Lz, double rc2,
                                            const FastCtx &c, int lane_id)
{
    {
        double rsq[U];
        unsigned row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double4 pj = tile[j0 + u];
            const double ax = wrap_abs(xi - pj.x, Lx);
            const double ay = wrap_abs(yi - pj.y, Ly);
            const double az = wrap_abs(zi - pj.z, Lz);
            rsq[u] = (ax * ax + ay * ay) + az * az;
            // 4th double: word offset tj*n_ti into the row table; the byte offset of the class row is read
            // here, unconditionally, so that its LDS latency is hidden behind the rsq chains
            row[u] = c.rowtab_me[(int)__double_as_longlong(pj.w)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool in = rsq[u] < rc2;
            if (DIAG) in = in && (j0 + u > lane_id);
            if (in) {
                int k;
                if (MODE == 0) {
                    // g1 = sqrt(rsq)/ddr + near, evaluated in f32: |error| < nbins*2.9e-7 (cvt 2^-25 after the
                    // sqrt, v_sqrt_f32 1 ulp, rounded 1/ddr 2^-24, the fma 2^-24). near = nbins*1e-6 + 1e-5
                    // is > 3x that bound. If fract(g1) >= 2*near the true value is at least `near` - error
                    // away from both neighbouring integers, so trunc(g1) is the reference bin; otherwise
                    // (~0.1 % of pairs) the exact edge table decides.
                    const float g1 = __builtin_fmaf(__builtin_amdgcn_sqrtf((float)rsq[u]), c.gscale, c.near);
                    k = (int)g1;
                    if (__builtin_amdgcn_fractf(g1) < c.near2) {
                        // g1 is within 2*near above the integer k: the true bin is k or k - 1 (|error| < near),
                        // and the exact edge of k decides
                        k = k > c.nbins ? c.nbins : k;
                        k = rsq[u] < c.edges[k] ? k - 1 : k;
                    }
                } else {
                    // CN edge tables (a few sorted cutoffs^2): count the edges at or below rsq
                    k = 0;
                    for (int e = 1; e <= c.nbins; ++e) k += rsq[u] >= c.edges[e] ? 1 : 0;
                }
                // one VALU op for the address (row already holds the absolute LDS byte address of the row),
                // then the LDS increment
                const unsigned addr = ((unsigned)k << 2) + row[u];
                asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
            }
        }
    }
}

template <bool DIAG, int U, int MODE>
__device__ __forceinline__ void sweep_fast(const double4 *__restrict__ tile, double xi, double yi, double zi,
                                           double Lx, double Ly, double Lz, double rc2, const FastCtx &c,
                                           int lane_id)
{
    for (int j0 = 0; j0 < TILE; j0 += U) sweep_group<DIAG, U, MODE>(tile, j0, xi, yi, zi, Lx, Ly, Lz, rc2, c, lane_id);
}

// Culled path: only the 8-atom groups of the j-tile whose bounding box comes within reach of this wave's
// bounding box are swept. `mask` (wave-uniform) has one bit per group.
template <bool DIAG, int U, int MODE>
__device__ __forceinline__ void sweep_masked(const double4 *__restrict__ tile, unsigned mask, double xi,
                                             double yi, double zi, double Lx, double Ly, double Lz, double rc2,
                                             const FastCtx &c, int lane_id)
{
    static_assert(U == 8, "group boxes are built for 8 atoms");
    while (mask) {
        const int g = __builtin_ctz(mask);
        mask &= mask - 1;
        sweep_group<DIAG, U, MODE>(tile, g * U, xi, yi, zi, Lx, Ly, Lz, rc2, c, lane_id);
    }
}

template <bool TRI, int U, int MODE, bool LIST>
__global__ __launch_bounds__(TILE) void pair_hist_fast_kernel(const PairArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    // block -> (XCD, frame group, i-tile, slice of the j list); a block sweeps `fpb` frames of its XCD's
    // share (frames f with f % 8 == xcd) before it flushes, so the merge traffic drops by fpb
    const long long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const long long q = bid >> 3;
    const int fgroup = (int)(q / a.blocks_per_frame);
    const int within = (int)(q % a.blocks_per_frame);
    const int I = within % a.nTi;
    const int split = within / a.nTi;
    int t_begin = 0, t_end = 0;
    if (!LIST) {
        if (TRI) {
            const int S = tri_shifts(a.nTi, I);
            t_begin = (int)((long long)split * S / a.jsplit);
            t_end = (int)((long long)(split + 1) * S / a.jsplit);
        } else {
            t_begin = (int)((long long)split * a.nTj / a.jsplit);
            t_end = (int)((long long)(split + 1) * a.nTj / a.jsplit);
        }
        if (t_begin >= t_end) return;
    }
    if ((fgroup * a.fpb) * 8 + xcd >= a.n_frames) return;

    // ---- LDS carve-up: hist | tiles | group boxes | row table ----
    // (the exact edge table stays in global memory: only the ~0.1 % guard-band pairs read it, and keeping
    //  its 3 KB out of LDS is what lets a fourth block fit on a CU at 400 bins x 11 classes)
    const int row_len = a.nbins + 1;
    const int hist_words = (a.n_cls + 1) * row_len;
    unsigned *s_hist = reinterpret_cast<unsigned *>(smem);
    size_t off = ((size_t)hist_words * 4 + 15) & ~size_t(15);
    double4 *s_tile = reinterpret_cast<double4 *>(smem + off);
    off += sizeof(double4) * 2 * TILE;
    float4 *s_sph = reinterpret_cast<float4 *>(smem + off);  // [2][32][2] group boxes of the staged j-tiles
    off += LIST ? sizeof(float4) * 4 * (TILE / 8) : 0;
    double *s_edges = reinterpret_cast<double *>(smem + off);  // CN mode only: its few edges are read per candidate
    off += MODE == 1 ? (size_t)(a.nbins + 2) * 8 : 0;
    unsigned *s_row = reinterpret_cast<unsigned *>(smem + off);

    // LDS byte address of the histogram (dynamic LDS starts after any static LDS of the kernel)
    const unsigned lds_base =
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)smem;
    for (int k = tid; k < hist_words; k += TILE) s_hist[k] = 0u;
    for (int k = tid; k < a.n_ti * a.n_tj; k += TILE) {
        const int ti = k % a.n_ti, tj = k / a.n_ti;
        const unsigned cl = a.cls[ti * a.n_tj + tj];
        s_row[k] = lds_base + (cl == 0xFFu ? (unsigned)a.n_cls : cl) * (unsigned)row_len * 4u;
    }

    FastCtx c;
    c.hist = s_hist;
    c.edges = a.edges;  // global, nbins+2 entries, last = +inf
    if (MODE == 1) {
        for (int k = tid; k <= a.nbins + 1; k += TILE) s_edges[k] = a.edges[k];
        c.edges = s_edges;
    }
    c.gscale = a.gscale;
    c.near = (float)a.nbins * 1.0e-6f + 1.0e-5f;
    c.near2 = 2.0f * c.near;
    c.nbins = a.nbins;

    const unsigned short *row_list = nullptr;
    auto tile_of = [&](int t) -> int {
        if (LIST) return (int)row_list[t];
        if (TRI) {
            int J = I + t;
            return J >= a.nTi ? J - a.nTi : J;
        }
        return t;
    };
    const int n_ti = a.n_ti;
    auto pack = [n_ti](const JAtom &p) -> double4 {
        return make_double4(p.x, p.y, p.z, __longlong_as_double((long long)p.t * n_ti));
    };

    int f_last = 0;
    for (int kf = 0; kf < a.fpb; ++kf) {
        const int f = (fgroup * a.fpb + kf) * 8 + xcd;
        if (f >= a.n_frames) break;
        f_last = f;
        if (LIST) {  // this i-tile's neighbour tiles in this frame, sliced over the j-splits
            const long long rowid = (long long)f * a.nTi + I;
            const int cnt = a.list_cnt[rowid];
            row_list = a.list + rowid * a.nTi;
            t_begin = (int)((long long)split * cnt / a.jsplit);
            t_end = (int)((long long)(split + 1) * cnt / a.jsplit);
            if (t_begin >= t_end) continue;  // block-uniform
        }
        const double *xi_f = a.xi + (long long)f * 3 * a.ni;
        const double *xj_f = a.xj + (long long)f * 3 * a.nj;
        const int *ti_f = a.ti + (long long)f * a.ti_fs;
        const int *tj_f = a.tj + (long long)f * a.tj_fs;
        const double Lx = a.box[3 * f], Ly = a.box[3 * f + 1], Lz = a.box[3 * f + 2];
        const JAtom me = load_atom(xi_f, ti_f, a.ni, (long long)I * TILE + tid, PAD_I);
        c.rowtab_me = s_row + me.t;
        // culled path: this wave's bounding box and the group boxes of the j-tiles
        float4 wlo = make_float4(0.f, 0.f, 0.f, 0.f), whi = wlo;
        const float4 *gs_f = nullptr;
        float4 nsp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LIST) {
            const long long w = ((long long)f * a.nTi + I) * (TILE / 64) + (tid >> 6);
            wlo = a.wsph[2 * w];
            whi = a.wsph[2 * w + 1];
            gs_f = a.gsph + (long long)f * a.nTi * (TILE / 8) * 2;
            if (tid < TILE / 4) nsp = gs_f[(long long)tile_of(t_begin) * (TILE / 4) + tid];
        }

        JAtom nxt = load_atom(xj_f, tj_f, a.nj, (long long)tile_of(t_begin) * TILE + tid, PAD_J);
        __syncthreads();  // tables ready (first frame) / previous frame's last tile fully read
        s_tile[tid] = pack(nxt);
        if (LIST && tid < TILE / 4) s_sph[tid] = nsp;
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const int buf = (t - t_begin) & 1;
            if (t + 1 < t_end) {
                nxt = load_atom(xj_f, tj_f, a.nj, (long long)tile_of(t + 1) * TILE + tid, PAD_J);
                if (LIST && tid < TILE / 4) nsp = gs_f[(long long)tile_of(t + 1) * (TILE / 4) + tid];
            }
            const double4 *cur = s_tile + buf * TILE;
            const bool diag = LIST ? (tile_of(t) == I) : (TRI && t == 0);
            if (LIST) {
                // lanes 0..31 (and their mirror 32..63) test one group box each against the wave's box
                const float4 glo = s_sph[buf * (TILE / 4) + 2 * (tid & 31)];
                const float4 ghi = s_sph[buf * (TILE / 4) + 2 * (tid & 31) + 1];
                const float gx = gapf(wlo.x, whi.x, glo.x, ghi.x, (float)Lx);
                const float gy = gapf(wlo.y, whi.y, glo.y, ghi.y, (float)Ly);
                const float gz = gapf(wlo.z, whi.z, glo.z, ghi.z, (float)Lz);
                const bool keep = wlo.w > 0.f && glo.w > 0.f && gx * gx + gy * gy + gz * gz < a.reach * a.reach;
                const unsigned mask = (unsigned)__builtin_amdgcn_ballot_w64(keep);
                if (diag)
                    sweep_masked<true, U, MODE>(cur, mask, me.x, me.y, me.z, Lx, Ly, Lz, a.rc2, c, tid);
                else
                    sweep_masked<false, U, MODE>(cur, mask, me.x, me.y, me.z, Lx, Ly, Lz, a.rc2, c, tid);
            } else if (diag) {
                sweep_fast<true, U, MODE>(cur, me.x, me.y, me.z, Lx, Ly, Lz, a.rc2, c, tid);
            } else {
                sweep_fast<false, U, MODE>(cur, me.x, me.y, me.z, Lx, Ly, Lz, a.rc2, c, tid);
            }
            if (t + 1 < t_end) {
                s_tile[(buf ^ 1) * TILE + tid] = pack(nxt);
                if (LIST && tid < TILE / 4) s_sph[(buf ^ 1) * (TILE / 4) + tid] = nsp;
            }
            __syncthreads();
        }
    }

    // ---- flush: real classes -> global histogram rows, word nbins of every row -> overflow ----
    // the LDS increments are inline asm the compiler does not count: drain them before the last barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int out_words = a.n_cls * a.nbins;
    unsigned long long *g =
        a.hist + (size_t)(a.per_frame ? f_last : (int)(bid % a.slots)) * (size_t)out_words;
    unsigned ovf = 0;
    for (int w = tid; w < hist_words; w += TILE) {
        const unsigned v = s_hist[w];
        if (!v) continue;
        const int cl = w / row_len, k = w - cl * row_len;
        if (k == a.nbins)
            ovf += v;
        else if (cl < a.n_cls)
            atomicAdd(&g[(size_t)cl * a.nbins + k], (unsigned long long)v);
    }
    if (ovf) atomicAdd(a.overflow, (unsigned long long)ovf);
}

size_t lds_bytes_fast(int nbins, int n_cls, int n_ti, int n_tj)
{
    size_t off = ((size_t)(n_cls + 1) * (nbins + 1) * 4 + 15) & ~size_t(15);
    off += sizeof(double4) * 2 * TILE;
    off += sizeof(float4) * 4 * (TILE / 8);  // group boxes (culled path)
    off += nbins <= 64 ? (size_t)(nbins + 2) * 8 : 0;  // CN mode keeps its edge table in LDS
    off += (size_t)n_ti * n_tj * 4;
    return (off + 15) & ~size_t(15);
}

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

constexpr int MORTON_BITS = 5;                       // 32 cells per axis
constexpr int MORTON_CELLS = 1 << (3 * MORTON_BITS); // 32768

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

// 4th double of a sorted record: low word = type * n_ti (word offset into the [tj][ti] row table), high word =
// float(near + type * row_len), the addend of the bin guess that carries the row of the ordered-pair layout
// (MODE 2 of the scalar-j kernel; 0 otherwise).
__device__ __forceinline__ double pack_w(int t, int n_ti, float near, int row_len)
{
    const unsigned lo = (unsigned)(t * n_ti);
    const unsigned hi = row_len > 0 ? __float_as_uint(near + (float)(t * row_len)) : 0u;
    return __hiloint2double((int)hi, (int)lo);
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
__global__ __launch_bounds__(SORT_THREADS) void cull_sort_lds_kernel(
    const double *__restrict__ xyz, const int *__restrict__ type, long long type_fs,
    const double *__restrict__ box, long long n, unsigned short *__restrict__ keys, double *__restrict__ sxyz,
    int *__restrict__ stype, double4 *__restrict__ aos, long long n_pad, int n_ti, float near, int row_len)
{
    extern __shared__ unsigned s_cells[];  // [MORTON_CELLS] + 3 x 16 doubles of scratch behind it
    double *s_red = reinterpret_cast<double *>(s_cells + MORTON_CELLS);
    __shared__ unsigned s_part[SORT_THREADS];
    const int f = blockIdx.x, tid = threadIdx.x;
    const double *x = xyz + (size_t)f * 3 * n;
    const double L[3] = {box[3 * f], box[3 * f + 1], box[3 * f + 2]};
    for (int k = tid; k < MORTON_CELLS; k += SORT_THREADS) s_cells[k] = 0u;
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
        atomicAdd(&s_cells[key], 1u);
    }
    __syncthreads();
    // ---- exclusive scan of the populations ----
    constexpr int PER = MORTON_CELLS / SORT_THREADS;
    const int base = tid * PER;
    unsigned sum = 0;
    for (int k = 0; k < PER; ++k) sum += s_cells[base + k];
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < SORT_THREADS; d <<= 1) {
        const unsigned add = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    unsigned run = tid ? s_part[tid - 1] : 0u;
    for (int k = 0; k < PER; ++k) {
        const unsigned v = s_cells[base + k];
        s_cells[base + k] = run;
        run += v;
    }
    __syncthreads();
    // ---- scatter ----
    for (long long i = tid; i < n; i += SORT_THREADS) {
        const unsigned pos = atomicAdd(&s_cells[kf[i]], 1u);
        const double px = x[i], py = x[n + i], pz = x[2 * n + i];
        const int t = type[(size_t)f * type_fs + i];
        if (sxyz) {
            double *o = sxyz + (size_t)f * 3 * n;
            o[pos] = px;
            o[n + pos] = py;
            o[2 * n + pos] = pz;
            stype[(size_t)f * n + pos] = t;
        }
        aos[(size_t)f * n_pad + pos] = make_double4(px, py, pz, pack_w(t, n_ti, near, row_len));
    }
    for (long long i = n + tid; i < n_pad; i += SORT_THREADS)
        aos[(size_t)f * n_pad + i] = make_double4(PAD_J, PAD_J, PAD_J, __longlong_as_double(0LL));
}

// scatter atoms to their sorted position (cells[] holds running offsets)
__global__ void cull_scatter_kernel(const double *__restrict__ xyz, const int *__restrict__ type,
                                    long long type_fs, long long n, const unsigned short *__restrict__ keys,
                                    unsigned *__restrict__ cells, double *__restrict__ sxyz,
                                    int *__restrict__ stype, double4 *__restrict__ aos, long long n_pad,
                                    int n_ti, float near, int row_len)
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
    aos[(size_t)f * n_pad + pos] = make_double4(px, py, pz, pack_w(t, n_ti, near, row_len));
    // the pad records behind the last atom (never in cutoff: rsq overflows to +inf)
    if (i < n_pad - n) aos[(size_t)f * n_pad + n + i] = make_double4(PAD_J, PAD_J, PAD_J, __longlong_as_double(0LL));
}

// Bounding boxes of one tile of the sorted records, coordinates as given (not wrapped), all three levels
// in one pass: bbox[f][tile][6] (doubles, min xyz / max xyz, for the tile-pair lists), and in f32, widened
// so that rounding can only make them larger, the boxes of every 8 consecutive atoms (one step of the pair
// sweep) and of every 64 (the i atoms of one wave): boxes[2*g] = (lo.xyz, 1), boxes[2*g+1] = (hi.xyz, 1);
// groups without atoms get w = 0 (never within reach).
__global__ __launch_bounds__(TILE) void cull_boxes_kernel(const double4 *__restrict__ aos,
                                                          const double *__restrict__ box, long long n, int nT,
                                                          double *__restrict__ bbox, float4 *__restrict__ gboxes,
                                                          float4 *__restrict__ wboxes)
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
    auto fold = [&](int m) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            lo[ax] = __builtin_fmin(lo[ax], __shfl_xor(lo[ax], m, 64));
            hi[ax] = __builtin_fmax(hi[ax], __shfl_xor(hi[ax], m, 64));
        }
    };
    fold(1);
    fold(2);
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
        widened(l4, h4);
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
    for (int h = 0; h < 8 / U; ++h) {
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
    const float4 *gb_f = a.gsph + (long long)f * a.nTj * (TILE / 8) * 2;  // group boxes of the j set
    const double4 *ats_j = a.aos_j + (long long)f * n_pad_j;
    const float fLx = (float)L.Lx, fLy = (float)L.Ly, fLz = (float)L.Lz;
    for (int t = t_begin; t < t_end; ++t) {
        const int J = __builtin_amdgcn_readfirstlane((int)row_list[t]);
        // lanes 0..31 (mirrored in 32..63) test one 8-atom group box each against this wave's box
        const float4 glo = gb_f[((long long)J * (TILE / 8) + (lane & 31)) * 2];
        const float4 ghi = gb_f[((long long)J * (TILE / 8) + (lane & 31)) * 2 + 1];
        const float gx = gapf(wlo.x, whi.x, glo.x, ghi.x, fLx);
        const float gy = gapf(wlo.y, whi.y, glo.y, ghi.y, fLy);
        const float gz = gapf(wlo.z, whi.z, glo.z, ghi.z, fLz);
        const bool keep = wlo.w > 0.f && glo.w > 0.f && gx * gx + gy * gy + gz * gz < a.reach * a.reach;
        const double4 *tile = ats_j + (long long)J * TILE;
        if (a.tri && J == I) {
            unsigned mask = (unsigned)__builtin_amdgcn_ballot_w64(keep);
            while (mask) {
                const int g = __builtin_ctz(mask);
                mask &= mask - 1;
                sweep_group_sj<true, MODE, 7>(tile + g * 8, g * 8, me.x, me.y, me.z, L, a.rc2, c, lane_in_tile);
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
        unsigned m8 = (unsigned)__builtin_amdgcn_ballot_w64(keep && none);
        while (m8) {
            const int g = __builtin_ctz(m8);
            m8 &= m8 - 1;
            sweep_group_sj<false, MODE, 8>(tile + g * 8, g * 8, me.x, me.y, me.z, L, a.rc2, c, lane_in_tile);
        }
        if (!__builtin_amdgcn_ballot_w64(keep && !none)) continue;
        const unsigned xm = (unsigned)__builtin_amdgcn_ballot_w64(cx == 1u), xp = (unsigned)__builtin_amdgcn_ballot_w64(cx == 2u);
        const unsigned ym = (unsigned)__builtin_amdgcn_ballot_w64(cy == 1u), yp = (unsigned)__builtin_amdgcn_ballot_w64(cy == 2u);
        const unsigned zm = (unsigned)__builtin_amdgcn_ballot_w64(cz == 1u), zp = (unsigned)__builtin_amdgcn_ballot_w64(cz == 2u);
#define SJ_VARIANT(A)                                                                                        \
    {                                                                                                        \
        unsigned mk = (unsigned)__builtin_amdgcn_ballot_w64(keep && !none && amb == (A));                    \
        while (mk) {                                                                                         \
            const int g = __builtin_ctz(mk);                                                                 \
            mk &= mk - 1;                                                                                    \
            AxisL S = L;                                                                                     \
            S.sx = ((xm >> g) & 1u) ? -L.Lx : ((xp >> g) & 1u) ? L.Lx : 0.0;                                 \
            S.sy = ((ym >> g) & 1u) ? -L.Ly : ((yp >> g) & 1u) ? L.Ly : 0.0;                                 \
            S.sz = ((zm >> g) & 1u) ? -L.Lz : ((zp >> g) & 1u) ? L.Lz : 0.0;                                 \
            sweep_group_sj<false, MODE, (A)>(tile + g * 8, g * 8, me.x, me.y, me.z, S, a.rc2, c, lane_in_tile); \
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

__global__ void reduce_slots_kernel(const unsigned long long *__restrict__ in,
                                    unsigned long long *__restrict__ out, int words, int slots)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= words) return;
    unsigned long long s = 0;
    for (int r = 0; r < slots; ++r) s += in[(size_t)r * words + k];
    out[k] = s;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

struct PairProblem {
    int64_t n_frames;
    int64_t ni, nj;
    const double *d_xi, *d_xj;  // device
    const int *d_ti, *d_tj;     // device compact type index
    int64_t ti_fs, tj_fs;
    const double *d_box;        // device [F][3]
    const double *h_box;        // host   [F][3]
    bool tri;
    int n_ti, n_tj;
    std::vector<unsigned char> cls;  // [n_ti][n_tj] -> class id (< n_cls)
    int n_cls;
    int nbins;
    const double *edges;  // host [nbins+1]
    double rc2;
    float gscale;
    int per_frame;
};

size_t lds_bytes(int nbins, int n_cls, int n_ti, int n_tj)
{
    size_t off = (((size_t)(nbins + 2) * 8) + 15) & ~size_t(15);
    off += sizeof(JAtom) * 2 * TILE;
    off += (size_t)n_cls * nbins * 4;
    off += 16;
    off += (size_t)n_ti * n_tj;
    return (off + 15) & ~size_t(15);
}

// Spatial sort + bounding boxes of ONE atom set of a batch of frames (culled path): Hilbert-sorted records
// `aos` [F][nT*256], tile boxes `bbox` [F][nT][6], 8-atom and 64-atom boxes `gs` / `ws`. `slot` = workspace ids of
// {records, tile boxes, group boxes, wave boxes}; keys, cell counters and the SoA copy are shared scratch.
struct SortedSet {
    const double4 *aos = nullptr;
    const double *bbox = nullptr;
    const float4 *gs = nullptr, *ws = nullptr;
    const double *sx = nullptr;
    const int *st = nullptr;
};

int cull_prepare_set(mdhip_ctx *ctx, int64_t F, const double *d_x, const int *d_t, long long t_fs,
                     const double *d_box, long long N, int nT, int n_ti, float near, int row_len, bool want_soa,
                     const int slot[4], SortedSet &out)
{
    MD_WS(d_sx, double, WS_SORT_XYZ, want_soa ? (size_t)F * 3 * N * 8 : 64);
    MD_WS(d_st, int, WS_SORT_TYPE, want_soa ? (size_t)F * N * 4 : 64);
    MD_WS(d_keys, unsigned short, WS_KEYS, (size_t)F * N * 2);
    MD_WS(d_cells, unsigned, WS_CELLS, (size_t)F * MORTON_CELLS * 4);
    MD_WS(d_ao, double4, slot[0], (size_t)F * nT * TILE * sizeof(double4));
    MD_WS(d_bbox, double, slot[1], (size_t)F * nT * 6 * 8);
    MD_WS(d_gs, float4, slot[2], (size_t)F * nT * (TILE / 8) * 2 * sizeof(float4));
    MD_WS(d_ws, float4, slot[3], (size_t)F * nT * (TILE / 64) * 2 * sizeof(float4));
    const dim3 ga((unsigned)((N + 255) / 256), (unsigned)F);
    // one block per frame with the cell counters in LDS when there are frames enough to fill the chip (or the
    // frames are small); the multi-block path with global counters otherwise
    const size_t sort_lds = (size_t)MORTON_CELLS * 4 + 3 * 16 * 8;
    const bool lds_sort = ctx->opt_rdf_sort != 0 && sort_lds + 8192 <= ctx->lds_max && N <= 262144 &&
                          (ctx->opt_rdf_sort == 1 || F >= ctx->cu_count / 4 || N <= 16384);
    if (lds_sort) {
        MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(cull_sort_lds_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_lds));
        hipLaunchKernelGGL(cull_sort_lds_kernel, dim3((unsigned)F), dim3(SORT_THREADS), sort_lds, ctx->stream, d_x,
                           d_t, t_fs, d_box, N, d_keys, want_soa ? d_sx : (double *)nullptr, d_st, d_ao,
                           (long long)nT * TILE, n_ti, near, row_len);
    } else {
        MD_HIP(hipMemsetAsync(d_cells, 0, (size_t)F * MORTON_CELLS * 4, ctx->stream));
        MD_WS(d_org, double, WS_ORIGIN, (size_t)F * 3 * 8);
        hipLaunchKernelGGL(cull_origin_kernel, dim3((unsigned)F), dim3(256), 0, ctx->stream, d_x, N, d_org);
        hipLaunchKernelGGL(cull_keys_kernel, ga, dim3(256), 0, ctx->stream, d_x, d_box, N, d_org, d_keys, d_cells);
        hipLaunchKernelGGL(cull_scan_kernel, dim3((unsigned)F), dim3(256), 0, ctx->stream, d_cells);
        hipLaunchKernelGGL(cull_scatter_kernel, ga, dim3(256), 0, ctx->stream, d_x, d_t, t_fs, N, d_keys, d_cells,
                           want_soa ? d_sx : (double *)nullptr, d_st, d_ao, (long long)nT * TILE, n_ti, near,
                           row_len);
    }
    hipLaunchKernelGGL(cull_boxes_kernel, dim3((unsigned)nT, (unsigned)F), dim3(TILE), 0, ctx->stream, d_ao, d_box, N,
                       nT, d_bbox, d_gs, d_ws);
    MD_HIP(hipGetLastError());
    out.aos = d_ao;
    out.bbox = d_bbox;
    out.gs = d_gs;
    out.ws = d_ws;
    out.sx = d_sx;
    out.st = d_st;
    return MDHIP_OK;
}

// Runs the kernel over one batch of frames (in several passes when the class rows do not fit LDS) and
// returns the class histograms on the host: H [F|1][n_cls][nbins], overflow count.
int pair_hist_run_batch(mdhip_ctx *ctx, const PairProblem &p, std::vector<uint64_t> &H, uint64_t *overflow)
{
    const int64_t F = p.n_frames;
    const int nTi = (int)((p.ni + TILE - 1) / TILE);
    const int nTj = (int)((p.nj + TILE - 1) / TILE);
    const size_t out_frames = p.per_frame ? (size_t)F : 1;
    H.assign(out_frames * p.n_cls * p.nbins, 0);
    *overflow = 0;
    if (F == 0 || p.ni == 0 || p.nj == 0) return MDHIP_OK;

    // kernel variant: 0 = reference-shaped loops with an edge-table lookup per pair (always used for CN
    // edge tables, gscale == 0); 1 = fast kernel (table-free binning with an exact guard band)
    const bool mode_cn = !(p.gscale > 0.f);  // CN edge table: a few sorted cutoffs^2, bins found by counting
    const bool fast = ctx->opt_rdf_variant == 1 && (mode_cn ? p.nbins <= 64 : p.nbins <= 100000);

    // spatial culling: worth it when the cutoff sphere is a small part of the box (atoms x sites: scalar-j
    // kernel only)
    bool cull = false;
    if (fast && nTi >= 8 && nTi <= 65535 && nTj <= 65535 && ctx->opt_rdf_cull != 0 &&
        (p.tri || (ctx->opt_rdf_sj != 0 && nTj >= 2))) {
        const double V = p.h_box[0] * p.h_box[1] * p.h_box[2];
        const double edge = 0.5 * (std::cbrt((double)TILE * V / (double)p.ni) +
                                   std::cbrt((double)TILE * V / (double)p.nj));
        const double reach = std::sqrt(p.rc2) + 0.8 * edge;
        const double est = 4.18879 * reach * reach * reach / V;  // share of tile pairs that survive
        cull = ctx->opt_rdf_cull == 1 || est < 1.5;  // measured: still +5 % at est = 1.08 (BASELINE C2)
    }

    // classes per pass limited by LDS (keep >= 2 blocks per CU when possible)
    const size_t lds_cap = ctx->lds_max > 0 ? ctx->lds_max : 65536;
    const size_t fixed = fast ? lds_bytes_fast(p.nbins, 0, p.n_ti, p.n_tj) : lds_bytes(p.nbins, 0, p.n_ti, p.n_tj);
    const size_t row_b = fast ? (size_t)(p.nbins + 1) * 4 : (size_t)p.nbins * 4;
    if (fixed + row_b > lds_cap)
        return mdhip_fail(ctx, MDHIP_ELIMIT, "pair_hist: %d bins do not fit LDS (%zu B)", p.nbins,
                          lds_cap);
    const size_t budget = lds_cap / 2 > fixed + row_b ? lds_cap / 2 : lds_cap;
    int cls_per_pass = (int)((budget - fixed) / row_b);
    if (cls_per_pass > p.n_cls) cls_per_pass = p.n_cls;
    if (cls_per_pass > 250) cls_per_pass = 250;

    // Ordered-pair rows (MODE 2 of the scalar-j kernel): one LDS row per (ti, tj) addressed without a table —
    // the row offset rides in the addend of the bin guess. Needs all n_ti^2 rows in LDS at >= 4 blocks per CU
    // and all classes in one pass.
    const size_t ord_b = lds_bytes_sj_ordered(p.nbins, p.n_ti, p.n_tj);
    const bool ordered = cull && ctx->opt_rdf_sj != 0 && !mode_cn && p.tri && ctx->opt_rdf_rows != 0 &&
                         p.n_cls <= 250 && ord_b <= lds_cap / 4 && (double)p.n_tj * (p.nbins + 1) < 65536.0;
    float near_ord = 0.f;
    if (ordered) {
        cls_per_pass = p.n_cls;
        // |error| of the f32 guess g = fma(sqrt((float)rsq), 1/ddr, near + tj*row_len): relative 2^-25 (conversion,
        // halved by the root) + 2^-23 (v_sqrt_f32, 1 ulp) + 2^-24 (rounded 1/ddr) = 2.1e-7 of the bin number, plus
        // half an ulp of the largest value each for the rounding of the addend and of the fma. near = 2 x that.
        const double maxg = (double)p.n_tj * (p.nbins + 1) + 1.0;  // the addend carries tj * row_len only
        const double ulp = std::ldexp(1.0, (int)std::floor(std::log2(maxg)) - 23);
        near_ord = (float)(2.0 * ((double)p.nbins * 2.1e-7 + ulp) + 1.0e-5);
    }
    const int n_pass = (p.n_cls + cls_per_pass - 1) / cls_per_pass;

    // geometry
    int max_list = p.tri ? tri_shifts(nTi, 0) : nTj;
    int jsplit = ctx->opt_rdf_jsplit;
    if (jsplit <= 0) {
        const int64_t want = (int64_t)ctx->cu_count * 48;  // ~12 blocks per CU slot: short tail
        const int64_t base = (int64_t)nTi * F;
        jsplit = (int)((want + base - 1) / base);
    }
    if (jsplit > max_list) jsplit = max_list;
    if (cull && jsplit > 4) jsplit = 4;
    // persistent scalar-j kernel: items are (frame, tile, wave, slice); 4 slices measured best at C2 and C3
    if (cull && ctx->opt_rdf_sj == 1 && !p.per_frame && ctx->opt_rdf_jsplit <= 0) jsplit = std::min(4, max_list);
    if (jsplit < 1) jsplit = 1;
    const int blocks_per_frame = nTi * jsplit;
    // frames per block (fast kernel, frame-summed output): as many as keeps >= `want` blocks in flight
    int fpb = 1;
    if (fast && !p.per_frame) {
        fpb = ctx->opt_rdf_fpb;
        if (fpb <= 0) {
            const int64_t want = (int64_t)ctx->cu_count * 24;
            fpb = (int)(((int64_t)blocks_per_frame * F) / want);
        }
        if (fpb < 1) fpb = 1;
        if (fpb > 64) fpb = 64;
    }
    const int64_t fgroups = (F + 8LL * fpb - 1) / (8LL * fpb);
    const int64_t grid = fgroups * 8 * blocks_per_frame;
    if (grid > 0x7fffffffLL)
        return mdhip_fail(ctx, MDHIP_ELIMIT, "pair_hist: grid of %lld blocks is too large",
                          (long long)grid);
    int slots = p.per_frame ? 1 : ctx->opt_rdf_slots;

    // device tables
    const size_t edges_b = (size_t)(p.nbins + 2) * 8;  // + a +inf sentinel after the last edge
    const size_t cls_b = ((size_t)p.n_ti * p.n_tj + 63) & ~size_t(63);
    // edges and the class table of every pass: one pinned staging buffer, one H2D copy
    const size_t tab_b = edges_b + (size_t)n_pass * cls_b;
    MD_WS(d_tab, unsigned char, WS_TABLES, tab_b);
    MD_PIN(h_tab, unsigned char, PIN_TABLES, tab_b);
    {
        double *e = reinterpret_cast<double *>(h_tab);
        std::copy(p.edges, p.edges + p.nbins + 1, e);
        e[p.nbins + 1] = std::numeric_limits<double>::infinity();
        for (int pass = 0; pass < n_pass; ++pass) {
            const int c0 = pass * cls_per_pass;
            const int nc = (p.n_cls - c0) < cls_per_pass ? (p.n_cls - c0) : cls_per_pass;
            unsigned char *t = h_tab + edges_b + (size_t)pass * cls_b;
            for (size_t k = 0; k < (size_t)p.n_ti * p.n_tj; ++k) {
                const int c = p.cls[k];
                t[k] = (c >= c0 && c < c0 + nc) ? (unsigned char)(c - c0) : 0xFF;
            }
        }
    }
    MD_HIP(hipMemcpyAsync(d_tab, h_tab, tab_b, hipMemcpyHostToDevice, ctx->stream));
    MD_WS(d_misc, unsigned long long, WS_MISC, 64);
    MD_HIP(hipMemsetAsync(d_misc, 0, 64, ctx->stream));

    // ---- culled path: Morton sort, tile boxes, neighbour-tile lists (once, shared by all class passes) ----
    const double *k_xi = p.d_xi, *k_xj = p.d_xj;
    const int *k_ti = p.d_ti, *k_tj = p.d_tj;
    long long k_ti_fs = p.ti_fs, k_tj_fs = p.tj_fs;
    const unsigned short *d_list = nullptr;
    const int *d_list_cnt = nullptr;
    const float4 *d_gsph = nullptr, *d_wsph = nullptr;
    const double4 *d_aos = nullptr;
    double prep_ms = 0.0;
    bool prep_timed = false;
    const double4 *d_aos_j = nullptr;
    if (cull) {
        const long long N = p.ni;
        const bool want_soa = ctx->opt_rdf_sj == 0;  // only the LDS-tile kernel (atom-atom) reads the SoA copy
        MD_WS(d_l, unsigned short, WS_LIST, (size_t)F * nTi * (p.tri ? nTi : nTj) * 2);
        MD_WS(d_lc, int, WS_LISTCNT, (size_t)F * nTi * 4);
        KernelTimer ptimer(ctx, 1, true);  // second event pair: collected after the pair kernel's sync
        SortedSet si, sj_set;
        const int slot_i[4] = {WS_SORT_AOS, WS_BBOX, WS_GSPH, WS_WSPH};
        int rc = cull_prepare_set(ctx, F, p.d_xi, p.d_ti, (long long)p.ti_fs, p.d_box, N, nTi, p.n_ti,
                                  ordered ? near_ord : 0.f, ordered ? p.nbins + 1 : 0, want_soa, slot_i, si);
        if (rc) return rc;
        if (p.tri) {
            sj_set = si;
            hipLaunchKernelGGL(cull_list_kernel<true>, dim3((unsigned)nTi, (unsigned)F), dim3(256), 0, ctx->stream,
                               si.bbox, si.bbox, nTi, p.d_box, nTi, p.rc2 * (1.0 + 1e-9) + 1e-9, d_l, d_lc);
        } else {
            const int slot_j[4] = {WS_SORT_AOS_J, WS_BBOX_J, WS_GSPH_J, WS_WSPH_J};
            rc = cull_prepare_set(ctx, F, p.d_xj, p.d_tj, (long long)p.tj_fs, p.d_box, p.nj, nTj, p.n_ti, 0.f, 0,
                                  false, slot_j, sj_set);
            if (rc) return rc;
            hipLaunchKernelGGL(cull_list_kernel<false>, dim3((unsigned)nTi, (unsigned)F), dim3(256), 0, ctx->stream,
                               si.bbox, sj_set.bbox, nTj, p.d_box, nTi, p.rc2 * (1.0 + 1e-9) + 1e-9, d_l, d_lc);
        }
        ptimer.stop();
        MD_HIP(hipGetLastError());
        prep_timed = true;
        d_gsph = sj_set.gs;  // 8-atom boxes of the j set
        d_wsph = si.ws;      // 64-atom boxes of the i set
        if (want_soa) {
            k_xi = k_xj = si.sx;
            k_ti = k_tj = si.st;
            k_ti_fs = k_tj_fs = N;
        }
        d_list = d_l;
        d_list_cnt = d_lc;
        d_aos = si.aos;
        d_aos_j = sj_set.aos;
    }

    double total_ms = 0.0;  // the pair kernel alone; the culling pre-pass is reported separately
    unsigned long long ov = 0;
    int launches = 0;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int c0 = pass * cls_per_pass;
        const int nc = (p.n_cls - c0) < cls_per_pass ? (p.n_cls - c0) : cls_per_pass;
        const size_t words = (size_t)nc * p.nbins;
        const size_t acc_frames = p.per_frame ? (size_t)F : (size_t)slots;
        MD_WS(d_hist, unsigned long long, WS_HIST, (acc_frames + 1) * words * 8);
        MD_HIP(hipMemsetAsync(d_hist, 0, acc_frames * words * 8, ctx->stream));

        PairArgs a;
        a.xi = k_xi;
        a.xj = k_xj;
        a.ti = k_ti;
        a.tj = k_tj;
        a.list = d_list;
        a.list_cnt = d_list_cnt;
        a.gsph = d_gsph;
        a.wsph = d_wsph;
        a.reach = (float)((std::sqrt(p.rc2) + 1e-3) * 1.00001);
        a.aos = d_aos;
        a.aos_j = d_aos_j;
        a.tri = p.tri ? 1 : 0;
        a.box = p.d_box;
        a.cls = d_tab + edges_b + (size_t)pass * cls_b;
        a.edges = reinterpret_cast<const double *>(d_tab);
        a.hist = d_hist;
        a.overflow = d_misc;
        a.ni = p.ni;
        a.nj = p.nj;
        a.ti_fs = k_ti_fs;
        a.tj_fs = k_tj_fs;
        a.rc2 = p.rc2;
        a.gscale = p.gscale;
        a.n_ti = p.n_ti;
        a.n_tj = p.n_tj;
        a.n_cls = nc;
        a.nbins = p.nbins;
        a.n_frames = (int)F;
        a.nTi = nTi;
        a.nTj = nTj;
        a.jsplit = jsplit;
        a.blocks_per_frame = blocks_per_frame;
        a.per_frame = p.per_frame;
        a.slots = slots;
        a.fpb = fpb;

        a.near = near_ord;
        const bool sj = cull && ctx->opt_rdf_sj != 0;  // wave-independent sweep with scalar loads of the j atoms
        const bool persist = sj && !p.per_frame && ctx->opt_rdf_sj != 2;  // resident grid + per-XCD work counters
        a.work = reinterpret_cast<unsigned *>(d_misc + 4);
        const size_t lds = ordered ? ord_b
                           : sj    ? lds_bytes_sj(p.nbins, nc, p.n_ti, p.n_tj, mode_cn)
                           : fast ? lds_bytes_fast(p.nbins, nc, p.n_ti, p.n_tj)
                                  : lds_bytes(p.nbins, nc, p.n_ti, p.n_tj);
        void (*kern)(const PairArgs);
#define MD_PICK(...) (ctx->last_kernel = #__VA_ARGS__, __VA_ARGS__)
        if (!fast)
            kern = p.tri ? MD_PICK(pair_hist_kernel<true>) : MD_PICK(pair_hist_kernel<false>);
        else if (ordered)
            kern = persist ? MD_PICK(pair_hist_sj_kernel<2, true>) : MD_PICK(pair_hist_sj_kernel<2, false>);
        else if (persist)
            kern = mode_cn ? MD_PICK(pair_hist_sj_kernel<1, true>) : MD_PICK(pair_hist_sj_kernel<0, true>);
        else if (sj)
            kern = mode_cn ? MD_PICK(pair_hist_sj_kernel<1, false>) : MD_PICK(pair_hist_sj_kernel<0, false>);
        else if (cull)
            kern = mode_cn ? MD_PICK(pair_hist_fast_kernel<true, 8, 1, true>)
                           : MD_PICK(pair_hist_fast_kernel<true, 8, 0, true>);
        else if (mode_cn)
            kern = p.tri ? MD_PICK(pair_hist_fast_kernel<true, 8, 1, false>)
                         : MD_PICK(pair_hist_fast_kernel<false, 8, 1, false>);
        else
            kern = p.tri ? MD_PICK(pair_hist_fast_kernel<true, 8, 0, false>)
                         : MD_PICK(pair_hist_fast_kernel<false, 8, 0, false>);
#undef MD_PICK
        if (lds > 65536)
            MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        long long launch_grid = grid;
        if (sj) {
            int per_cu = 0;
            MD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), TILE,
                                                                lds));
            if (per_cu < 1) per_cu = 1;
            const long long capacity = (long long)per_cu * ctx->cu_count;
            const long long block_items = (long long)nTi * jsplit;  // groups of 4 wave items per frame
            a.fpb = 1;
            if (persist) {
                launch_grid = std::min(capacity, F * block_items + 8);
                launch_grid = (launch_grid + 7) / 8 * 8;
                MD_HIP(hipMemsetAsync(d_misc + 4, 0, 32, ctx->stream));
            } else {
                // per-frame output: as many blocks per frame as keeps one resident set busy, each flushing once
                // blocks per frame: few frames in flight per XCD (their records should stay in its 4 MB L2: 32 B per
                // atom per frame), every wave still left with a few items to draw
                const long long cap_xcd = std::max<long long>(1, capacity / 8);
                const long long frames_xcd = (F + 7) / 8;
                const long long in_flight = std::max<long long>(1, std::min<long long>(frames_xcd, ctx->opt_rdf_inflight));
                long long bpf = std::max<long long>(1, cap_xcd / in_flight);
                bpf = std::min(bpf, std::max<long long>(1, block_items / 2));
                // every block stores one copy of its LDS histogram: keep that workspace within ~2 GiB
                while (bpf > 8 && (double)(((F + 7) / 8) * 8 * bpf) * (double)lds > 2147483648.0) bpf /= 2;
                a.blocks_per_frame = (int)bpf;
                launch_grid = ((F + 7) / 8) * 8 * bpf;
                MD_WS(d_work, unsigned, WS_WORK, (size_t)F * 4);
                MD_HIP(hipMemsetAsync(d_work, 0, (size_t)F * 4, ctx->stream));
                a.work = d_work;
            }
        }
        // scalar-j kernels: every block stores its LDS histogram into its own slice; a merge kernel adds them up
        const int sj_rows = ordered ? p.n_ti * p.n_tj : nc + 1;
        const int sj_words = sj_rows * (p.nbins + 1);
        unsigned long long *d_rows = nullptr;
        if (sj) {
            MD_WS(d_sl, unsigned, WS_SLICES, (size_t)launch_grid * sj_words * 4);
            d_rows = (unsigned long long *)mdhip_ws(ctx, WS_ROWS, out_frames * (size_t)sj_words * 8);
            if (!d_rows) return MDHIP_ENOMEM;
            if (!p.per_frame) MD_HIP(hipMemsetAsync(d_rows, 0, (size_t)sj_words * 8, ctx->stream));
            a.slices = d_sl;
        }
        KernelTimer timer(ctx);
        hipLaunchKernelGGL(kern, dim3((unsigned)launch_grid), dim3(TILE), lds, ctx->stream, a);
        if (sj) {
            const unsigned gy = p.per_frame ? (unsigned)F : (unsigned)std::min<long long>(64, launch_grid);
            hipLaunchKernelGGL(merge_slices_kernel, dim3((unsigned)((sj_words + 255) / 256), gy), dim3(256), 0,
                               ctx->stream, a.slices, sj_words, launch_grid, p.per_frame, a.blocks_per_frame, d_rows);
        }
        timer.stop();
        MD_HIP(hipGetLastError());

        if (sj) {
            // D2H of the row sums (pinned staging), then rows -> classes and the overflow words on the host
            MD_PIN(hrows, uint64_t, PIN_OUT, out_frames * (size_t)sj_words * 8);
            MD_HIP(hipMemcpyAsync(hrows, d_rows, out_frames * (size_t)sj_words * 8, hipMemcpyDeviceToHost,
                                  ctx->stream));
            MD_HIP(hipStreamSynchronize(ctx->stream));
            timer.collect();
            total_ms += ctx->last_ms;
            ++launches;
            if (prep_timed) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3) == hipSuccess) prep_ms = ms;
                prep_timed = false;
            }
            const int row_len = p.nbins + 1;
            for (size_t fr = 0; fr < out_frames; ++fr)
                for (int r = 0; r < sj_rows; ++r) {
                    const uint64_t *src = hrows + (fr * sj_rows + r) * row_len;
                    ov += src[p.nbins];
                    // ordered rows (ti, tj) -> class of the unordered pair; class rows of this pass -> c0 + r, the
                    // extra row holds the pairs whose class belongs to another pass
                    const int cl = ordered ? (int)p.cls[r] : (r < nc ? c0 + r : -1);
                    if (cl < 0) continue;
                    uint64_t *dst = &H[(fr * p.n_cls + cl) * p.nbins];
                    for (int k = 0; k < p.nbins; ++k) dst[k] += src[k];
                }
            continue;
        }

        unsigned long long *d_final = d_hist;
        if (!p.per_frame && slots > 1) {
            d_final = d_hist + (size_t)slots * words;
            hipLaunchKernelGGL(reduce_slots_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256),
                               0, ctx->stream, d_hist, d_final, (int)words, slots);
            MD_HIP(hipGetLastError());
        }
        // D2H (pinned staging) into the right class rows; the overflow word rides along with the last pass
        MD_PIN(tmp, uint64_t, PIN_OUT, (out_frames * words + 1) * 8);
        MD_HIP(hipMemcpyAsync(tmp, d_final, out_frames * words * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (pass == n_pass - 1)
            MD_HIP(hipMemcpyAsync(tmp + out_frames * words, d_misc, 8, hipMemcpyDeviceToHost, ctx->stream));
        MD_HIP(hipStreamSynchronize(ctx->stream));
        if (pass == n_pass - 1) ov = tmp[out_frames * words];
        timer.collect();
        total_ms += ctx->last_ms;
        ++launches;
        if (prep_timed) {  // the pre-pass ran ahead of the first pass on the same stream: its events are complete
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3) == hipSuccess) prep_ms = ms;
            prep_timed = false;
        }
        for (size_t fr = 0; fr < out_frames; ++fr)
            memcpy(&H[(fr * p.n_cls + c0) * p.nbins], &tmp[fr * words], words * 8);
    }
    // with several passes every in-cutoff overflow pair is seen once per pass
    *overflow = ov / (uint64_t)n_pass;
    ctx->last_ms = total_ms;
    ctx->last_launches = launches;
    ctx->last_aux_ms = prep_ms;
    return MDHIP_OK;
}

// Splits the frames into batches so that the culled path's workspace (sorted copy, keys, cell counts, boxes,
// neighbour-tile lists) stays within ~2 GiB and a launch's grid.y within 65535, and merges the batches.
int pair_hist_run(mdhip_ctx *ctx, const PairProblem &p, std::vector<uint64_t> &H, uint64_t *overflow)
{
    const int64_t F = p.n_frames;
    const int64_t nT = (p.ni + TILE - 1) / TILE;
    const double per_frame_b = 38.0 * (double)p.ni + 4.0 * MORTON_CELLS + 2.0 * (double)nT * (double)nT + 1024.0;
    int64_t batch = (int64_t)(2147483648.0 / per_frame_b);
    if (batch < 1) batch = 1;
    if (batch > 32768) batch = 32768;
    if (ctx->opt_rdf_batch > 0) batch = ctx->opt_rdf_batch;
    if (F <= batch) return pair_hist_run_batch(ctx, p, H, overflow);
    const size_t row = (size_t)p.n_cls * p.nbins;
    H.assign((p.per_frame ? (size_t)F : 1) * row, 0);
    *overflow = 0;
    double ms = 0.0, aux = 0.0;
    int launches = 0;
    std::vector<uint64_t> part;
    for (int64_t f0 = 0; f0 < F; f0 += batch) {
        PairProblem q = p;
        q.n_frames = std::min<int64_t>(batch, F - f0);
        q.d_xi = p.d_xi + (size_t)f0 * 3 * p.ni;
        q.d_xj = p.d_xj + (size_t)f0 * 3 * p.nj;
        q.d_ti = p.d_ti + (size_t)f0 * p.ti_fs;
        q.d_tj = p.d_tj + (size_t)f0 * p.tj_fs;
        q.d_box = p.d_box + (size_t)f0 * 3;
        q.h_box = p.h_box + (size_t)f0 * 3;
        uint64_t ov = 0;
        int rc = pair_hist_run_batch(ctx, q, part, &ov);
        if (rc) return rc;
        *overflow += ov;
        ms += ctx->last_ms;
        aux += ctx->last_aux_ms;
        launches += ctx->last_launches;
        if (p.per_frame)
            std::copy(part.begin(), part.end(), H.begin() + (size_t)f0 * row);
        else
            for (size_t k = 0; k < row; ++k) H[k] += part[k];
    }
    ctx->last_ms = ms;
    ctx->last_aux_ms = aux;
    ctx->last_launches = launches;
    return MDHIP_OK;
}

// labels -> compact indices 0..T-1 (sorted unique labels)
void compact_labels(const int32_t *lab, size_t n, std::vector<int32_t> &uniq, std::vector<int32_t> &idx)
{
    idx.resize(n);
    uniq.clear();
    if (n == 0) return;
    int32_t lo = lab[0], hi = lab[0];
    for (size_t k = 1; k < n; ++k) {
        lo = lab[k] < lo ? lab[k] : lo;
        hi = lab[k] > hi ? lab[k] : hi;
    }
    const int64_t span = (int64_t)hi - (int64_t)lo + 1;
    if (span <= (1 << 20)) {  // the usual case (LAMMPS types 1..T): one presence table, O(n)
        std::vector<int32_t> slot((size_t)span, -1);
        for (size_t k = 0; k < n; ++k) slot[(size_t)(lab[k] - lo)] = 0;
        int32_t next = 0;
        for (int64_t v = 0; v < span; ++v)
            if (slot[(size_t)v] == 0) {
                slot[(size_t)v] = next++;
                uniq.push_back((int32_t)(lo + v));
            }
        for (size_t k = 0; k < n; ++k) idx[k] = slot[(size_t)(lab[k] - lo)];
        return;
    }
    uniq.assign(lab, lab + n);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    for (size_t k = 0; k < n; ++k)
        idx[k] = (int32_t)(std::lower_bound(uniq.begin(), uniq.end(), lab[k]) - uniq.begin());
}

int find_label(const std::vector<int32_t> &uniq, int32_t lab)
{
    auto it = std::lower_bound(uniq.begin(), uniq.end(), lab);
    return (it != uniq.end() && *it == lab) ? (int)(it - uniq.begin()) : -1;
}

// Relations -> classes. Triangular: unordered type pairs {a,b}; rectangular: ordered (atom type,
// site type). rel_cls[kl] = class id, or -1 when a label does not occur in the data (the reference
// then counts nothing); the last class collects every pair no relation asks for.
void build_classes(bool tri, const std::vector<int32_t> &ui, const std::vector<int32_t> &uj, int n_rel,
                   const int32_t *rel, std::vector<unsigned char> &cls, std::vector<int> &rel_cls,
                   int &n_cls)
{
    const int n_ti = (int)ui.size(), n_tj = (int)uj.size();
    std::vector<int> map((size_t)n_ti * n_tj, -1);
    rel_cls.assign(n_rel, -1);
    int next = 0;
    for (int kl = 0; kl < n_rel; ++kl) {
        const int a = find_label(ui, rel[2 * kl]);
        const int b = find_label(uj, rel[2 * kl + 1]);
        if (a < 0 || b < 0) continue;
        int &slot = map[(size_t)a * n_tj + b];
        if (slot < 0) {
            slot = next++;
            if (tri) map[(size_t)b * n_tj + a] = slot;
        }
        rel_cls[kl] = slot;
    }
    n_cls = next + 1;
    cls.resize((size_t)n_ti * n_tj);
    for (size_t k = 0; k < cls.size(); ++k) cls[k] = (unsigned char)(map[k] < 0 ? next : map[k]);
}

struct RelJob {
    bool tri;
    int64_t F, ni, nj;
    const double *xi, *xj;  // host|dev
    int xi_dev, xj_dev;
    const int32_t *lab_i;  // host labels [ni] or [F][ni]
    int64_t lab_i_fs;
    const int32_t *lab_j;  // host labels [nj] (rectangular only)
    const double *box;     // host [F][3]
    int n_rel;
    const int32_t *rel;
    int nbins;
    const double *edges;  // host [nbins+1]
    double rc2;
    float gscale;
    int per_frame;
};

// Stages everything, runs the kernel and returns class histograms + the relation->class map.
int run_job(mdhip_ctx *ctx, const RelJob &j, std::vector<uint64_t> &H, std::vector<int> &rel_cls,
            int &n_cls, uint64_t *overflow)
{
    std::vector<int32_t> ui, idx_i, uj, idx_j;
    const size_t n_lab_i = j.lab_i_fs ? (size_t)j.F * j.ni : (size_t)j.ni;
    compact_labels(j.lab_i, n_lab_i, ui, idx_i);
    if (!j.tri) compact_labels(j.lab_j, (size_t)j.nj, uj, idx_j);
    const std::vector<int32_t> &ujr = j.tri ? ui : uj;
    if (ui.size() * ujr.size() > 16384)
        return mdhip_fail(ctx, MDHIP_ELIMIT, "too many distinct types (%zu x %zu)", ui.size(),
                          ujr.size());

    PairProblem p;
    p.tri = j.tri;
    build_classes(j.tri, ui, ujr, j.n_rel, j.rel, p.cls, rel_cls, n_cls);
    p.n_cls = n_cls;
    p.n_ti = (int)ui.size();
    p.n_tj = (int)ujr.size();

    int rc;
    p.d_xi = (const double *)mdhip_stage(ctx, WS_XYZ_I, j.xi, (size_t)j.F * 3 * j.ni * 8, j.xi_dev, &rc);
    if (rc) return rc;
    // compact types of both sets and the box lengths: one pinned staging buffer (owned by the context, so the
    // asynchronous copies need no sync before this function's vectors go away)
    const size_t ti_b = (idx_i.size() * 4 + 63) & ~size_t(63), tj_b = (idx_j.size() * 4 + 63) & ~size_t(63);
    const size_t box_b = (size_t)j.F * 3 * 8;
    MD_PIN(h_in, unsigned char, PIN_TYPES, ti_b + tj_b + box_b);
    memcpy(h_in, idx_i.data(), idx_i.size() * 4);
    memcpy(h_in + ti_b, idx_j.data(), idx_j.size() * 4);
    memcpy(h_in + ti_b + tj_b, j.box, box_b);
    MD_WS(d_ti, int, WS_TYPE_I, idx_i.size() * 4);
    MD_HIP(hipMemcpyAsync(d_ti, h_in, idx_i.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    p.d_ti = d_ti;
    p.ti_fs = j.lab_i_fs;
    if (j.tri) {
        p.d_xj = p.d_xi;
        p.d_tj = p.d_ti;
        p.tj_fs = p.ti_fs;
        p.nj = j.ni;
    } else {
        p.d_xj = (const double *)mdhip_stage(ctx, WS_XYZ_J, j.xj, (size_t)j.F * 3 * j.nj * 8, j.xj_dev, &rc);
        if (rc) return rc;
        MD_WS(d_tj, int, WS_TYPE_J, idx_j.size() * 4);
        MD_HIP(hipMemcpyAsync(d_tj, h_in + ti_b, idx_j.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        p.d_tj = d_tj;
        p.tj_fs = 0;
        p.nj = j.nj;
    }
    MD_WS(d_box, double, WS_BOX, (size_t)j.F * 3 * 8);
    MD_HIP(hipMemcpyAsync(d_box, h_in + ti_b + tj_b, box_b, hipMemcpyHostToDevice, ctx->stream));
    p.d_box = d_box;
    p.h_box = j.box;
    p.n_frames = j.F;
    p.ni = j.ni;
    p.nbins = j.nbins;
    p.edges = j.edges;
    p.rc2 = j.rc2;
    p.gscale = j.gscale;
    p.per_frame = j.per_frame;
    return pair_hist_run(ctx, p, H, overflow);
}

// CN: edges are the sorted distinct positive cutoffs^2; rank[kl] = number of bins below relation kl's cutoff.
void cn_edges(int n_rel, const double *rc2, std::vector<double> &edges, std::vector<int> &rank)
{
    std::vector<double> q;
    for (int kl = 0; kl < n_rel; ++kl)
        if (rc2[kl] > 0.0) q.push_back(rc2[kl]);
    std::sort(q.begin(), q.end());
    q.erase(std::unique(q.begin(), q.end()), q.end());
    edges.assign(1, 0.0);
    edges.insert(edges.end(), q.begin(), q.end());
    rank.assign(n_rel, 0);
    for (int kl = 0; kl < n_rel; ++kl)
        if (rc2[kl] > 0.0)
            rank[kl] = (int)(std::lower_bound(q.begin(), q.end(), rc2[kl]) - q.begin()) + 1;
}

int check_common(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const void *xyz, const void *type,
                 const void *box, int n_rel, const void *rel)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(n_frames >= 0 && n_atoms >= 0, "negative sizes");
    MD_REQUIRE(n_frames == 0 || n_atoms == 0 || (xyz && type && box), "NULL input array");
    MD_REQUIRE(n_rel >= 0 && (n_rel == 0 || rel), "bad relation table");
    MD_REQUIRE(n_frames < (1LL << 31) && n_atoms < (1LL << 31), "sizes exceed 2^31");
    return MDHIP_OK;
}

}  // namespace

extern "C" {

int mdhip_rdf_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                     int on_device, const int32_t *type, int64_t type_frame_stride,
                     const double *box, int n_rel, const int32_t *rel, double r_cut_sq,
                     double bin_size, int nbins, const double *edges, int per_frame,
                     uint64_t *hist_full, uint64_t *hist_part, uint64_t *overflow)
{
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(nbins >= 1 && bin_size > 0.0, "nbins and bin_size must be positive");
    MD_REQUIRE(type_frame_stride == 0 || type_frame_stride == n_atoms,
               "type_frame_stride must be 0 or n_atoms");
    MD_REQUIRE(hist_full && (n_rel == 0 || hist_part), "NULL output");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    std::fill(hist_full, hist_full + out_frames * nbins, (uint64_t)0);
    if (n_rel) std::fill(hist_part, hist_part + out_frames * n_rel * nbins, (uint64_t)0);
    if (overflow) *overflow = 0;
    if (n_frames == 0 || n_atoms < 2) return MDHIP_OK;

    std::vector<double> own_edges;
    if (!edges) {
        own_edges.resize(nbins + 1);
        mdhip_bin_edges(bin_size, nbins, own_edges.data());
        edges = own_edges.data();
    }
    RelJob j{};
    j.tri = true;
    j.F = n_frames;
    j.ni = j.nj = n_atoms;
    j.xi = xyz;
    j.xi_dev = on_device;
    j.lab_i = type;
    j.lab_i_fs = type_frame_stride;
    j.box = box;
    j.n_rel = n_rel;
    j.rel = rel;
    j.nbins = nbins;
    j.edges = edges;
    j.rc2 = r_cut_sq;
    j.gscale = (float)(1.0 / bin_size);
    j.per_frame = per_frame;
    std::vector<uint64_t> H;
    std::vector<int> rel_cls;
    int n_cls = 0;
    uint64_t ov = 0;
    rc = run_job(ctx, j, H, rel_cls, n_cls, &ov);
    if (rc) return rc;
    if (overflow) *overflow = ov;
    // rdf_full[bin] += 2 per pair (rdf_cn.py:85-86); rdf_part: +1 per unordered {a,b} pair, +2 when a == b
    for (size_t f = 0; f < out_frames; ++f) {
        const uint64_t *Hf = &H[f * n_cls * nbins];
        uint64_t *full = hist_full + f * nbins;
        for (int c = 0; c < n_cls; ++c)
            for (int b = 0; b < nbins; ++b) full[b] += 2 * Hf[(size_t)c * nbins + b];
        for (int kl = 0; kl < n_rel; ++kl) {
            if (rel_cls[kl] < 0) continue;
            const uint64_t mult = rel[2 * kl] == rel[2 * kl + 1] ? 2 : 1;
            uint64_t *part = hist_part + (f * n_rel + kl) * nbins;
            const uint64_t *row = Hf + (size_t)rel_cls[kl] * nbins;
            for (int b = 0; b < nbins; ++b) part[b] = mult * row[b];
        }
    }
    return MDHIP_OK;
}

int mdhip_cn_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                    int on_device, const int32_t *type, int64_t type_frame_stride,
                    const double *box, int n_rel, const int32_t *rel, const double *r_cut_sq,
                    int per_frame, uint64_t *cn)
{
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(n_rel == 0 || (r_cut_sq && cn), "NULL cutoff or output");
    MD_REQUIRE(type_frame_stride == 0 || type_frame_stride == n_atoms,
               "type_frame_stride must be 0 or n_atoms");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    std::fill(cn, cn + out_frames * n_rel, (uint64_t)0);
    if (n_frames == 0 || n_atoms < 2 || n_rel == 0) return MDHIP_OK;
    std::vector<double> edges;
    std::vector<int> rank;
    cn_edges(n_rel, r_cut_sq, edges, rank);
    const int nbins = (int)edges.size() - 1;
    if (nbins == 0) return MDHIP_OK;
    RelJob j{};
    j.tri = true;
    j.F = n_frames;
    j.ni = j.nj = n_atoms;
    j.xi = xyz;
    j.xi_dev = on_device;
    j.lab_i = type;
    j.lab_i_fs = type_frame_stride;
    j.box = box;
    j.n_rel = n_rel;
    j.rel = rel;
    j.nbins = nbins;
    j.edges = edges.data();
    j.rc2 = edges.back();
    j.gscale = 0.f;
    j.per_frame = per_frame;
    std::vector<uint64_t> H;
    std::vector<int> rel_cls;
    int n_cls = 0;
    uint64_t ov = 0;
    rc = run_job(ctx, j, H, rel_cls, n_cls, &ov);
    if (rc) return rc;
    for (size_t f = 0; f < out_frames; ++f)
        for (int kl = 0; kl < n_rel; ++kl) {
            if (rel_cls[kl] < 0) continue;
            const uint64_t mult = rel[2 * kl] == rel[2 * kl + 1] ? 2 : 1;
            const uint64_t *row = &H[(f * n_cls + rel_cls[kl]) * nbins];
            uint64_t s = 0;
            for (int b = 0; b < rank[kl]; ++b) s += row[b];
            cn[f * n_rel + kl] = mult * s;
        }
    return MDHIP_OK;
}

int mdhip_rdf_sites(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                    int xyz_on_device, const int32_t *type, int64_t n_sites, const double *sites,
                    int sites_on_device, const int32_t *site_type, const double *box, int n_rel,
                    const int32_t *rel, double r_cut_sq, double bin_size, int nbins,
                    const double *edges, int per_frame, uint64_t *hist_part, uint64_t *overflow)
{
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(n_sites >= 0 && n_sites < (1LL << 31), "bad n_sites");
    MD_REQUIRE(n_frames == 0 || n_sites == 0 || (sites && site_type), "NULL site arrays");
    MD_REQUIRE(nbins >= 1 && bin_size > 0.0, "nbins and bin_size must be positive");
    MD_REQUIRE(n_rel == 0 || hist_part, "NULL output");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    if (n_rel) std::fill(hist_part, hist_part + out_frames * n_rel * nbins, (uint64_t)0);
    if (overflow) *overflow = 0;
    if (n_frames == 0 || n_atoms == 0 || n_sites == 0 || n_rel == 0) return MDHIP_OK;
    std::vector<double> own_edges;
    if (!edges) {
        own_edges.resize(nbins + 1);
        mdhip_bin_edges(bin_size, nbins, own_edges.data());
        edges = own_edges.data();
    }
    RelJob j{};
    j.tri = false;
    j.F = n_frames;
    j.ni = n_atoms;
    j.nj = n_sites;
    j.xi = xyz;
    j.xi_dev = xyz_on_device;
    j.xj = sites;
    j.xj_dev = sites_on_device;
    j.lab_i = type;
    j.lab_i_fs = 0;
    j.lab_j = site_type;
    j.box = box;
    j.n_rel = n_rel;
    j.rel = rel;
    j.nbins = nbins;
    j.edges = edges;
    j.rc2 = r_cut_sq;
    j.gscale = (float)(1.0 / bin_size);
    j.per_frame = per_frame;
    std::vector<uint64_t> H;
    std::vector<int> rel_cls;
    int n_cls = 0;
    uint64_t ov = 0;
    rc = run_job(ctx, j, H, rel_cls, n_cls, &ov);
    if (rc) return rc;
    if (overflow) *overflow = ov;
    for (size_t f = 0; f < out_frames; ++f)
        for (int kl = 0; kl < n_rel; ++kl) {
            if (rel_cls[kl] < 0) continue;
            memcpy(hist_part + (f * n_rel + kl) * nbins, &H[(f * n_cls + rel_cls[kl]) * nbins],
                   (size_t)nbins * 8);
        }
    return MDHIP_OK;
}

int mdhip_cn_sites(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                   int xyz_on_device, const int32_t *type, int64_t n_sites, const double *sites,
                   int sites_on_device, const int32_t *site_type, const double *box, int n_rel,
                   const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn)
{
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(n_sites >= 0 && n_sites < (1LL << 31), "bad n_sites");
    MD_REQUIRE(n_frames == 0 || n_sites == 0 || (sites && site_type), "NULL site arrays");
    MD_REQUIRE(n_rel == 0 || (r_cut_sq && cn), "NULL cutoff or output");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    std::fill(cn, cn + out_frames * n_rel, (uint64_t)0);
    if (n_frames == 0 || n_atoms == 0 || n_sites == 0 || n_rel == 0) return MDHIP_OK;
    std::vector<double> edges;
    std::vector<int> rank;
    cn_edges(n_rel, r_cut_sq, edges, rank);
    const int nbins = (int)edges.size() - 1;
    if (nbins == 0) return MDHIP_OK;
    RelJob j{};
    j.tri = false;
    j.F = n_frames;
    j.ni = n_atoms;
    j.nj = n_sites;
    j.xi = xyz;
    j.xi_dev = xyz_on_device;
    j.xj = sites;
    j.xj_dev = sites_on_device;
    j.lab_i = type;
    j.lab_i_fs = 0;
    j.lab_j = site_type;
    j.box = box;
    j.n_rel = n_rel;
    j.rel = rel;
    j.nbins = nbins;
    j.edges = edges.data();
    j.rc2 = edges.back();
    j.gscale = 0.f;
    j.per_frame = per_frame;
    std::vector<uint64_t> H;
    std::vector<int> rel_cls;
    int n_cls = 0;
    uint64_t ov = 0;
    rc = run_job(ctx, j, H, rel_cls, n_cls, &ov);
    if (rc) return rc;
    for (size_t f = 0; f < out_frames; ++f)
        for (int kl = 0; kl < n_rel; ++kl) {
            if (rel_cls[kl] < 0) continue;
            const uint64_t *row = &H[(f * n_cls + rel_cls[kl]) * nbins];
            uint64_t s = 0;
            for (int b = 0; b < rank[kl]; ++b) s += row[b];
            cn[f * n_rel + kl] = s;
        }
    return MDHIP_OK;
}

}  // extern "C"
