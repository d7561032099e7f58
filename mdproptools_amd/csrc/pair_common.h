// pair_common.h — types, constants and small device helpers shared by the pair-histogram translation units
// (pair_dense.hip: LDS-tile kernels; pair_cull.hip: spatial sort, boxes, tile lists; pair_sj.hip: scalar-j
// kernel; pair_hist.hip: host orchestration and the C-ABI). See pair_hist.hip for the formulation.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "ctx.h"

namespace mdpair {

constexpr int TILE = 256;
constexpr int SJ_GROUP = 4;  // j atoms per culling group of the scalar-j kernel (64 groups per tile: one per lane)
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
    // ordered-pair rows (MODE 2-4 of the scalar-j kernel): row of (ti, tj) = A[ti] + B[tj] with host-chosen integers
    // (pair_hist.hip, row displacement); the records carry A[t] * n_ti in the low word of w and near + B[t] * row_len
    // in the high word. The plain layout is A[t] = t * n_tj, B[t] = t (row_mul = n_tj with A[t] = t in the record).
    int row_mul;     // lane row base = (low word / n_ti) * row_mul rows
    int n_rows_ord;  // rows of the ordered layout: max A + max B + 1
    int n_frames, nTi, nTj, jsplit, blocks_per_frame;
    int per_frame, slots;
    int fpb;  // frames swept per block before the flush (fast kernel, frame-summed output only)
    // culled path: per (frame, i-tile) the j-tiles (>= I) whose bounding boxes come within the cutoff
    const unsigned short *list;  // [F][nTi][nTi]
    const int *list_cnt;         // [F][nTi]
    const float4 *gsph;          // [F][nTi*32][2] bounding box (lo, hi; w = 1 if non-empty) of every 8 sorted atoms
    const float4 *gsph4;         // [F][nTj*64][2] the same for every 4 sorted atoms of the j set (scalar-j kernel)
    const float4 *wsph;          // [F][nTi*4][2]  bounding box of every 64 sorted atoms (one wave's i atoms)
                                 // (packed-f32 sweep: both as (centre, half extents), see cull_boxes_kernel)
    float reach;                 // r_cut rounded up, plus slack for the f32 box test
    const double4 *aos;          // [F][nTi*256] sorted atoms (x, y, z, bits = type * n_ti), padded with +1e300
    unsigned *work;              // work counters of the scalar-j kernel, zeroed per launch: [8] per XCD
                                 // (frame-summed output) or [F] per frame (per-frame output)
    float near;                  // MODE 2: guard band half-width (also folded into the records' row offsets)
    unsigned *slices;            // scalar-j kernels: [blocks][LDS histogram words], every block stores its own copy
    const double4 *aos_j;        // sorted records of the j set (== aos for atom-atom), [F][nTj*256]
    int tri;                     // 1: atom-atom (i < j inside the diagonal tile), 0: atoms x sites
    // packed-f32 classification sweep (MODE 3 of the scalar-j kernel, pair_sj.hip)
    const float *rel;            // [F][nTj*128][8] f32 records of the j set relative to the centre of their block (64
                                 // or 256 sorted atoms), two atoms per record: (x0, x1, y0, y1, z0, z1, w0, w1),
                                 // w = the bin-guess addend of pack_w
    const double *cen;           // [F][nTj][8] tile centre (x, y, z) and half extents (hx, hy, hz)
    float s_cap;                 // largest |relative coordinate| sum (i + j, per axis) the error bound `near` covers
    float cut_lo;                // MODE 4 (cutoff inside a bin): sqrt(rsq32) >= cut_lo may lie beyond the cutoff
    float rc2hi;                 // f32 pre-filter: every in-cutoff pair has rsq32 < rc2hi (see pk_error_bound)
    // coordination numbers from the RDF sweep (pair_hist_sj_kernel<., ., true>)
    int n_cn;                    // != 0: on
    const unsigned *cn_tab;      // word index of every row's split bin (-1: none; padded to 8 bytes) | cutoff^2 per row
    float cn_reach;              // groups whose box is farther than this from the wave's box hold no split-bin pair    // overflow guard of the per-block 32-bit histogram words (scalar-j kernels)
    unsigned guard_off;          // LDS byte offset of the block's tile counter (behind everything else)
    unsigned guard_tiles;        // a block that swept more neighbour tiles than this may have wrapped a word
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

struct FastCtx {
    unsigned *hist;               // LDS offset 0
    const double *edges;          // global memory, nbins+2 entries, last = +inf
    const unsigned *rowtab_me;    // LDS: &rowtab[0][ti] of the table [n_tj][n_ti] -> LDS byte address of the class row
    float gscale, near, near2;    // guard band half-width and its double
    unsigned rowbase_me;          // MODE 2: LDS byte address of row (ti, 0) of the ordered-pair histogram
    unsigned lds_base;            // LDS byte address of the histogram
    int nbins;
};

// 4th double of a sorted record: low word = type * n_ti (word offset into the [tj][ti] row table), high word =
// float(near + type * row_len), the addend of the bin guess that carries the row of the ordered-pair layout
// (MODE 2 of the scalar-j kernel; 0 otherwise). With displaced rows (`disp` != nullptr: [2][n_types] = A | B, see
// PairArgs::row_mul) the low word carries A[t] * n_ti and the addend B[t] * row_len.
struct RowDisp {
    const int *lo = nullptr;  // device: A[t] (i role) — nullptr: t
    const int *hi = nullptr;  // device: B[t] (j role) — nullptr: t
};
__device__ __forceinline__ double pack_w(int t, int n_ti, float near, int row_len, const RowDisp &d)
{
    const int tl = d.lo ? d.lo[t] : t, th = d.hi ? d.hi[t] : t;
    const unsigned lo = (unsigned)(tl * n_ti);
    const unsigned hi = row_len > 0 ? __float_as_uint(near + (float)(th * row_len)) : 0u;
    return __hiloint2double((int)hi, (int)lo);
}

// Spatial sort + bounding boxes of ONE atom set of a batch of frames (culled path): Hilbert-sorted records
// `aos` [F][nT*256], tile boxes `bbox` [F][nT][6], 8-atom and 64-atom boxes `gs` / `ws`. `slot` = workspace ids of
// {records, tile boxes, group boxes, wave boxes}; keys, cell counters and the SoA copy are shared scratch.
struct SortedSet {
    const float *rel = nullptr;   // packed-f32 records (tile-relative), see PairArgs::rel
    const double *cen = nullptr;  // tile centres and half extents
    const double4 *aos = nullptr;
    const double *bbox = nullptr;
    const float4 *gs = nullptr, *ws = nullptr, *gs4 = nullptr;
    const double *sx = nullptr;
    const int *st = nullptr;
};

// ---- host entry points of the kernel translation units ----
typedef void (*PairKernel)(const PairArgs);

// pair_dense.hip
size_t lds_bytes(int nbins, int n_cls, int n_ti, int n_tj);       // edge-table kernel (rdf_variant 0)
size_t lds_bytes_fast(int nbins, int n_cls, int n_ti, int n_tj);  // LDS-tile fast kernel
PairKernel dense_kernel(bool fast, bool tri, bool mode_cn, bool list, const char **name);
void launch_reduce_slots(hipStream_t stream, const unsigned long long *in, unsigned long long *out, int words, int slots);

// pair_sj.hip
size_t lds_bytes_sj(int nbins, int n_cls, int n_ti, int n_tj, bool mode_cn);
size_t lds_bytes_sj_ordered(int nbins, int n_rows);
size_t lds_bytes_sj_pk(int nbins, int n_rows, int n_cn = 0, bool big = false);  // big: 16 waves' queues
size_t lds_bytes_sj_pk_rows(int nbins, int n_cls, int n_ti, int n_tj, int n_cn = 0);  // class rows + row table + queues
int sj_block_threads(int mode, bool big = false);  // threads per block of the scalar-j kernels (mode as sj_kernel)
PairKernel sj_kernel(int mode /* 0 RDF class rows, 1 CN, 2 RDF ordered-pair rows, 3 = 2 with the packed-f32 sweep,
                                 4 = 3 with the cutoff guard (cutoff inside a bin), 5 / 6 = 3 / 4 with class rows */,
                     bool persist, bool cn, bool big /* modes 3, 4 without cn: one 16-wave block per CU */, const char **name);
// error bound (in bins) of the packed-f32 bin guess for |relative coordinates| <= s_cap per axis pair sum; 0 = not usable
double pk_error_bound(double r_cut, double bin_size, int nbins, int n_tj, double s_cap, double l_max);
void launch_derive_rdf(hipStream_t stream, const unsigned long long *rows, int n_rows, int nbins, const int *rowcls,
                       int n_rel, const int *relcls, const int *relmult, const unsigned long long *guard,
                       unsigned long long *out);
void launch_merge_slices(hipStream_t stream, const unsigned *slices, int hist_words, long long n_blocks, int per_frame,
                         int bpf, unsigned grid_y, unsigned long long *rows);

// pair_cull.hip
constexpr int MORTON_BITS = 5;                       // 32 cells per axis
constexpr int MORTON_CELLS = 1 << (3 * MORTON_BITS); // 32768
int cull_prepare_set(mdhip_ctx *ctx, int64_t F, const double *d_x, const int *d_t, long long t_fs,
                     const double *d_box, long long N, int nT, int n_ti, float near, int row_len, RowDisp disp,
                     bool want_soa, int want_rel /* != 0: the tile-relative f32 records of the packed sweep and their tiles' centres */,
                     int rel_w_type /* w of the f32 records: 0 bin-guess addend, 1 row-table offset */,
                     int cbox /* 1: 4-atom and 64-atom boxes as (centre, half extents): packed-f32 sweep */,
                     const int slot[5], SortedSet &out);
void launch_cull_lists(hipStream_t stream, bool tri, int64_t F, const double *bbox_i, const double *bbox_j, int nTi,
                       int nTj, const double *d_box, double rc2_test, unsigned short *list, int *cnt);

}  // namespace mdpair
