// pair_dense.hip — the LDS-tile pair kernels: the edge-table kernel of the first commit (rdf_variant 0, A/B
// baseline and fallback for more than 64 CN cutoffs) and the fast kernel (dense half-shell sweep for frames with
// fewer than 8 tiles, the un-culled atom x site loops, and the LDS-tile variant of the culled sweep).
// Formulation, exactness argument and binning: pair_hist.hip.
//
// Work decomposition: a block owns one tile of 256 "i" atoms of one frame (one atom per lane, in registers)
// and sweeps a list of 256-atom "j" tiles staged through LDS (double-buffered, one barrier per tile). For the
// triangular (atom-atom) case the j list is the half shell J = I, I+1, ..., I+nT/2 (mod nT), which covers every
// unordered tile pair once with equal work per block; only the J == I tile needs the i<j mask. Class histograms
// are LDS-private per block (ds_add_u32) and flushed once with 64-bit global atomics into one of `slots` replicas.
#include "pair_common.h"

#pragma clang fp contract(off)

namespace mdpair {
namespace {

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

__global__ void reduce_slots_kernel(const unsigned long long *__restrict__ in,
                                    unsigned long long *__restrict__ out, int words, int slots)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= words) return;
    unsigned long long s = 0;
    for (int r = 0; r < slots; ++r) s += in[(size_t)r * words + k];
    out[k] = s;
}

}  // namespace

size_t lds_bytes_fast(int nbins, int n_cls, int n_ti, int n_tj)
{
    size_t off = ((size_t)(n_cls + 1) * (nbins + 1) * 4 + 15) & ~size_t(15);
    off += sizeof(double4) * 2 * TILE;
    off += sizeof(float4) * 4 * (TILE / 8);  // group boxes (culled path)
    off += nbins <= 64 ? (size_t)(nbins + 2) * 8 : 0;  // CN mode keeps its edge table in LDS
    off += (size_t)n_ti * n_tj * 4;
    return (off + 15) & ~size_t(15);
}

size_t lds_bytes(int nbins, int n_cls, int n_ti, int n_tj)
{
    size_t off = (((size_t)(nbins + 2) * 8) + 15) & ~size_t(15);
    off += sizeof(JAtom) * 2 * TILE;
    off += (size_t)n_cls * nbins * 4;
    off += 16;
    off += (size_t)n_ti * n_tj;
    return (off + 15) & ~size_t(15);
}

PairKernel dense_kernel(bool fast, bool tri, bool mode_cn, bool list, const char **name)
{
#define MD_PICK(...) (*name = #__VA_ARGS__, __VA_ARGS__)
    if (!fast) return tri ? MD_PICK(pair_hist_kernel<true>) : MD_PICK(pair_hist_kernel<false>);
    if (list)
        return mode_cn ? MD_PICK(pair_hist_fast_kernel<true, 8, 1, true>) : MD_PICK(pair_hist_fast_kernel<true, 8, 0, true>);
    if (mode_cn)
        return tri ? MD_PICK(pair_hist_fast_kernel<true, 8, 1, false>) : MD_PICK(pair_hist_fast_kernel<false, 8, 1, false>);
    return tri ? MD_PICK(pair_hist_fast_kernel<true, 8, 0, false>) : MD_PICK(pair_hist_fast_kernel<false, 8, 0, false>);
#undef MD_PICK
}

void launch_reduce_slots(hipStream_t stream, const unsigned long long *in, unsigned long long *out, int words, int slots)
{
    hipLaunchKernelGGL(reduce_slots_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, in, out, words,
                       slots);
}

}  // namespace mdpair
