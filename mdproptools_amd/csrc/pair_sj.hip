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
// Returns the number of neighbour tiles of the slice (the overflow guard of the 32-bit LDS words counts them).
template <int MODE>
__device__ __forceinline__ int sj_item(const PairArgs &a, FastCtx &c, const unsigned *s_row, int f, int I, int wq,
                                       int split, int lane)
{
    const long long n_pad = (long long)a.nTi * TILE, n_pad_j = (long long)a.nTj * TILE;
    const long long rowid = (long long)f * a.nTi + I;
    const int cnt = a.list_cnt[rowid];
    const unsigned short *row_list = a.list + rowid * a.nTj;  // (nTj == nTi for atom-atom)
    const int t_begin = (int)((long long)split * cnt / a.jsplit);
    const int t_end = (int)((long long)(split + 1) * cnt / a.jsplit);
    if (t_begin >= t_end) return 0;
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
        c.rowbase_me = c.lds_base + (unsigned)ti_me * (unsigned)a.row_mul * (unsigned)(a.nbins + 1) * 4u;
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
    return t_end - t_begin;
}

// ------------------------------------------------------------------------------------------------
// MODE 3: packed-f32 classification with an exact deferred resolver (ordered-pair rows, as MODE 2).
//
// Only the BIN of a pair enters the result, not its rsq. The sweep therefore evaluates rsq in f32 with the
// packed instructions of gfx950 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two pairs per VALU issue slot, the
// j operands as SGPR pairs): 3 slots per pair instead of 8-15 f64 slots. The f32 value differs
// from the reference's f64 rsq by a BOUNDED amount (pk_error_bound below, in bins: `err`); the bin guess
//   g = fma(v_sqrt_f32(rsq32), 1/ddr, near + tj*row_len),   near = 2*err + slack
// is trusted only when fract(g) >= 2*near, i.e. when the true sqrt(rsq)/ddr is provably farther than `err` from
// every integer — then trunc(g) IS the reference's bin. Every other pair (~0.2 %, the band around each bin edge;
// the cutoff sits on such an edge by the host's choice of when to use this mode) is not binned here: its (i, j)
// indices go to a per-wave queue in LDS, and the queue is drained 64 entries at a time by the EXACT chain of
// MODE 2 (f64 coordinates re-read from the sorted records, the reference's operations in the reference's
// order, exact cutoff test, exact edge table). So the integers are the reference's; only who computes which
// pair changes. Lane-parallel draining makes the exact chain cost ~1/64 of what an inline fallback would.
//
// Keeping the f32 error small, and the wrap out of the pair loop: the j atoms are stored relative to the centre c
// of the box of their tile of 256 sorted atoms (64-atom blocks were slower: retired in round 4), and every lane moves
// its own i atom to the periodic image nearest to that centre, once per neighbour tile:
//   xr_j = f32(x_j - c)                          (pre-pass, |xr_j| <= half extent h of the block)
//   q    = x_i - c;  xr_i = f32(q - L rint(q/L)) (f64 per lane and block, |xr_i| <= L/2)
// so that d' = xr_i - xr_j is d = x_i - x_j moved by a whole number of box lengths, |d'| <= L/2 + e (e = how far
// the group's atoms reach from c). With |d| < 1.5 L the reference's single wrap yields the nearest image of d, and
//   |d'| <= L/2            : d' IS that nearest image;
//   L/2 < |d'| <= L/2 + e  : the nearest image is L - |d'| away.
// So wherever |d'| <= L - r_cut - margin the plain difference is enough: either d' is the nearest image, or both d'
// and the nearest image exceed the cutoff on that axis alone and the pair counts nowhere either way. The wave tests
// this per (wave box, group box) and axis, lane-parallel over the groups of a tile: when all its lanes sit at the
// same image n (the wave box does not straddle a wrap boundary of this centre), d' ranges over
// [wlo' - ghi', whi' - glo'] (primed = relative to c, the wave box moved by n L). Only groups where that interval
// reaches beyond L - r_cut (cutoffs close to L/2: the far corner of a neighbour at the edge of the cutoff) take the
// per-pair f32 wrap d' - L rint(d'/L) on that axis (VAR bit). Blocks that the bound does not cover — some
// |x_i - x_j| >= 1.5 L (atoms box lengths outside the cell, where the reference's single wrap is not the nearest
// image) or |xr_i| + h beyond s_cap — are swept by the exact f64 chain (sweep_group_sj<., 2, 7>: general wrap,
// valid for every d).
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PK_QCAP = 320;     // queue entries per wave: drained above 64, one group of 4 j atoms adds at most 256
constexpr int PK_QSTRIDE = 320;  // words per wave
constexpr int PK_THREADS = 512;  // threads per block of the MODE 3 kernel (8 independent waves, one LDS histogram)
constexpr int PK_WAVES_PER_SIMD = 6;  // HIP's second launch bound: the register budget (<= 80 VGPRs) for 3 such blocks per CU
// BIG (round 6): ONE block of 16 waves per CU sharing one histogram of up to ~140 KB — ordered rows that fit neither a third of
// LDS nor a displaced layout (every pair of nine types named: 81 rows x 401 words = 130 KB) stay on the table-free sweep,
// at 4 waves per SIMD (128 registers: no spills) instead of 6, rather than going through the class rows in two passes
constexpr int PK_BIG_THREADS = 1024;

#ifndef PK_INLINE_PUSH
#define PK_INLINE_PUSH 0  // 1 = queue pushes inside the hand-written pair block (bin_pair2q, A/B builds). Measured build
                          // against build (tools/ab_libs.py, integers identical): push in line behind a taken branch
                          // C2 +1.1 %, C3 +2.3 %; push out of line (no taken branch on the common path, 0.5 scalar
                          // issues fewer per slot, no spills) C2 +0.5 %, C3 +1.5 %, CN -1 % — fewer instructions, not
                          // less time: the masks-out form stays
#endif
struct PkCtx {
    f32x2 x2, y2, z2;     // this lane's i atom (its image nearest to the j block's centre) relative to that centre,
                          // both halves equal
    f32x2 Lx2, Ly2, Lz2;  // box lengths (f32) for the axes that still need the per-pair wrap
    f32x2 iLx2, iLy2, iLz2;  // and their reciprocals
    float rc2hi;          // pre-filter: every pair with rsq < r_cut^2 has rsq32 < rc2hi
    float cut_lo;         // CUTG: sqrt(rsq32) >= cut_lo may be beyond the cutoff (pk_cut_lo)
    unsigned long long full;  // the exec mask of the sweep (the pair block narrows exec and puts this back)
    const unsigned *rowtab;  // ROWS: the class-row table [n_tj][n_ti] of LDS byte addresses (nullptr: ordered rows)
    unsigned *queue;      // this wave's queue (LDS)
    unsigned qaddr;       // its LDS byte address (wave-uniform)
    int qn;               // entries queued (wave-uniform: kept in an SGPR)
    unsigned long long *lost;  // device counter of entries that did not fit the queue (must stay 0)
    // what the resolver needs
    const double4 *ats_i;  // sorted f64 records of the i atoms of this wave (64 consecutive)
    const double4 *ats_j;  // sorted f64 records of the frame's j set
    double Lx, Ly, Lz, rc2;
    int n_ti, row_mul;
    // CN in the same sweep (CNG). The histogram bins are exact, so a pair in a bin below the one that holds its
    // class's coordination cutoff is inside that cutoff and a pair in a bin above it is not: only the pairs of that one
    // SPLIT bin need the exact comparison rsq < cutoff^2. Their histogram words are flagged (one bit per word); a
    // pair that the sweep has binned into a flagged word is queued as a "CN only" entry, and the exact chain counts it
    // in its row's split counter when it is inside the cutoff. The host adds the bins below the split bin.
    const unsigned *cn_kc;     // LDS: per histogram row the LDS byte address of its split-bin word (~0u: none), [rows]
    const unsigned *cn_kc_me;  // this lane's part of it: ordered rows &cn_kc[ti * n_tj] (index tj); class rows: the table
                               // [n_tj][n_ti] of split-word addresses beside the row table, &tab[ti] (index = rowtab's)
    const double *cn_c2;       // LDS: cutoff^2 of the row's class (0: none), [rows]
    unsigned cn_base;          // LDS byte address of the split counters [rows]
    float inv_row_len;         // 1 / (nbins + 1)
};
constexpr unsigned PK_CN_ONLY = 0x80000000u;  // queue entry: already binned by the sweep, count the split bin only

// d - L rint(d / L) on both halves: the image of d nearest to zero (|d| < 1.5 L). The rint comes out of the adder:
// d / L + 1.5 * 2^23 has an ulp of one, so the fused multiply-add rounds the exact quotient to an integer (one rounding;
// |d / L| < 2^22) and subtracting the constant again is exact — three packed instructions per two pairs, where
// v_rndne_f32 (not packed) made it four. One more rounding in the last fma; the rint may fall either way within rounding
// of |d| = L/2, where both images have the same magnitude.
__device__ __forceinline__ f32x2 wrap_pk(f32x2 d, f32x2 L, f32x2 iL)
{
    constexpr float M = 12582912.0f;  // 1.5 * 2^23
    f32x2 n = __builtin_elementwise_fma(d, iL, f32x2{M, M});
    asm volatile("" : "+v"(n));  // (keeps the two steps apart: (x + M) - M is not x)
    n = n - f32x2{M, M};
    return __builtin_elementwise_fma(-n, L, d);
}

// The exact chain for the n queued pairs: entry = (j index in the frame << 6) | lane of the i atom.
// CNG: a resolved pair that lands in a flagged (split) word is counted in its row's split counter when
// rsq < cutoff^2 of the row's class (rdf_cn.py:100-119: strict); PK_CN_ONLY entries were binned by the sweep already.
template <bool CNG>
__device__ __forceinline__ void pk_drain(PkCtx &p, const FastCtx &c, int lane)
{
    // The pushes carry no capacity test (pk_push): a wave drains above 64 entries before every group and a group adds at
    // most 4 x 64, so qn <= PK_QCAP here — by construction. A check even on this rare path is not free (round 4, build
    // against build: +1.1 % at C2, +2.2 % at C3 — the function is inlined at every drain point and the extra live
    // values move the whole kernel's allocation), so it is a DEBUG build's: -DPK_CAPCHECK counts entries beyond the
    // queue in the `lost` word, which the host turns into an error. The invariant itself is exercised at its limit by
    // tests/test_gpu_hardening.py::test_every_pair_ambiguous_fills_the_queues (every pair of every group ambiguous:
    // 64 + 256 entries per wave, against the oracle).
#ifdef PK_CAPCHECK
    if (p.qn > PK_QCAP) {
        if (lane == 0) atomicAdd(p.lost, (unsigned long long)(p.qn - PK_QCAP));
        p.qn = PK_QCAP;
    }
#endif
    const int n = p.qn < PK_QCAP ? p.qn : PK_QCAP;
    for (int b = 0; b < n; b += 64) {
        if (b + lane < n) {
            const unsigned e = p.queue[b + lane];
            const double4 ri = p.ats_i[e & 63u];
            const double4 rj = p.ats_j[(e & ~PK_CN_ONLY) >> 6];
            const double ax = wrap_abs(ri.x - rj.x, p.Lx);
            const double ay = wrap_abs(ri.y - rj.y, p.Ly);
            const double az = wrap_abs(ri.z - rj.z, p.Lz);
            const double rsq = (ax * ax + ay * ay) + az * az;
            if (rsq < p.rc2) {
                const int ti = (int)((unsigned)__double2loint(ri.w)) / p.n_ti;  // low word of w = type * n_ti
                // ordered rows: row (ti, .) + the offset of tj in the addend; class rows: table[tj][ti], addend near
                const unsigned rowbase = p.rowtab ? p.rowtab[__double2loint(rj.w) + ti]
                                                  : c.lds_base + (unsigned)ti * (unsigned)p.row_mul * (unsigned)(c.nbins + 1) * 4u;
                const float nearoff = p.rowtab ? c.near : __int_as_float(__double2hiint(rj.w));
                const float g1 = __builtin_fmaf(__builtin_amdgcn_sqrtf((float)rsq), c.gscale, nearoff);
                int k = (int)g1;
                if (__builtin_amdgcn_fractf(g1) < c.near2) {
                    const int koff = (int)nearoff;
                    int kk = k - koff;
                    kk = kk > c.nbins ? c.nbins : (kk < 0 ? 0 : kk);
                    k = koff + (rsq < c.edges[kk] ? kk - 1 : kk);
                }
                const unsigned addr = ((unsigned)k << 2) + rowbase;
                if (!CNG || !(e & PK_CN_ONLY)) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
                if (CNG) {
                    // row = word / row_len without a division (word < 2^20, the product is within 1e-3 of the quotient:
                    // truncation after adding half a word is exact)
                    const unsigned w = (addr - c.lds_base) >> 2;
                    const unsigned row = (unsigned)(((float)w + 0.5f) * p.inv_row_len);
                    if (p.cn_kc[row] == addr && rsq < p.cn_c2[row]) {
                        const unsigned caddr = p.cn_base + row * 4u;
                        asm volatile("ds_add_u32 %0, %1" ::"v"(caddr), "v"(1u) : "memory");
                    }
                }
            }
        }
    }
    p.qn = 0;
}

// The f32 records of one group: four j atoms as two packed records (x0, x1, y0, y1 | z0, z1, w0, w1) each. Loaded
// with plain (compiler-visible) loads from a wave-uniform address, so that the compiler tracks their latency and
// the load of the NEXT group can be in flight while the current one is swept.
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct RelQ {
    f32x4 a, b, c, d;
};
__device__ __forceinline__ RelQ load_relq(const float *__restrict__ p)
{
    // constant address space: the records were written by an earlier launch and are only read here, and the address
    // is wave-uniform, so these become scalar loads (s_load_dwordx4/x8 into SGPRs through the scalar cache) — vector
    // loads of one address by 64 lanes would cost the texture path 16 cycles each
    typedef const __attribute__((address_space(4))) f32x4 *cptr;
    const cptr q = (cptr)(unsigned long long)p;
    RelQ r;
    r.a = q[0];
    r.b = q[1];
    r.c = q[2];
    r.d = q[3];
    return r;
}

// Two pair slots of the packed sweep (the two halves of one packed rsq), hand-scheduled. The kernel is bound by ISSUE
// slots, and a scalar instruction costs a SIMD as much issue time as a vector one (tools/ubench_salu.hip: 16 s_add_u32
// 31 ns, 16 v_pk_fma_f32 33 ns per trip at 6 waves; interleaved 1:1 36 ns, 2:1 62 ns), so the exec-mask bookkeeping is
// written with the compares that write exec themselves:
//     v_cmpx_gt  c = exec = lanes inside the cutoff pre-filter      (was v_cmp + s_mov amb,0 + s_and_saveexec)
//     s_cbranch_execz                                               (no lane inside: skip the bin guess)
//     sqrt, fma, fract, cvt, lshl_add                               (bin guess + LDS address)
//     v_cmpx_ge  d = exec = lanes whose guess is safe               (was v_cmp + s_andn2 amb + s_and exec)
//     ds_add_u32
//     s_andn2    amb = c & ~d  (c = 0 on the skipped path)          (the lanes inside the error band of an edge)
//     s_mov      exec = full
// = 8 VALU + 2 SALU + 1 branch per slot (round 1: 7 VALU + 5 SALU + 1 branch, and a compare + branch on amb per slot
// behind it; the caller now tests amb0 | amb1 once). `full` is the exec mask of the sweep (read once per item).
// INVARIANT (also pk_push): these blocks rewrite exec and put `full` back, and the compiler does not know — every call
// must sit in wave-uniform control flow with exactly the lanes of `full` live (sj_item_pk reads it at its top; every
// branch between there and the sweeps is on a wave-uniform value: ballots, scalar masks, readfirstlane'd indices).
// A call placed under a divergent branch would silently re-enable lanes. work_loop_sane() checks at the top of every
// work item that all 64 lanes are present, so `full` is ~0 for every sweep; reading exec inside the block instead
// would cost one more scalar instruction per two slots in a loop that is bound by issue slots, scalar ones included.
// v_sqrt_f32 needs one wait state before its result is read.
// CUTG: the cutoff does not sit on a bin edge, so the band of an edge does not decide in/out of the cutoff: lanes whose
// f32 distance reaches the cutoff's own error band (sqrt(rsq32) >= cut_lo) are ambiguous too: one more v_cmpx (t < cl).
// RET: also hand back the LDS address every lane computed and the mask of the lanes that added there (the CN check of
// the groups near the wave looks those words up in the flag bits: +1 SALU per slot).
// (-DPK_NO_EXECZ, an A/B build only, round 6: the slot without its skip — a slot none of whose lanes is inside the
// cutoff then issues its six bin-guess instructions with an empty exec mask instead of one taken branch. Measured
// against the shipped form in one process: profiles/r06_ab_execz.txt.)
#ifdef PK_NO_EXECZ
#define BP_SKIP(L) ""
#else
#define BP_SKIP(L) "s_cbranch_execz " L "f\n\t"
#endif
#define BP_SLOT(K, L)                                                \
    "v_cmpx_gt_f32_e64 %[c" K "], %[rc2], %[rsq" K "]\n\t"            \
    BP_SKIP(L)                                                       \
    "v_sqrt_f32 %[t" K "], %[rsq" K "]\n\t"                           \
    "s_nop 0\n\t"
#define BP_SLOT_CUT(K) "v_cmpx_lt_f32_e64 %[d" K "], %[t" K "], %[cl]\n\t"
#define BP_SLOT_MID(K)                                               \
    "v_fma_f32 %[t" K "], %[t" K "], %[gs], %[no" K "]\n\t"           \
    "v_fract_f32 %[fr], %[t" K "]\n\t"                               \
    "v_cvt_i32_f32 %[t" K "], %[t" K "]\n\t"                          \
    "v_lshl_add_u32 %[t" K "], %[t" K "], 2, %[rb" K "]\n\t"          \
    "v_cmpx_ge_f32_e64 %[d" K "], %[fr], %[n2]\n\t"                  \
    "ds_add_u32 %[t" K "], %[one]\n\t"
#define BP_SLOT_END(K, L)                                            \
    L ":\n\t"                                                       \
    "s_andn2_b64 %[c" K "], %[c" K "], %[d" K "]\n\t"                 \
    "s_mov_b64 exec, %[full]\n\t"
// RET: the lanes that added = c & d (c = 0 when the slot was skipped), before c becomes amb
#define BP_SLOT_DONE(K, L) L ":\n\t" "s_and_b64 %[dn" K "], %[c" K "], %[d" K "]\n\t"
struct PairOut {
    unsigned long long amb0, amb1;    // lanes whose pair lies inside the error band (to be queued for the exact chain)
    unsigned long long done0, done1;  // RET: lanes that added to the histogram
    unsigned addr0, addr1;            // RET: the LDS address every lane computed
};
// Returns whether any lane of either slot is ambiguous (one test for both slots; leaving the asm through an
// `asm goto` exit on the s_or's scc would save the compare, but callbr with outputs crashes this compiler's ISel).
template <bool CUTG, bool RET>
__device__ __forceinline__ bool bin_pair2(float rsq0, float rsq1, float rc2hi, float gscale, float nearoff0, float nearoff1,
                                          float near2, unsigned rowbase0, unsigned rowbase1, float cut_lo,
                                          unsigned long long full, PairOut &o)
{
    unsigned long long c0, c1, d0, d1;
    float t0, t1, fr;
#define BP_OUT [c0] "=&s"(c0), [c1] "=&s"(c1), [d0] "=&s"(d0), [d1] "=&s"(d1), [t0] "=&v"(t0), [t1] "=&v"(t1), [fr] "=&v"(fr)
#define BP_IN                                                                                                          \
    [rc2] "s"(rc2hi), [rsq0] "v"(rsq0), [rsq1] "v"(rsq1), [gs] "v"(gscale), [no0] "s"(nearoff0), [no1] "s"(nearoff1),   \
        [n2] "v"(near2), [rb0] "v"(rowbase0), [rb1] "v"(rowbase1), [one] "v"(1u), [full] "s"(full)
    if (RET) {
        unsigned long long dn0, dn1;
        if (CUTG)
            asm volatile(BP_SLOT("0", "1") BP_SLOT_CUT("0") BP_SLOT_MID("0") BP_SLOT_DONE("0", "1")
                             "s_andn2_b64 %[c0], %[c0], %[d0]\n\ts_mov_b64 exec, %[full]\n\t"
                         BP_SLOT("1", "2") BP_SLOT_CUT("1") BP_SLOT_MID("1") BP_SLOT_DONE("1", "2")
                             "s_andn2_b64 %[c1], %[c1], %[d1]\n\ts_mov_b64 exec, %[full]"
                         : BP_OUT, [dn0] "=&s"(dn0), [dn1] "=&s"(dn1)
                         : BP_IN, [cl] "v"(cut_lo)
                         : "vcc", "scc", "memory");
        else
            asm volatile(BP_SLOT("0", "1") BP_SLOT_MID("0") BP_SLOT_DONE("0", "1")
                             "s_andn2_b64 %[c0], %[c0], %[d0]\n\ts_mov_b64 exec, %[full]\n\t"
                         BP_SLOT("1", "2") BP_SLOT_MID("1") BP_SLOT_DONE("1", "2")
                             "s_andn2_b64 %[c1], %[c1], %[d1]\n\ts_mov_b64 exec, %[full]"
                         : BP_OUT, [dn0] "=&s"(dn0), [dn1] "=&s"(dn1)
                         : BP_IN
                         : "vcc", "scc", "memory");
        o.done0 = dn0;
        o.done1 = dn1;
        o.addr0 = __float_as_uint(t0);
        o.addr1 = __float_as_uint(t1);
        o.amb0 = c0;
        o.amb1 = c1;
        return (c0 | c1) != 0;
    }
    if (CUTG) {
        asm volatile(BP_SLOT("0", "1") BP_SLOT_CUT("0") BP_SLOT_MID("0") BP_SLOT_END("0", "1")
                     BP_SLOT("1", "2") BP_SLOT_CUT("1") BP_SLOT_MID("1") BP_SLOT_END("1", "2")
                     : BP_OUT
                     : BP_IN, [cl] "v"(cut_lo)
                     : "vcc", "scc", "memory");
    } else {
        asm volatile(BP_SLOT("0", "1") BP_SLOT_MID("0") BP_SLOT_END("0", "1")
                     BP_SLOT("1", "2") BP_SLOT_MID("1") BP_SLOT_END("1", "2")
                     : BP_OUT
                     : BP_IN
                     : "vcc", "scc", "memory");
    }
    o.amb0 = c0;
    o.amb1 = c1;
    return (c0 | c1) != 0;
#undef BP_OUT
#undef BP_IN
}
// bin_pair2 with the queue push of the ambiguous lanes INSIDE the block (round 3; the plain variant without the CN flag
// check): `s_andn2 amb = c & ~d` leaves SCC = (amb != 0), so the block branches on it where it stands instead of handing
// the masks out for an s_or + s_cmp + s_cbranch per two slots and a compare + branch per slot behind those; the rare
// path is pk_push's own seven instructions + the tag ((jbase + 4 g + KOFF + slot) << 6, from scalars that are live
// anyway) + the count. Per slot on the common path: execz branch, andn2, scc branch, exec restore = 4 scalar issues
// (4.5 before), and the C++ around the block has no control flow left for the compiler to structurize.
// The push sits OUT OF LINE (subsection 1 of the kernel's text section, behind the function): on the common path both
// branches of a slot fall through — a taken branch costs several issue cycles, and the first form of this block, with
// the push in line behind an `s_cbranch_scc0` that was taken on every unambiguous slot, measured 1-2 % SLOWER than
// the masks-out form it replaced. L = the slot's skip / return label, R = the label of its out-of-line push.
#define BQ_PUSH(K, L, R)                                               \
    "s_andn2_b64 %[c" K "], %[c" K "], %[d" K "]\n\t"                  \
    "s_cbranch_scc1 " R "f\n\t"                                        \
    L ":\n\t"                                                         \
    "s_mov_b64 exec, %[full]\n\t"                                      \
    ".subsection 1\n\t"                                                \
    R ":\n\t"                                                         \
    "s_mov_b64 exec, %[c" K "]\n\t"                                    \
    "v_mbcnt_lo_u32_b32 %[fr], exec_lo, 0\n\t"                         \
    "v_mbcnt_hi_u32_b32 %[fr], exec_hi, %[fr]\n\t"                     \
    "s_lshl2_add_u32 %[st], %[qn], %[qaddr]\n\t"                       \
    "v_lshl_add_u32 %[t" K "], %[fr], 2, %[st]\n\t"                    \
    "s_lshl2_add_u32 %[st], %[g], %[jbase]\n\t"                        \
    "s_add_i32 %[st], %[st], %[ko" K "]\n\t"                           \
    "s_lshl_b32 %[st], %[st], 6\n\t"                                   \
    "v_or_b32 %[fr], %[st], %[lane]\n\t"                               \
    "ds_write_b32 %[t" K "], %[fr]\n\t"                                \
    "s_bcnt1_i32_b64 %[st], exec\n\t"                                  \
    "s_add_i32 %[qn], %[qn], %[st]\n\t"                                \
    "s_branch " L "b\n\t"                                              \
    ".subsection 0\n\t"
template <bool CUTG, int KOFF>
__device__ __forceinline__ void bin_pair2q(float rsq0, float rsq1, float rc2hi, float gscale, float nearoff0, float nearoff1,
                                           float near2, unsigned rowbase0, unsigned rowbase1, float cut_lo,
                                           unsigned long long full, int &qn, unsigned qaddr, int jbase, int g, int lane)
{
    unsigned long long c0, c1, d0, d1;
    float t0, t1, fr;
    unsigned st;
    // (the count travels through the block as a tied scalar operand; the readfirstlane tells the compiler's
    // uniformity analysis what it cannot see through an asm with vector outputs — it folds to nothing)
    int qn_io = __builtin_amdgcn_readfirstlane(qn);
#define BQ_OUT                                                                                                         \
    [c0] "=&s"(c0), [c1] "=&s"(c1), [d0] "=&s"(d0), [d1] "=&s"(d1), [t0] "=&v"(t0), [t1] "=&v"(t1), [fr] "=&v"(fr),    \
        [st] "=&s"(st), [qn] "+s"(qn_io)
#define BQ_IN                                                                                                          \
    [rc2] "s"(rc2hi), [rsq0] "v"(rsq0), [rsq1] "v"(rsq1), [gs] "v"(gscale), [no0] "s"(nearoff0), [no1] "s"(nearoff1),   \
        [n2] "v"(near2), [rb0] "v"(rowbase0), [rb1] "v"(rowbase1), [one] "v"(1u), [full] "s"(full), [qaddr] "s"(qaddr), \
        [jbase] "s"(jbase), [g] "s"(g), [lane] "v"(lane), [ko0] "n"(KOFF), [ko1] "n"(KOFF + 1)
    // (a slot with no lane inside the cutoff jumps over its bin guess AND its push: labels 1 / 2)
    if (CUTG) {
        asm volatile(BP_SLOT("0", "1") BP_SLOT_CUT("0") BP_SLOT_MID("0") BQ_PUSH("0", "1", "3")
                     BP_SLOT("1", "2") BP_SLOT_CUT("1") BP_SLOT_MID("1") BQ_PUSH("1", "2", "4")
                     : BQ_OUT
                     : BQ_IN, [cl] "v"(cut_lo)
                     : "vcc", "scc", "memory");
    } else {
        asm volatile(BP_SLOT("0", "1") BP_SLOT_MID("0") BQ_PUSH("0", "1", "3")
                     BP_SLOT("1", "2") BP_SLOT_MID("1") BQ_PUSH("1", "2", "4")
                     : BQ_OUT
                     : BQ_IN
                     : "vcc", "scc", "memory");
    }
    qn = qn_io;
#undef BQ_OUT
#undef BQ_IN
}
#undef BQ_PUSH
#undef BP_SLOT
#undef BP_SLOT_CUT
#undef BP_SLOT_MID
#undef BP_SLOT_END
#undef BP_SLOT_DONE

// jbase + 4 g of a group, computed where it is needed (the rare queue pushes): the opaque copy keeps the compiler from
// hoisting the two scalar instructions into every group's sweep (the kernel is bound by issue slots, scalar ones too)
struct GroupIdx {
    int jbase, g;
};
__device__ __forceinline__ int jidx_here(GroupIdx k)
{
    int g = k.g;
    asm volatile("" : "+s"(g));
    return k.jbase + g * SJ_GROUP;
}

// Appends one queue entry per lane of `mask` (wave-uniform, non-zero): (tag | lane) at queue[qn + rank of the lane in
// the mask]. The exec mask itself selects the lanes — v_mbcnt ranks them, one LDS store — and is put back to the sweep's
// mask (p.full: every call sits in wave-uniform control flow with the sweep's lanes live, as bin_pair2 requires):
// 4 VALU + 1 LDS + 3 SALU, where the compiler's form of `if ((mask >> lane) & 1) queue[pos] = ...` with its capacity test
// took ~9 VALU + ~12 SALU + 4 branches per ambiguous slot (6 % of the slots at C2 take this path). No capacity test:
// the queue is drained above 64 entries before every group and a group appends at most 4 x 64 (PK_QCAP = 320).
// Measured build against build (tools/ab_libs.py): C2 -2.0 %, C3 RDF -2.3 %, one-sweep RDF+CN -4.3 %; integers identical
// (soak_pk 1500 + soak_cn 800 against the oracle).
__device__ __forceinline__ void pk_push(PkCtx &p, unsigned long long mask, unsigned tag, int lane)
{
    unsigned rank, addr;
    const unsigned qa = p.qaddr + 4u * (unsigned)p.qn;
    asm volatile(
        "s_mov_b64 exec, %[m]\n\t"
        "v_mbcnt_lo_u32_b32 %[r], %[mlo], 0\n\t"
        "v_mbcnt_hi_u32_b32 %[r], %[mhi], %[r]\n\t"
        "v_lshl_add_u32 %[a], %[r], 2, %[qa]\n\t"
        "v_or_b32 %[r], %[tag], %[lane]\n\t"
        "ds_write_b32 %[a], %[r]\n\t"
        "s_mov_b64 exec, %[full]"
        : [r] "=&v"(rank), [a] "=&v"(addr)
        : [m] "s"(mask), [mlo] "s"((unsigned)mask), [mhi] "s"((unsigned)(mask >> 32)), [qa] "s"(qa), [tag] "s"(tag),
          [lane] "v"(lane), [full] "s"(p.full)
        : "memory");
    p.qn += __builtin_popcountll(mask);
}

// The four j atoms of one group against the wave's 64 i atoms.
// VAR: bit k = axis k takes the per-pair f32 wrap. jidx0 = index of the group's first atom in the frame's j set.
// PF: load the records at `next_p` into `next` (the group swept after this one) once the first operation on this
// group's records has been issued. Scalar loads return out of order, so the compiler can only wait for ALL of
// them (lgkmcnt(0)) before the first use of `rq`: that wait has to come before the prefetch is issued — the two
// empty asm statements pin that order (the first depends on dx, i.e. on a use of rq) — and is free, because rq was
// itself prefetched during the previous sweep.
// CNG: the group is near enough to the wave to hold pairs of a CN split bin: every pair the sweep bins is looked up in
// the flag bits, and the ones in a flagged word are queued as PK_CN_ONLY entries.
template <bool DIAG, int VAR, bool PF, bool CUTG, bool ROWS, bool CNG>
__device__ __forceinline__ void sweep_group_pk(const RelQ &rq, GroupIdx jidx0, int local0, PkCtx &p, const FastCtx &c,
                                               int lane_in_tile, int lane, const float *next_p, RelQ &next)
{
    unsigned row[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 ra = h ? rq.c : rq.a, rb = h ? rq.d : rq.b;
        const f32x2 xj = {ra[0], ra[1]};
        const f32x2 yj = {ra[2], ra[3]};
        const f32x2 zj = {rb[0], rb[1]};
        f32x2 dx = p.x2 - xj;
        if (ROWS && h == 0) {
            // class rows: the LDS address of the row of (ti, tj) from the table, for the four j atoms at once; the
            // lookups (and the wait for them, which the asm operands force here) come before the prefetch is issued,
            // because scalar loads and LDS share the lgkmcnt counter
            row[0] = c.rowtab_me[__float_as_uint(rq.b[2])];
            row[1] = c.rowtab_me[__float_as_uint(rq.b[3])];
            row[2] = c.rowtab_me[__float_as_uint(rq.d[2])];
            row[3] = c.rowtab_me[__float_as_uint(rq.d[3])];
            asm volatile("" : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3])::"memory");
        }
        if (PF && h == 0) {
            asm volatile("" ::"v"(dx) : "memory");
            next = load_relq(next_p);
            asm volatile("" ::: "memory");
        }
        f32x2 dy = p.y2 - yj, dz = p.z2 - zj;
        if (VAR & 1) dx = wrap_pk(dx, p.Lx2, p.iLx2);
        if (VAR & 2) dy = wrap_pk(dy, p.Ly2, p.iLy2);
        if (VAR & 4) dz = wrap_pk(dz, p.Lz2, p.iLz2);
        f32x2 rsq = dx * dx;
        rsq = __builtin_elementwise_fma(dy, dy, rsq);
        rsq = __builtin_elementwise_fma(dz, dz, rsq);
        // ordered rows: w = the bin-guess addend near + tj * row_len; class rows: w = the table offset of tj
        const float nearoff0 = ROWS ? c.near : rb[2], nearoff1 = ROWS ? c.near : rb[3];
        const unsigned rowbase0 = ROWS ? row[2 * h] : c.rowbase_me, rowbase1 = ROWS ? row[2 * h + 1] : c.rowbase_me;
        float r2a = rsq[0], r2b = rsq[1];
        if (DIAG) {  // i < j inside the diagonal tile
            r2a = local0 + 2 * h > lane_in_tile ? r2a : 3.0e38f;
            r2b = local0 + 2 * h + 1 > lane_in_tile ? r2b : 3.0e38f;
        }
        unsigned kaddr0 = 0, kaddr1 = 0;
        if (CNG) {
            // the split-bin word of the row (ti of this lane, tj of this j atom), looked up BEFORE the pair block so
            // that the LDS read runs under it. Ordered rows: tj from the addend near + tj * row_len (near < 1: the
            // truncated quotient is tj); class rows: the row table's own index. (Hoisting the four lookups of a
            // group costs 4 VGPRs, which the 80-register budget pays for with spills: measured slower.)
            kaddr0 = ROWS ? p.cn_kc_me[__float_as_uint(rb[2])] : p.cn_kc_me[(int)(nearoff0 * p.inv_row_len)];
            kaddr1 = ROWS ? p.cn_kc_me[__float_as_uint(rb[3])] : p.cn_kc_me[(int)(nearoff1 * p.inv_row_len)];
        }
        if constexpr (!CNG && PK_INLINE_PUSH) {
            if (h == 0)
                bin_pair2q<CUTG, 0>(r2a, r2b, p.rc2hi, c.gscale, nearoff0, nearoff1, c.near2, rowbase0, rowbase1, p.cut_lo,
                                    p.full, p.qn, p.qaddr, jidx0.jbase, jidx0.g, lane);
            else
                bin_pair2q<CUTG, 2>(r2a, r2b, p.rc2hi, c.gscale, nearoff0, nearoff1, c.near2, rowbase0, rowbase1, p.cut_lo,
                                    p.full, p.qn, p.qaddr, jidx0.jbase, jidx0.g, lane);
            continue;
        }
        PairOut o;
        const bool any_amb = bin_pair2<CUTG, CNG>(r2a, r2b, p.rc2hi, c.gscale, nearoff0, nearoff1, c.near2, rowbase0,
                                                  rowbase1, p.cut_lo, p.full, o);
        if (CNG && (o.done0 | o.done1)) {  // wave-uniform
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned long long done = u ? o.done1 : o.done0;
                const unsigned long long hm = __builtin_amdgcn_ballot_w64((u ? o.addr1 : o.addr0) == (u ? kaddr1 : kaddr0)) & done;
                if (hm) pk_push(p, hm, PK_CN_ONLY | ((unsigned)(jidx_here(jidx0) + 2 * h + u) << 6), lane);
            }
        }
        if (any_amb) {  // wave-uniform, rare: some lane's pair is inside the error band -> the exact chain, later
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // (the copy through a volatile asm keeps the per-lane test inside this branch: the compiler would
                // otherwise fold both conditions into one divergent branch and pay 3 VALU per pair for it)
                unsigned long long amb_in;
                asm volatile("s_mov_b64 %0, %1" : "=s"(amb_in) : "s"(u ? o.amb1 : o.amb0));
                if (amb_in) pk_push(p, amb_in, (unsigned)(jidx_here(jidx0) + 2 * h + u) << 6, lane);
            }
        }
    }
}

// Does the plain difference d' = xr_i - xr_j suffice on this axis for every pair of (wave box, group box)?
// All lanes of the wave must sit at the same image n relative to the centre c — the rint of both ends of the box
// agree, with 1e-3 of slack so that every lane's own f64 rint agrees too — and |d'| must stay within th.
__device__ __forceinline__ bool axis_plain(float wlo, float whi, float glo, float ghi, float c, float L, float iL,
                                           float th)
{
    const float n0 = __builtin_rintf((wlo - c) * iL - 1.0e-3f), n1 = __builtin_rintf((whi - c) * iL + 1.0e-3f);
    const float lo = (wlo - c) - n0 * L, hi = (whi - c) - n0 * L;
    const float dmax = __builtin_fmaxf(__builtin_fabsf(lo - (ghi - c)), __builtin_fabsf(hi - (glo - c)));
    // (c is rounded to f32 here: 2^-24 |c| on each difference — nothing next to the margin inside th for coordinates
    // within a few box lengths of the origin, but subtracted so that the test stays conservative for any offset)
    return n0 == n1 && dmax <= th - 1.0e-6f * __builtin_fabsf(c);
}

// One work item of the packed-f32 sweep (ordered-pair rows; atom-atom or atoms x sites): the 64 i atoms of wave `wq` of tile I of
// frame f against slice `split` of the tile's neighbour list.
template <bool CUTG, bool ROWS, bool CNG>
__device__ __forceinline__ int sj_item_pk(const PairArgs &a, FastCtx &c, const unsigned *s_row, unsigned *queue,
                                           const unsigned *cn_lds, int f, int I, int wq, int split, int lane)
{
    const long long n_pad = (long long)a.nTi * TILE, n_pad_j = (long long)a.nTj * TILE;
    const long long rowid = (long long)f * a.nTi + I;
    const int cnt = a.list_cnt[rowid];
    const unsigned short *row_list = a.list + rowid * a.nTj;
    const int t_begin = (int)((long long)split * cnt / a.jsplit);
    const int t_end = (int)((long long)(split + 1) * cnt / a.jsplit);
    AxisL L;
    L.Lx = a.box[3 * f];
    L.Ly = a.box[3 * f + 1];
    L.Lz = a.box[3 * f + 2];
    L.sx = L.sy = L.sz = 0.0;
    const double iLx = 1.0 / L.Lx, iLy = 1.0 / L.Ly, iLz = 1.0 / L.Lz;
    const double4 *ats = a.aos + (long long)f * n_pad;        // the i set (== the j set for atom-atom)
    const double4 *ats_j = a.aos_j + (long long)f * n_pad_j;  // the j set (atoms x sites: the sites)
    const int lane_in_tile = wq * 64 + lane;
    const long long ig = (long long)I * TILE + lane_in_tile;
    const bool real_i = ig < a.ni;
    {
        const int ti_me = (int)((unsigned)__double_as_longlong(ats[ig].w)) / a.n_ti;  // (pad records: type 0)
        c.rowtab_me = ROWS ? s_row + ti_me : nullptr;
        c.rowbase_me = c.lds_base + (unsigned)ti_me * (unsigned)a.row_mul * (unsigned)(a.nbins + 1) * 4u;
    }
    const long long w = ((long long)f * a.nTi + I) * (TILE / 64) + wq;
    // The wave's box and the group boxes as (centre, half extents; cull_boxes_kernel, cbox). The wave's:
    // wave-uniform, through the constant address space, so that it lives in SGPRs, not in 8 VGPRs.
    typedef const __attribute__((address_space(4))) f32x4 *cbox;
    const cbox wb = (cbox)(unsigned long long)(a.wsph + 2 * w);
    const f32x4 wc = wb[0], wh = wb[1];
    const float4 *gb_f = a.gsph4 + (long long)f * a.nTj * (TILE / SJ_GROUP) * 2;
    const float fLx = (float)L.Lx, fLy = (float)L.Ly, fLz = (float)L.Lz;
    const float r_cut = __builtin_sqrtf((float)a.rc2);
    const float fiLx = (float)iLx, fiLy = (float)iLy, fiLz = (float)iLz;
    // What depends on (wave box, tile centre) only is computed ONCE per neighbour tile, lanes 0 .. 2 taking one axis
    // each (round 3 computed it in every lane, per group: two thirds of the axis tests' instructions), and handed to
    // the wave through v_readlane. Per-axis constants of this lane's axis:
    const int ax = lane < 3 ? lane : 0;
    const float L_a = (float)a.box[3 * f + ax], iL_a = (float)(1.0 / a.box[3 * f + ax]);
    // |d'| up to here needs no per-pair wrap on that axis (see above); the margin covers the f32 box arithmetic
    const float th_a = L_a - r_cut - (1.0e-3f * L_a + 1.0e-3f);
    const float cw_a = reinterpret_cast<const float *>(a.wsph + 2 * w)[ax];
    const float hw_a = reinterpret_cast<const float *>(a.wsph + 2 * w)[4 + ax];
    const float cap_a = a.s_cap * (1.0f - 1.0e-6f);
    PkCtx p;
    p.Lx2 = f32x2{fLx, fLx};
    p.Ly2 = f32x2{fLy, fLy};
    p.Lz2 = f32x2{fLz, fLz};
    p.iLx2 = f32x2{(float)iLx, (float)iLx};
    p.iLy2 = f32x2{(float)iLy, (float)iLy};
    p.iLz2 = f32x2{(float)iLz, (float)iLz};
    p.rc2hi = a.rc2hi;
    p.cut_lo = a.cut_lo;
    asm volatile("s_mov_b64 %0, exec" : "=s"(p.full));
    p.rowtab = ROWS ? s_row : nullptr;
    p.queue = queue;
    p.qaddr = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned *)queue);
    p.qn = 0;
    p.lost = a.overflow + 1;
    p.ats_i = ats + (long long)I * TILE + wq * 64;
    p.ats_j = ats_j;
    p.Lx = L.Lx;
    p.Ly = L.Ly;
    p.Lz = L.Lz;
    p.rc2 = a.rc2;
    p.n_ti = a.n_ti;
    p.row_mul = a.row_mul;
    if (CNG) {
        // LDS behind the queues: split-word address per row | cutoff^2 per row | (class rows) split-word address [tj][ti]
        const int rows = ROWS ? a.n_cls + 1 : a.n_rows_ord;
        const int ti_me = (int)((unsigned)__double_as_longlong(ats[ig].w)) / a.n_ti;
        p.cn_kc = cn_lds;
        p.cn_c2 = reinterpret_cast<const double *>(cn_lds + ((rows + 1) & ~1));
        p.cn_kc_me = ROWS ? cn_lds + ((rows + 1) & ~1) + 2 * rows + ti_me : cn_lds + ti_me * a.row_mul;
        p.cn_base = c.lds_base + (unsigned)(rows * (a.nbins + 1)) * 4u;
    }
    p.inv_row_len = 1.0f / (float)(a.nbins + 1);
    const float *rel_f = a.rel + (long long)f * n_pad_j * 4;  // 4 floats per j atom
    const double *cen_f = a.cen + (long long)f * a.nTj * 8;
    const int I_u = __builtin_amdgcn_readfirstlane(I);  // (uniform anyway; this tells the compiler: masks are selected on it)
    const float reach2 = a.reach * a.reach, cn_reach2 = a.cn_reach * a.cn_reach;
    for (int t = t_begin; t < t_end; ++t) {
        const int J = __builtin_amdgcn_readfirstlane((int)row_list[t]);
        // every lane tests one 4-atom group box of the tile against this wave's box: per axis the distance of the
        // centres at the nearest image minus the half extents (a lower bound of the reference's per-axis distance
        // min(|d|, ||d| - L|): the nearest image is never farther than the single wrap's)
        const float4 gc = gb_f[((long long)J * (TILE / SJ_GROUP) + lane) * 2];
        const float4 gh = gb_f[((long long)J * (TILE / SJ_GROUP) + lane) * 2 + 1];
        const double *cb = cen_f + (long long)J * 8;
        const double cd_a = cb[ax], hd_a = cb[3 + ax];  // (lanes 0 .. 2: the tile's centre and half extent on their axis)
        const float sx = wh[0] + gh.x, sy = wh[1] + gh.y, sz = wh[2] + gh.z;
        const float dx = wc[0] - gc.x, dy = wc[1] - gc.y, dz = wc[2] - gc.z;
        const float gx = __builtin_fmaxf(__builtin_fabsf(__builtin_fmaf(-__builtin_rintf(dx * fiLx), fLx, dx)) - sx, 0.f);
        const float gy = __builtin_fmaxf(__builtin_fabsf(__builtin_fmaf(-__builtin_rintf(dy * fiLy), fLy, dy)) - sy, 0.f);
        const float gz = __builtin_fmaxf(__builtin_fabsf(__builtin_fmaf(-__builtin_rintf(dz * fiLz), fLz, dz)) - sz, 0.f);
        const float g2 = gx * gx + gy * gy + gz * gz;
        unsigned long long km = __builtin_amdgcn_ballot_w64(g2 < reach2);  // (a box without atoms: g2 ~ 1e36)
        // the diagonal tile counts i < j: the groups below this wave's first atom hold no such pair
        if (a.tri && J == I_u) km &= ~0ull << (wq * (64 / SJ_GROUP));
        if (!km) continue;
        // CNG: groups whose box comes within the largest split bin of the wave's box are swept with the flag check
        const unsigned long long nm = CNG ? __builtin_amdgcn_ballot_w64(g2 < cn_reach2) : 0ull;
        const float *rtile = rel_f + (long long)J * TILE * 4;
        const bool diag = a.tri && J == I_u;
        // (wave box, tile centre c), one axis per lane: do all lanes of the wave sit at the same image n relative to c
        // — the rint of both ends of the box agree, with 1e-3 of slack so that every lane's own f64 rint agrees too —,
        // the wave's centre moved there, the threshold for |d'| (-1 = no group is plain on this axis), and whether
        // the error bound covers the tile: every |x_i - x_j| < 1.5 L (single wrap = nearest image), |xr_i| + |xr_j|
        // within s_cap, |d'| <= L/2 + h < 1.5 L for the per-pair wrap.
        const float c_a = (float)cd_a, he_a = (float)hd_a;
        const float rel_a = cw_a - c_a;
        const float n0 = __builtin_rintf(__builtin_fmaf(rel_a - hw_a, iL_a, -1.0e-3f));
        const float n1 = __builtin_rintf(__builtin_fmaf(rel_a + hw_a, iL_a, 1.0e-3f));
        const bool same = n0 == n1;
        const float cwn_a = __builtin_fmaf(-n0, L_a, cw_a);
        // (c is rounded to f32 here: 2^-24 |c| on each difference — nothing next to the margin inside th for
        // coordinates within a few box lengths of the origin, but subtracted so that the test stays conservative)
        const float thc_a = same ? __builtin_fmaf(-1.0e-6f, __builtin_fabsf(c_a), th_a) : -1.0f;
        // the largest |xr_i| of the wave: its box at image n0, or L/2 when the wave straddles a wrap boundary
        const float wmax_a = same ? __builtin_fabsf(__builtin_fmaf(-n0, L_a, rel_a)) + hw_a : 0.5000001f * L_a;
        const bool ok_a = (__builtin_fabsf(rel_a) + hw_a + he_a < 1.49f * L_a) && (wmax_a + he_a < cap_a) &&
                          (he_a < 0.9f * L_a);
        const bool covered = (__builtin_amdgcn_ballot_w64(ok_a) & 7ull) == 7ull;
#define PK_LANE(V, K) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(V), K))
        const float cwnx = PK_LANE(cwn_a, 0), cwny = PK_LANE(cwn_a, 1), cwnz = PK_LANE(cwn_a, 2);
        const float thcx = PK_LANE(thc_a, 0), thcy = PK_LANE(thc_a, 1), thcz = PK_LANE(thc_a, 2);
#undef PK_LANE
        // per group and axis: does the plain difference d' = xr_i - xr_j suffice for every pair of (wave box, group
        // box)? |d'| <= |c_wave - n L - c_group| + h_wave + h_group must stay within th. The axes that fail take the
        // per-pair wrap (variant bits 1 / 2 / 4); the masks of the groups per variant are scalar arithmetic from here.
        const unsigned long long mx = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(cwnx - gc.x) + sx <= thcx));
        const unsigned long long my = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(cwny - gc.y) + sy <= thcy));
        const unsigned long long mz = __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(cwnz - gc.z) + sz <= thcz));
        const unsigned long long wrapm = mx | my | mz;
        auto variant = [&](unsigned A) {
            return km & ((A & 1u) ? mx : ~mx) & ((A & 2u) ? my : ~my) & ((A & 4u) ? mz : ~mz);
        };
        {
            if (!covered) {
                // not covered (rare: atoms box lengths outside the cell): every pair of the kept groups goes to the
                // queue, i.e. to the exact f64 chain with the general wrap (valid for every d)
                unsigned long long mk = km;
                while (mk) {
                    const int g = __builtin_ctzll(mk);
                    mk &= mk - 1;
#pragma unroll 1
                    for (int u = 0; u < SJ_GROUP; ++u) {
                        if (p.qn > 64) pk_drain<CNG>(p, c, lane);
                        const bool want = real_i && (!diag || g * SJ_GROUP + u > lane_in_tile);
                        const unsigned long long wm = __builtin_amdgcn_ballot_w64(want);
                        if (want)
                            p.queue[p.qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u))] =
                                ((unsigned)(J * TILE + g * SJ_GROUP + u) << 6) | (unsigned)lane;
                        p.qn += __builtin_popcountll(wm);
                    }
                }
                continue;
            }
            // this lane's i atom at the periodic image nearest to the tile's centre
            // (the i atom is re-read per tile rather than kept live through the pair loop: 8 VGPRs)
            const double4 me = ats[ig];
            const double qx = me.x - cb[0], qy = me.y - cb[1], qz = me.z - cb[2];
            const double wx = __builtin_fma(-__builtin_rint(qx * iLx), L.Lx, qx);
            const double wy = __builtin_fma(-__builtin_rint(qy * iLy), L.Ly, qy);
            const double wz = __builtin_fma(-__builtin_rint(qz * iLz), L.Lz, qz);
            {
                const float xr = real_i ? (float)wx : -1.0e18f;
                const float yr = real_i ? (float)wy : -1.0e18f;
                const float zr = real_i ? (float)wz : -1.0e18f;
                p.x2 = f32x2{xr, xr};
                p.y2 = f32x2{yr, yr};
                p.z2 = f32x2{zr, zr};
            }
            const int jbase = J * TILE;
#define PK_DRAIN_CHECK() \
    if (p.qn > 64) pk_drain<CNG>(p, c, lane)
            // the common variant (no per-pair wrap), software-pipelined: the records of the next group are loaded
            // while the current group is swept; two buffers, loop unrolled by two (no register copies). With CNG the
            // groups near the wave run the same pipeline in its split-bin-checking form (an unpipelined near loop
            // exposed every group's record load: a near slot then cost ~2.5 slots).
// (mask walk in scalar instructions that exist: s_ff1 gives -1 for an empty mask, & 63 turns that into a group whose
// records are merely prefetched and never swept; s_bitset0 clears the bit in one instruction instead of the three of
// mk &= mk - 1; the record address is 32-bit offset arithmetic)
#define PK_NEXT(G)                                                                                                      \
    {                                                                                                                   \
        asm("s_ff1_i32_b64 %0, %1" : "=s"(G) : "s"(mk));                                                                \
        G &= 63;                                                                                                        \
        asm("s_bitset0_b64 %0, %1" : "+s"(mk) : "s"(G));                                                                \
    }
#define PK_REC(G) (const float *)((const char *)rtile + ((unsigned)(G) << 6))
#define PK_PIPELINED(MASK, CN, VAR, DG)                                                                                 \
    {                                                                                                                   \
        unsigned long long mk = (MASK);                                                                                 \
        if (mk) {                                                                                                       \
            int gA, gB;                                                                                                 \
            PK_NEXT(gA)                                                                                                 \
            RelQ qA = load_relq(PK_REC(gA)), qB;                                                                        \
            for (;;) {                                                                                                  \
                PK_DRAIN_CHECK();                                                                                       \
                const bool moreB = mk != 0;                                                                             \
                PK_NEXT(gB)                                                                                             \
                sweep_group_pk<DG, VAR, true, CUTG, ROWS, CN>(qA, GroupIdx{jbase, gA}, gA * SJ_GROUP, p, c,             \
                                                                 lane_in_tile, lane, PK_REC(gB), qB);                   \
                if (!moreB) break;                                                                                      \
                PK_DRAIN_CHECK();                                                                                       \
                const bool moreA = mk != 0;                                                                             \
                PK_NEXT(gA)                                                                                             \
                sweep_group_pk<DG, VAR, true, CUTG, ROWS, CN>(qB, GroupIdx{jbase, gB}, gB * SJ_GROUP, p, c,             \
                                                                 lane_in_tile, lane, PK_REC(gA), qA);                   \
                if (!moreA) break;                                                                                      \
            }                                                                                                           \
        }                                                                                                               \
    }
            if (diag) {
                // the diagonal tile (i < j inside the tile): its plain groups through the same pipeline (round 4; they
                // were swept one by one with their record loads exposed)
                PK_PIPELINED(km & ~wrapm & ~nm, false, 0, true)
            } else {
                PK_PIPELINED(km & ~wrapm & ~nm, false, 0, false)
                if constexpr (CNG) PK_PIPELINED(km & ~wrapm & nm, true, 0, false)
                // the groups that take the per-pair wrap on some axis run the same software pipeline, variant by variant
                // (round 3: their records were loaded and waited for group by group before — at C2, where a third of the
                // swept groups wrap on some axis, 3.04 -> 2.82 ms build against build; C3, which has none, -0.9 %)
                if (km & wrapm & ~nm) {
                    PK_PIPELINED(variant(1u) & ~nm, false, 1, false)
                    PK_PIPELINED(variant(2u) & ~nm, false, 2, false)
                    PK_PIPELINED(variant(4u) & ~nm, false, 4, false)
                    PK_PIPELINED(variant(3u) & ~nm, false, 3, false)
                    PK_PIPELINED(variant(5u) & ~nm, false, 5, false)
                    PK_PIPELINED(variant(6u) & ~nm, false, 6, false)
                    PK_PIPELINED(variant(7u) & ~nm, false, 7, false)
                }
            }
#undef PK_PIPELINED
#undef PK_NEXT
#undef PK_REC
            // what is left — the diagonal tile, and wrapping groups the CN check walks —: not pipelined
#define PK_SWEEP_CASES(CN)                                                                                             \
    switch (A) {                                                                                                       \
    case 1: sweep_group_pk<false, 1, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 2: sweep_group_pk<false, 2, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 3: sweep_group_pk<false, 3, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 4: sweep_group_pk<false, 4, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 5: sweep_group_pk<false, 5, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 6: sweep_group_pk<false, 6, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 7: sweep_group_pk<false, 7, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break; \
    case 9: sweep_group_pk<true, 0, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break;  \
    case 8: sweep_group_pk<true, 7, false, CUTG, ROWS, CN>(q, j0, l0, p, c, lane_in_tile, lane, nullptr, qnone); break;  \
    default: break;                                                                                                    \
    }
            // (the diagonal tile, i < j inside the tile: 9 = plain where every axis qualifies, 8 = the wrap on all axes)
            if (!diag && !(CNG && (km & wrapm & nm))) continue;
            for (unsigned A = diag ? 8u : 1u; A <= (diag ? 9u : 7u); ++A) {
                unsigned long long mk = diag ? (A == 9u ? km & ~wrapm & nm : km & wrapm) : variant(A) & nm;
                while (mk) {
                    const int g = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mk));
                    mk &= mk - 1;
                    RelQ q = load_relq(rtile + g * SJ_GROUP * 4), qnone;
                    PK_DRAIN_CHECK();
                    const GroupIdx j0{jbase, g};
                    const int l0 = g * SJ_GROUP;
                    if constexpr (CNG) {
                        if ((nm >> g) & 1ull) {
                            PK_SWEEP_CASES(true)
                            continue;
                        }
                    }
                    PK_SWEEP_CASES(false)
                }
            }
#undef PK_SWEEP_CASES
#undef PK_DRAIN_CHECK
        }
    }
    if (p.qn > 0) pk_drain<CNG>(p, c, lane);
    return t_end > t_begin ? t_end - t_begin : 0;
}

// Assertion at the top of every work-loop iteration: the whole wave is here (the item index is drawn by lane 0
// and broadcast) and the wave has not drawn more items than exist. A violation would make the wave spin on
// item 0; it is recorded in overflow[2] and ends the loop, and the host turns it into an error.
__device__ __forceinline__ bool work_loop_sane(const PairArgs &a, long long iter, long long n_items)
{
    if (__builtin_amdgcn_ballot_w64(true) == ~0ull && iter <= n_items + 1) return true;
    atomicOr(a.overflow + 2, __builtin_amdgcn_ballot_w64(true) == ~0ull ? 2ull : 1ull);
    return false;
}

// PERSIST = true (frame-summed output): the grid is one resident set of blocks; every WAVE draws items
// (frame, tile, wave, list slice) from its XCD's counter — frames stay dealt to XCDs (f % 8) so a frame's
// records live in one L2 — and the block flushes its LDS histograms once, when its four waves have run
// dry. Every wave leaves the loop as soon as the counter passes the item count.
// PERSIST = false (per-frame output): block = (frame, tile, list slice), one flush per block.
// CNG (packed-f32 modes only): coordination numbers from the same sweep; one split counter per row sits right behind
// the histogram rows in LDS and travels with them through the slices. (Register budget as the plain variant, 6 waves per
// SIMD: the CNG variant then spills 18 dwords; a 5-wave budget without spills measured 1.40x RDF alone instead of 1.15x.)
template <int MODE, bool PERSIST, bool CNG = false, bool BIG = false>
__global__ __launch_bounds__(BIG ? PK_BIG_THREADS : MODE >= 3 ? PK_THREADS : TILE, BIG ? 4 : MODE >= 3 ? PK_WAVES_PER_SIMD : 1) void
pair_hist_sj_kernel(const PairArgs a)
{
    // threads per block: the waves are independent (they share only the LDS histogram), so the block size is free.
    // MODE 3 runs 8 waves per block: LDS (one histogram per block) then allows 6 waves per SIMD instead of 5, which
    // this latency-bound sweep (scalar record loads from L2) converts into VALU utilisation.
    constexpr int BS = BIG ? PK_BIG_THREADS : MODE >= 3 ? PK_THREADS : TILE;
    // packed-f32 modes: 3 ordered rows, 4 = 3 with the cutoff guard (CUTG), 5 class rows + row table, 6 = 5 with CUTG
    constexpr bool ORDERED = MODE >= 2 && MODE <= 4;
    constexpr bool PK_ROWS = MODE >= 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const long long bid = blockIdx.x;
    const int xcd = (int)(bid & 7);

    // ---- LDS: hist | (CN edges) | row table ----
    // MODE 2: one row per ORDERED type pair (ti, tj), addressed without a table (see sweep_group_sj)
    const int row_len = a.nbins + 1;
    const int n_rows = ORDERED ? a.n_rows_ord : a.n_cls + 1;
    const int hist_words = n_rows * (row_len + (CNG ? 1 : 0));  // CNG: + one split counter per row, behind the rows
    unsigned *s_hist = reinterpret_cast<unsigned *>(smem);
    size_t off = ((size_t)hist_words * 4 + 15) & ~size_t(15);
    double *s_edges = reinterpret_cast<double *>(smem + off);
    off += MODE == 1 ? (((size_t)(a.nbins + 2) * 8 + 15) & ~size_t(15)) : 0;
    unsigned *s_row = reinterpret_cast<unsigned *>(smem + off);  // class rows: the row table [n_tj][n_ti]
    // packed-f32 modes: the waves' queues of deferred pairs, behind the row table when there is one
    unsigned *s_queue = s_row + (PK_ROWS ? (a.n_ti * a.n_tj + 3) / 4 * 4 : 0);
    // CNG, behind the queues: the LDS address of every row's split-bin word (the host gives word indices; -1: none),
    // the rows' cutoffs^2 (8-byte aligned) and, for class rows, the split-word addresses as a table [n_tj][n_ti]
    unsigned *s_cn = s_queue + (MODE >= 3 ? (BS / 64) * PK_QSTRIDE : 0);
    const unsigned lds_base =
        (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)smem;
    // Overflow guard of the 32-bit histogram words: the block counts the neighbour tiles its waves sweep (one tile = at
    // most 64 x 256 increments of any one word); past a.guard_tiles it raises overflow[3] and the host splits the batch.
    unsigned *s_guard = reinterpret_cast<unsigned *>(smem + a.guard_off);
    if (tid == 0) *s_guard = 0u;
    for (int k = tid; k < hist_words; k += BS) s_hist[k] = 0u;
    if (!ORDERED)
        for (int k = tid; k < a.n_ti * a.n_tj; k += BS) {
            const int ti = k % a.n_ti, tj = k / a.n_ti;
            const unsigned cl = a.cls[ti * a.n_tj + tj];
            s_row[k] = lds_base + (cl == 0xFFu ? (unsigned)a.n_cls : cl) * (unsigned)row_len * 4u;
        }
    FastCtx c;
    c.hist = s_hist;
    c.edges = a.edges;
    if (MODE == 1) {
        for (int k = tid; k <= a.nbins + 1; k += BS) s_edges[k] = a.edges[k];
        c.edges = s_edges;
    }
    c.gscale = a.gscale;
    if (MODE >= 2) {
        // the bin guess is fma(sqrt, gscale, addend) with the addend in an SGPR (it belongs to the j atom): a VOP3
        // may read one SGPR, so gscale has to live in a VGPR or every guess pays a v_mov
        float gs;
        asm volatile("v_mov_b32 %0, %1" : "=v"(gs) : "s"(a.gscale));
        c.gscale = gs;
    }
    c.near = MODE >= 2 ? a.near : (float)a.nbins * 1.0e-6f + 1.0e-5f;
    c.near2 = 2.0f * c.near;
    c.nbins = a.nbins;
    c.lds_base = lds_base;
    c.rowbase_me = lds_base;
    if (CNG) {
        const int pad = (n_rows + 1) & ~1;
        for (int k = tid; k < n_rows; k += BS) {
            const int w = (int)a.cn_tab[k];
            s_cn[k] = w < 0 ? ~0u : lds_base + (unsigned)w * 4u;
        }
        for (int k = tid; k < 2 * n_rows; k += BS) s_cn[pad + k] = a.cn_tab[pad + k];
        if (PK_ROWS)
            for (int k = tid; k < a.n_ti * a.n_tj; k += BS) {
                const int ti = k % a.n_ti, tj = k / a.n_ti;
                const unsigned cl = a.cls[ti * a.n_tj + tj];
                const int w = (int)a.cn_tab[cl == 0xFFu ? a.n_cls : (int)cl];
                s_cn[pad + 2 * n_rows + k] = w < 0 ? ~0u : lds_base + (unsigned)w * 4u;
            }
    }
    __syncthreads();  // tables ready; from here on the waves do not synchronise until the flush

    const int lane = tid & 63;
    int f_out = 0;
    if (PERSIST) {
        const int nfx = a.n_frames > xcd ? (a.n_frames - xcd + 7) / 8 : 0;  // frames of this XCD
        const int ipf = a.nTi * (TILE / 64) * a.jsplit;                       // items per frame
        const long long n_items = (long long)nfx * ipf;
        long long guard = 0;
        for (;;) {
            unsigned it = 0;
            if (!work_loop_sane(a, ++guard, n_items)) break;
            if (lane == 0) it = atomicAdd(&a.work[xcd], 1u);
            it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);
            if ((long long)it >= n_items) break;
            const int fx = (int)(it / (unsigned)ipf), r = (int)(it % (unsigned)ipf);
            const int split = r % a.jsplit, wI = r / a.jsplit;
            int tiles;
            if (MODE >= 3)
                tiles = sj_item_pk<MODE == 4 || MODE == 6, PK_ROWS, CNG>(a, c, s_row, s_queue + (tid >> 6) * PK_QSTRIDE, s_cn,
                                                                         fx * 8 + xcd, wI >> 2, wI & 3, split, lane);
            else
                tiles = sj_item<MODE >= 3 ? 2 : MODE>(a, c, s_row, fx * 8 + xcd, wI >> 2, wI & 3, split, lane);
            if (lane == 0) atomicAdd(s_guard, (unsigned)tiles);
        }
    } else {
        // a.blocks_per_frame blocks share one frame and flush once each into the frame's row
        // (frames stay dealt to XCDs, f % 8 = XCD, so that a frame's records are fetched into one L2)
        const int f = (int)((bid >> 3) / a.blocks_per_frame) * 8 + xcd;
        f_out = f < a.n_frames ? f : 0;
        const unsigned ipf = f < a.n_frames ? (unsigned)(a.nTi * (TILE / 64) * a.jsplit) : 0u;
        int guard = 0;
        for (;;) {  // the frame's blocks draw its wave items from the frame's counter (integer sums: any order)
            unsigned it = 0;
            if (ipf == 0u) break;
            if (!work_loop_sane(a, ++guard, ipf)) break;
            if (lane == 0) it = atomicAdd(&a.work[f], 1u);
            it = (unsigned)__builtin_amdgcn_readfirstlane((int)it);
            if (it >= ipf) break;
            const int split = (int)(it % (unsigned)a.jsplit), wI = (int)(it / (unsigned)a.jsplit);
            int tiles;
            if (MODE >= 3)
                tiles = sj_item_pk<MODE == 4 || MODE == 6, PK_ROWS, CNG>(a, c, s_row, s_queue + (tid >> 6) * PK_QSTRIDE, s_cn, f, wI >> 2,
                                                                         wI & 3, split, lane);
            else
                tiles = sj_item<MODE >= 3 ? 2 : MODE>(a, c, s_row, f, wI >> 2, wI & 3, split, lane);
            if (lane == 0) atomicAdd(s_guard, (unsigned)tiles);
        }
    }

    // ---- flush: the block's LDS histogram goes to its own slice with plain coalesced stores (device-scope
    // atomics on rows spread over HBM cost ~40 ps each: 10^7 of them per launch were 6 % of the kernel);
    // merge_slices_kernel adds the slices up afterwards ----
    (void)f_out;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && *s_guard > a.guard_tiles) atomicOr(a.overflow + 3, 1ull);
    unsigned *slice = a.slices + (size_t)bid * (size_t)hist_words;
    for (int w = tid; w < hist_words; w += BS) slice[w] = s_hist[w];
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

// Row sums -> the reference's outputs, on the device (frame-summed RDF for collectives without a host round trip):
// out = full [nbins] | part [n_rel][nbins] | overflow [1], ADDED to what is there (batches of frames accumulate).
//   full[b]     += 2 * sum over counted rows            (rdf_cn.py:85-86)
//   part[kl][b] += mult[kl] * sum over the rows of relation kl's class   (rdf_cn.py:87-96; mult 2 when a == b)
// blockIdx.y = 0: full, 1..n_rel: part, n_rel + 1: overflow. rowcls[r] = class of row r (-1: not counted).
__global__ void derive_rdf_kernel(const unsigned long long *__restrict__ rows, int n_rows, int nbins,
                                  const int *__restrict__ rowcls, int n_rel, const int *__restrict__ relcls,
                                  const int *__restrict__ relmult, const unsigned long long *__restrict__ guard,
                                  unsigned long long *__restrict__ out)
{
    if (*guard) return;  // a block may have wrapped a 32-bit word: the host runs this batch again in halves
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int row_len = nbins + 1;
    if (y == n_rel + 1) {
        if (b == 0) {
            unsigned long long ov = 0;
            for (int r = 0; r < n_rows; ++r) ov += rows[(size_t)r * row_len + nbins];
            out[(size_t)(1 + n_rel) * nbins] += ov;
        }
        return;
    }
    if (b >= nbins) return;
    unsigned long long sum = 0;
    if (y == 0) {
        for (int r = 0; r < n_rows; ++r)
            if (rowcls[r] >= 0) sum += rows[(size_t)r * row_len + b];
        out[b] += 2ull * sum;
    } else {
        const int want = relcls[y - 1];
        if (want < 0) return;
        for (int r = 0; r < n_rows; ++r)
            if (rowcls[r] == want) sum += rows[(size_t)r * row_len + b];
        out[(size_t)y * nbins + b] += (unsigned long long)relmult[y - 1] * sum;
    }
}

}  // namespace

void launch_derive_rdf(hipStream_t stream, const unsigned long long *rows, int n_rows, int nbins, const int *rowcls,
                       int n_rel, const int *relcls, const int *relmult, const unsigned long long *guard,
                       unsigned long long *out)
{
    hipLaunchKernelGGL(derive_rdf_kernel, dim3((unsigned)((nbins + 127) / 128), (unsigned)(n_rel + 2)), dim3(128), 0,
                       stream, rows, n_rows, nbins, rowcls, n_rel, relcls, relmult, guard, out);
}

size_t lds_bytes_sj_ordered(int nbins, int n_rows)
{
    return (((size_t)n_rows * (nbins + 1) * 4 + 15) & ~size_t(15)) + 16;
}

int sj_block_threads(int mode, bool big) { return big ? PK_BIG_THREADS : mode >= 3 ? PK_THREADS : TILE; }

// CN tables behind the queues: split-word address per row (padded to 8 bytes) + one double per row + (class rows)
// the split-word addresses as a table [n_tj][n_ti]
static size_t cn_table_bytes(int rows, int n_tab)
{
    return ((((size_t)rows + 1) & ~size_t(1)) + 2 * (size_t)rows + (size_t)n_tab) * 4;
}

size_t lds_bytes_sj_pk_rows(int nbins, int n_cls, int n_ti, int n_tj, int n_cn)
{
    const size_t hist = ((size_t)(n_cls + 1) * (nbins + 1 + (n_cn ? 1 : 0)) * 4 + 15) & ~size_t(15);
    return hist + (size_t)((n_ti * n_tj + 3) / 4 * 4) * 4 + (size_t)(PK_THREADS / 64) * PK_QSTRIDE * 4 +
           (n_cn ? cn_table_bytes(n_cls + 1, n_ti * n_tj) : 0);
}

size_t lds_bytes_sj_pk(int nbins, int n_rows, int n_cn, bool big)
{
    return (((size_t)n_rows * (nbins + 1 + (n_cn ? 1 : 0)) * 4 + 15) & ~size_t(15)) +
           (size_t)((big ? PK_BIG_THREADS : PK_THREADS) / 64) * PK_QSTRIDE * 4 + (n_cn ? cn_table_bytes(n_rows, 0) : 0);
}

// Error bound of the packed-f32 bin guess, in bins (see the MODE 3 header). u = 2^-24 (f32 round to nearest).
//   per axis   d32 = fl(xr_i - xr_j): the two inputs carry u |xr_i| + u |xr_j| <= u s_cap, the subtraction adds
//              u |d'| <= u s_cap (|d'| <= |xr_i| + |xr_j|)
//              f32 wrap of an axis: 1-Lipschitz in d'; adds |L32 - L| <= u L and u |result|
//   rsq32      one rounded product and two fused multiply-adds: relative 3u
//   r32        v_sqrt_f32 (1 ulp = 2u): relative 1.5u + 2u
//   => |r32 - sqrt(rsq_ref)| <= sqrt(3) u (2 s_cap + l_max) + 5.5 u r   (the three axes add in quadrature; r <= the
//      cutoff for every pair whose bin matters; the reference's own f64 roundings are 2^-29 of these terms)
//   g          1/ddr rounded to f32 (u nbins), the addend near + tj*row_len and the fma each rounded at the size of
//              the largest guess (half an ulp each)
// 5 % head room on top; the caller uses near = 2 err + slack (one err for the guess, one for the f32 pre-filter
// rsq32 < rc2hi, which has to let every in-cutoff pair through and still keep the rejected ones inside the band of
// the cutoff's edge).
double pk_error_bound(double r_cut, double bin_size, int nbins, int n_tj, double s_cap, double l_max)
{
    const double u = std::ldexp(1.0, -24);
    const double e_r = std::sqrt(3.0) * u * (2.0 * s_cap + l_max) + 5.5 * u * r_cut;
    const double maxg = (double)n_tj * (nbins + 1) + 2.0;
    const double ulp = std::ldexp(1.0, (int)std::floor(std::log2(maxg)) - 23);
    return 1.05 * (e_r / bin_size + u * (double)(nbins + 1)) + ulp;
}

size_t lds_bytes_sj(int nbins, int n_cls, int n_ti, int n_tj, bool mode_cn)
{
    size_t off = ((size_t)(n_cls + 1) * (nbins + 1) * 4 + 15) & ~size_t(15);
    off += mode_cn ? (((size_t)(nbins + 2) * 8 + 15) & ~size_t(15)) : 0;
    off += (size_t)n_ti * n_tj * 4;
    return (off + 15) & ~size_t(15);
}

PairKernel sj_kernel(int mode, bool persist, bool cn, bool big, const char **name)
{
#define MD_PICK(...) (*name = #__VA_ARGS__, __VA_ARGS__)
    if (big && !cn && mode == 4) return persist ? MD_PICK(pair_hist_sj_kernel<4, true, false, true>) : MD_PICK(pair_hist_sj_kernel<4, false, false, true>);
    if (big && !cn && mode == 3) return persist ? MD_PICK(pair_hist_sj_kernel<3, true, false, true>) : MD_PICK(pair_hist_sj_kernel<3, false, false, true>);
    if (cn && mode == 6) return persist ? MD_PICK(pair_hist_sj_kernel<6, true, true>) : MD_PICK(pair_hist_sj_kernel<6, false, true>);
    if (cn && mode == 5) return persist ? MD_PICK(pair_hist_sj_kernel<5, true, true>) : MD_PICK(pair_hist_sj_kernel<5, false, true>);
    if (cn && mode == 4) return persist ? MD_PICK(pair_hist_sj_kernel<4, true, true>) : MD_PICK(pair_hist_sj_kernel<4, false, true>);
    if (cn && mode == 3) return persist ? MD_PICK(pair_hist_sj_kernel<3, true, true>) : MD_PICK(pair_hist_sj_kernel<3, false, true>);
    if (mode == 6) return persist ? MD_PICK(pair_hist_sj_kernel<6, true, false>) : MD_PICK(pair_hist_sj_kernel<6, false, false>);
    if (mode == 5) return persist ? MD_PICK(pair_hist_sj_kernel<5, true, false>) : MD_PICK(pair_hist_sj_kernel<5, false, false>);
    if (mode == 4) return persist ? MD_PICK(pair_hist_sj_kernel<4, true, false>) : MD_PICK(pair_hist_sj_kernel<4, false, false>);
    if (mode == 3) return persist ? MD_PICK(pair_hist_sj_kernel<3, true, false>) : MD_PICK(pair_hist_sj_kernel<3, false, false>);
    if (mode == 2) return persist ? MD_PICK(pair_hist_sj_kernel<2, true, false>) : MD_PICK(pair_hist_sj_kernel<2, false, false>);
    if (mode == 1) return persist ? MD_PICK(pair_hist_sj_kernel<1, true, false>) : MD_PICK(pair_hist_sj_kernel<1, false, false>);
    return persist ? MD_PICK(pair_hist_sj_kernel<0, true, false>) : MD_PICK(pair_hist_sj_kernel<0, false, false>);
#undef MD_PICK
}

void launch_merge_slices(hipStream_t stream, const unsigned *slices, int hist_words, long long n_blocks, int per_frame,
                         int bpf, unsigned grid_y, unsigned long long *rows)
{
    hipLaunchKernelGGL(merge_slices_kernel, dim3((unsigned)((hist_words + 255) / 256), grid_y), dim3(256), 0, stream,
                       slices, hist_words, n_blocks, per_frame, bpf, rows);
}

}  // namespace mdpair
