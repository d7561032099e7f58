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
// Kernels live in their own translation units (shared types: pair_common.h):
//   pair_sj.hip     scalar-j kernel — the default of the spatially culled sweep (one i atom per lane, j atoms
//                   through scalar loads, wrap decisions hoisted per (wave box, group box), persistent grid)
//   pair_cull.hip   the spatial pre-pass: Hilbert sort, bounding boxes, neighbour-tile lists
//   pair_dense.hip  LDS-tile kernels: dense half-shell sweep for small frames, the edge-table kernel of the
//                   first commit (A/B baseline, fallback for > 64 CN cutoffs)
// This file: relations -> classes, edge tables, batching, launch geometry, rows -> outputs, the C-ABI.
#include <chrono>
#include <map>
#include <mutex>

#include "pair_common.h"

using namespace mdpair;

namespace {

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
    std::vector<int> cls;  // [n_ti][n_tj] -> class id (< n_cls; any number of classes: the device sees pass-local bytes)
    int n_cls;
    // displaced ordered rows (displace_rows below): row of (ti, tj) = disp_a[ti] + disp_b[tj]; disp_rows == 0: none found
    std::vector<int> disp_a, disp_b, disp_cls;  // disp_cls[row] -> class id (-1: no type pair lands there)
    int disp_rows = 0;
    int nbins;
    const double *edges;  // host [nbins+1]
    double rc2;
    float gscale;
    double bin_size;  // RDF: the reference's bin_size (0 for CN edge tables)
    int per_frame;
    // frame-summed RDF outputs kept on the device (mdhip_rdf_atomic_dev): full | part | overflow, accumulated by
    // derive_rdf_kernel straight from the row sums when the batch runs as one scalar-j pass; otherwise the batch
    // comes back as host class histograms like every other call and the caller adds them in
    unsigned long long *dev_out = nullptr;
    int n_rel = 0;
    const int *rel_cls = nullptr;   // host [n_rel]
    const int *rel_mult = nullptr;  // host [n_rel]
    // coordination numbers from the same sweep (mdhip_rdf_cn_atomic): one cutoff^2 per class (0: none). The bins of
    // the histogram are exact, so only the pairs of the bin that holds a class's cutoff (its split bin) need the exact
    // comparison: Hsplit counts those that are inside. Only the packed-f32 sweep does this — a batch that cannot run
    // it returns CN_UNFUSED and the caller runs a separate CN job
    int n_cn = 0;                             // != 0: on
    const double *cn_c2_cls = nullptr;        // host [n_cls]
    std::vector<uint64_t> *Hsplit = nullptr;  // out: [F|1][n_cls] pairs of the split bin with rsq < cutoff^2
    // Host-resident coordinates staged batch by batch (pair_hist_run): h_xi / h_xj are the caller's arrays, d_xi / d_xj
    // the (still empty) device buffers for all frames; the copy of batch k+1 runs on ctx->copy_stream while batch k is
    // swept. nullptr: the coordinates are on the device already.
    const double *h_xi = nullptr, *h_xj = nullptr;
};
constexpr int CN_UNFUSED = 1;  // (positive: not an error code of the ABI)
constexpr int SPLIT_BATCH = 2;  // a block may have wrapped a 32-bit LDS word: run the batch again in halves

// Row displacement for the ordered-pair rows. The table-free sweep adds to word A[ti] + B[tj] + bin — the lane holds
// A[ti] rows as its base, the j atom's record carries B[tj] rows in the addend of the bin guess — and the plain layout
// A = ti * n_tj, B = tj needs n_ti * n_tj rows: 36 rows x 401 words for the reference's own example (9 atom types, five
// relations naming five of them -> 6 type indices), which does not fit the third of LDS a block may use, and sent
// that shape to the class-row kernels (a row-table lookup per pair: 1.47x per pair). But rows only
// have to be distinct where CLASSES are: two ordered type pairs may share a row whenever they belong to the same class
// (their counts are added up at the flush anyway) — above all the many pairs that no relation names. So: find
// integers A, B with  A[i] + B[j] == A[k] + B[l]  =>  cls(i, j) == cls(k, l),  minimising max A + max B + 1. Greedy,
// type by type (A[k] and B[k] together, smallest resulting row count first), over a few orders; 15 rows instead of 36
// for the example above. The kernels are unchanged: they see A and B through the records (pack_w) and row_mul.
// Cached per class table (the search costs ~1 ms; a drop-in run repeats the same relations for every batch).
struct DispKey {
    int n_ti, n_tj;
    std::vector<int> cls;
    bool operator<(const DispKey &o) const
    {
        if (n_ti != o.n_ti) return n_ti < o.n_ti;
        if (n_tj != o.n_tj) return n_tj < o.n_tj;
        return cls < o.cls;
    }
};
struct DispVal {
    std::vector<int> a, b, row_cls;
    int rows = 0;
};

static int displace_greedy(int n_ti, int n_tj, const std::vector<int> &cls, const std::vector<int> &order, int give_up,
                           std::vector<int> &A, std::vector<int> &B, std::vector<int> &row_cls)
{
    const int n = std::max(n_ti, n_tj);
    A.assign(n_ti, -1);
    B.assign(n_tj, -1);
    row_cls.assign((size_t)2 * give_up + 2, -1);
    std::vector<int> done;  // type indices placed so far
    int max_a = 0, max_b = 0;
    std::vector<int> touched;
    for (int step = 0; step < n; ++step) {
        const int k = order[step];
        const bool has_a = k < n_ti, has_b = k < n_tj;
        int best_r = 1 << 30, best_s = 0, best_a = -1, best_b = -1;
        const int lim = give_up;
        for (int ai = 0; ai < (has_a ? lim : 1); ++ai) {
            for (int bi = 0; bi < (has_b ? lim : 1); ++bi) {
                const int ma = has_a ? std::max(max_a, ai) : max_a, mb = has_b ? std::max(max_b, bi) : max_b;
                const int r = ma + mb + 1, sc = ai + bi;
                if (r >= give_up) break;  // (bi only grows)
                if (r > best_r || (r == best_r && sc >= best_s)) continue;
                // consistent with every row written so far, and with itself?
                bool ok = true;
                touched.clear();
                auto put = [&](int row, int c) {
                    if (row_cls[row] >= 0) {
                        if (row_cls[row] != c) ok = false;
                    } else {
                        row_cls[row] = c;
                        touched.push_back(row);
                    }
                };
                if (has_a && has_b) put(ai + bi, cls[(size_t)k * n_tj + k]);
                for (size_t q = 0; q < done.size() && ok; ++q) {
                    const int u = done[q];
                    if (has_a && u < n_tj) put(ai + B[u], cls[(size_t)k * n_tj + u]);
                    if (ok && has_b && u < n_ti) put(A[u] + bi, cls[(size_t)u * n_tj + k]);
                }
                for (int row : touched) row_cls[row] = -1;
                if (!ok) continue;
                best_r = r;
                best_s = sc;
                best_a = ai;
                best_b = bi;
            }
        }
        if (best_r >= give_up) return 0;
        if (has_a) {
            A[k] = best_a;
            max_a = std::max(max_a, best_a);
        }
        if (has_b) {
            B[k] = best_b;
            max_b = std::max(max_b, best_b);
        }
        if (has_a && has_b) row_cls[best_a + best_b] = cls[(size_t)k * n_tj + k];
        for (int u : done) {
            if (has_a && u < n_tj) row_cls[best_a + B[u]] = cls[(size_t)k * n_tj + u];
            if (has_b && u < n_ti) row_cls[A[u] + best_b] = cls[(size_t)u * n_tj + k];
        }
        done.push_back(k);
    }
    const int rows = max_a + max_b + 1;
    row_cls.resize(rows);
    return rows;
}

// -> rows of the best displaced layout found (0: none with fewer rows than the plain layout, or too many types to try)
static int displace_rows(int n_ti, int n_tj, const std::vector<int> &cls, std::vector<int> &A, std::vector<int> &B,
                         std::vector<int> &row_cls)
{
    static std::mutex mu;
    static std::map<DispKey, DispVal> cache;
    const int plain = n_ti * n_tj;
    if (plain <= 4 || n_ti > 24 || n_tj > 24) return 0;
    DispKey key{n_ti, n_tj, cls};
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) {
            A = it->second.a;
            B = it->second.b;
            row_cls = it->second.row_cls;
            return it->second.rows;
        }
    }
    const int n = std::max(n_ti, n_tj);
    DispVal best;
    std::vector<int> order(n), a, b, rc;
    for (int k = 0; k < n; ++k) order[k] = k;
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    const auto t_start = std::chrono::steady_clock::now();
    for (int attempt = 0; attempt < 400; ++attempt) {
        if (attempt == 1) std::reverse(order.begin(), order.end());
        if (attempt >= 2)
            for (int k = n - 1; k > 0; --k) {  // a fixed pseudo-random shuffle: the same layout on every rank and run
                lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(order[k], order[(size_t)((lcg >> 33) % (uint64_t)(k + 1))]);
            }
        const int give_up = best.rows ? best.rows : plain;  // only strictly better layouts
        const int rows = displace_greedy(n_ti, n_tj, cls, order, give_up, a, b, rc);
        if (rows > 0 && (best.rows == 0 || rows < best.rows)) {
            best.rows = rows;
            best.a = a;
            best.b = b;
            best.row_cls = rc;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 0.02) break;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (cache.size() > 256) cache.clear();
        cache[key] = best;
    }
    A = best.a;
    B = best.b;
    row_cls = best.row_cls;
    return best.rows;
}

// Runs the kernel over one batch of frames (in several passes when the class rows do not fit LDS) and
// returns the class histograms on the host: H [F|1][n_cls][nbins], overflow count.
// `defer` (an asynchronous call's scope, or nullptr): when the batch runs as ONE scalar-j pass, its host half — flag
// check, timer read-out, folding the row sums into H — is left as a completion step of that call instead of being done
// behind a stream sync here; H, *overflow, p.Hsplit and *redo must then outlive the call (they are parts of the entry
// point's heap state). A launch that raised the overflow guard sets *redo there (nothing else is touched) instead of
// returning SPLIT_BATCH. Everything else (class passes, the dense kernels) completes inside, as before.
int pair_hist_run_batch(mdhip_ctx *ctx, const PairProblem &p, std::vector<uint64_t> &H, uint64_t *overflow,
                        CallScope *defer = nullptr, bool *redo = nullptr)
{
    const int64_t F = p.n_frames;
    const int nTi = (int)((p.ni + TILE - 1) / TILE);
    const int nTj = (int)((p.nj + TILE - 1) / TILE);
    const size_t out_frames = p.per_frame ? (size_t)F : 1;
    H.assign(out_frames * p.n_cls * p.nbins, 0);
    if (p.Hsplit) p.Hsplit->assign(out_frames * p.n_cls, 0);
    *overflow = 0;
    if (F == 0 || p.ni == 0 || p.nj == 0) return MDHIP_OK;

    // kernel variant: 0 = reference-shaped loops with an edge-table lookup per pair (always used for CN
    // edge tables, gscale == 0); 1 = fast kernel (table-free binning with an exact guard band)
    const bool mode_cn = !(p.gscale > 0.f);  // CN edge table: a few sorted cutoffs^2, bins found by counting
    // The table-free bin guess of the fast kernels indexes a row of nbins + 1 words: that is enough exactly when
    // nbins = int(r_cut / bin_size) as the reference computes it (rdf_cn.py:169); a caller that passes fewer bins
    // gets the edge-table kernel, which clamps.
    const bool bins_ok = mode_cn || !(p.bin_size > 0.0) || std::sqrt(p.rc2) / p.bin_size < (double)p.nbins + 1.0 - 1e-9;
    const bool fast = ctx->opt_rdf_variant == 1 && bins_ok && (mode_cn ? p.nbins <= 64 : p.nbins <= 100000);

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

    // Ordered-pair rows (MODE 2 / 3 of the scalar-j kernel): one LDS row per (ti, tj) addressed without a table —
    // the row offset rides in the addend of the bin guess. Needs all n_ti^2 rows in LDS and all classes in one pass:
    // at >= 4 blocks of 4 waves per CU for the all-f64 sweep, at 3 blocks of 8 waves (6 waves per SIMD) for the
    // packed-f32 sweep, whose blocks share one histogram among 8 waves.
    // Row layout of the ordered modes: plain (row = ti * n_tj + tj) unless that does not fit the packed sweep's third of
    // LDS and the displaced layout of displace_rows (row = A[ti] + B[tj], classes never mixed in a row) does.
    // (class rows of the packed sweep: as many classes per pass as fit a third of LDS)
    int pk_cls_fit = 0;
    for (int nc = std::min(p.n_cls, 250); nc >= 1; --nc)
        if (lds_bytes_sj_pk_rows(p.nbins, nc, p.n_ti, p.n_tj, p.n_cn) <= lds_cap / 3 - 512) {
            pk_cls_fit = nc;
            break;
        }
    int ord_rows = p.n_ti * p.n_tj, ord_maxb = p.n_tj - 1;
    bool displaced = false, big = false;
    {
        const size_t third = lds_cap / 3 - 512, whole = lds_cap - 1024;
        const bool have_disp = p.disp_rows > 0 && p.disp_rows < ord_rows && ctx->opt_rdf_disp != 0;
        auto use_disp = [&]() {
            displaced = true;
            ord_rows = p.disp_rows;
            ord_maxb = *std::max_element(p.disp_b.begin(), p.disp_b.end());
        };
        if (lds_bytes_sj_pk(p.nbins, ord_rows, p.n_cn) <= third) {
            if (have_disp && ctx->opt_rdf_disp == 2) use_disp();  // (A/B: whenever it has fewer rows)
        } else if (have_disp && lds_bytes_sj_pk(p.nbins, p.disp_rows, p.n_cn) <= third) {
            use_disp();
        } else if (p.n_cn == 0 && ctx->opt_rdf_big != 0 && lds_cap >= 160 * 1024 && pk_cls_fit < p.n_cls) {
            // neither fits a third of LDS, and the class rows would need several passes (all classes in ONE pass of class rows
            // measured faster than this: 3.12 against 3.37 ms at C1's shape): ONE 16-wave block per CU with the whole LDS for
            // its histogram (BIG, pair_sj.hip) — every pair of nine types named is 81 rows, 130 KB — on whichever layout has
            // fewer rows
            const int rows_small = have_disp ? p.disp_rows : ord_rows;
            if (lds_bytes_sj_pk(p.nbins, rows_small, 0, true) <= whole) {
                big = true;
                if (have_disp) use_disp();
            }
        }
    }
    const std::vector<int> &row_cls = displaced ? p.disp_cls : p.cls;  // ordered row -> class
    const size_t ord_b = lds_bytes_sj_ordered(p.nbins, ord_rows);
    const bool ord_base = cull && ctx->opt_rdf_sj != 0 && !mode_cn && ctx->opt_rdf_rows != 0 &&
                          p.n_cls <= 250 && (double)(ord_maxb + 1) * (p.nbins + 1) < 65536.0;
    bool ordered = ord_base && ord_b <= lds_cap / 4;
    // Packed-f32 classification (MODE 3-6 of the scalar-j kernel, header in pair_sj.hip) when the error band is
    // narrow: with the ordered rows when they fit a third of LDS (3 blocks of 8 waves per CU), else with class rows
    // and their row table (any number of types). The cutoff on a bin edge lets the band of that edge decide in/out
    // of the cutoff; a cutoff inside the last bin has its own band tested per pair (cut_guard).
    bool pk = false, pk_rows = false;
    float s_cap = 0.f, rc2hi = 0.f, near_pk_f = 0.f, cut_lo = 0.f;
    bool cut_guard = false;
    double pk_err = 0.0;  // error bound of the f32 distance, in bins
    int rel_block = 0;  // != 0: the packed sweep's f32 records are wanted (relative to their tile's centre)
    if (cull && ctx->opt_rdf_sj != 0 && !mode_cn && p.n_cls <= 250 && ctx->opt_rdf_pk != 0 && p.bin_size > 0.0) {
        const bool fits_ordered = ord_base && (big || lds_bytes_sj_pk(p.nbins, ord_rows, p.n_cn) <= lds_cap / 3 - 512);
        // class rows: as many classes per pass as fit a third of LDS. Round 6: when they do not all fit (every pair of nine
        // types named: 45 classes x 401 words = 72 KB) the packed sweep runs in SEVERAL passes over the pairs instead of
        // leaving the call to the all-f64 class-row kernel — C1's atoms with all 45 relations: 14.5 -> 6.3 ms per 200 frames
        // (`bench.py --shape C1full`); coordination numbers from the same sweep need one pass (else: two sweeps, as before)
        // (at least 8 classes per pass: with rows so long that fewer fit, the f64 kernel's half-of-LDS passes are as few)
        const bool fits_rows = pk_cls_fit >= p.n_cls || (pk_cls_fit >= 8 && p.n_cn == 0 && ctx->opt_rdf_pk_passes != 0);
        const double r_cut = std::sqrt(p.rc2);
        const double cpos = r_cut / p.bin_size, K = std::floor(cpos + 0.5);
        double l_max = 0.0, v_max = 0.0;
        for (int64_t f = 0; f < F; ++f) {
            const double *b = p.h_box + 3 * f;
            l_max = std::max(l_max, std::max(b[0], std::max(b[1], b[2])));
            v_max = std::max(v_max, b[0] * b[1] * b[2]);
        }
        // tile edge of the sparser of the two sets (atoms x sites: the sites)
        const double edge = std::cbrt((double)TILE * v_max / (double)std::min(p.ni, p.nj));
        const double cap = r_cut + 3.5 * edge;
        // (the guess carries tj * row_len with ordered rows, nothing with class rows)
        const double err = pk_error_bound(r_cut, p.bin_size, p.nbins, fits_ordered ? ord_maxb + 1 : 1, cap, l_max);
        const double u = std::ldexp(1.0, -24);
        const double near_pk = 2.0 * err + 4.5 * u * (p.nbins + 1) + 2.0e-5;
        const bool on_edge = std::fabs(cpos - K) <= 1e-6 && (K == (double)p.nbins || K == (double)p.nbins + 1.0);
        if ((fits_ordered || fits_rows) && (on_edge || std::floor(cpos) == (double)p.nbins) && near_pk <= 0.02 &&
            std::isfinite(l_max)) {
            pk = true;
            pk_err = err;
            pk_rows = !fits_ordered;
            ordered = fits_ordered;
            cut_guard = !on_edge;
            // sqrt(rsq32) < cut_lo  =>  sqrt(rsq) < cut_lo + err * bin_size < r_cut: inside the cutoff for certain
            cut_lo = std::nextafterf((float)(r_cut - 1.1 * err * p.bin_size), 0.f);
            // the f32 records are relative to the centre of their whole tile (64-atom blocks bought a little f32
            // precision for 4x the per-block work: measured slower in round 2, retired in round 4)
            rel_block = TILE;
            near_pk_f = (float)near_pk;
            s_cap = (float)cap;
            // every pair with rsq < r_cut^2 has sqrt(rsq32) <= r_cut + err * bin_size
            const double r_hi = r_cut + err * p.bin_size;
            rc2hi = std::nextafterf((float)(r_hi * r_hi * (1.0 + 2.0 * u)), std::numeric_limits<float>::infinity());
            if (pk_rows) {  // all classes in one pass when they fit, else balanced passes of at most pk_cls_fit classes
                const int np = (p.n_cls + pk_cls_fit - 1) / pk_cls_fit;
                cls_per_pass = (p.n_cls + np - 1) / np;
            }
        }
    }
    big = big && pk && ordered && ctx->opt_rdf_pk != 2;  // (only the packed ordered sweep has the 16-wave instance)
    if (p.n_cn > 0 && (!pk || ctx->opt_rdf_pk == 2)) return CN_UNFUSED;
    float near_ord = 0.f;
    if (ordered) {
        cls_per_pass = p.n_cls;
        // |error| of the f32 guess g = fma(sqrt((float)rsq), 1/ddr, near + tj*row_len): relative 2^-25 (conversion,
        // halved by the root) + 2^-23 (v_sqrt_f32, 1 ulp) + 2^-24 (rounded 1/ddr) = 2.1e-7 of the bin number, plus
        // half an ulp of the largest value each for the rounding of the addend and of the fma. near = 2 x that.
        const double maxg = (double)(ord_maxb + 1) * (p.nbins + 1) + 1.0;  // the addend carries B[tj] * row_len only
        const double ulp = std::ldexp(1.0, (int)std::floor(std::log2(maxg)) - 23);
        near_ord = (float)(2.0 * ((double)p.nbins * 2.1e-7 + ulp) + 1.0e-5);
        near_ord = std::max(near_ord, near_pk_f);  // one band for the f32 guess of either sweep
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
    // scalar-j kernels: items are (frame, tile, wave, slice); 4 slices measured best at C2 and C3, for the persistent
    // grid and for per-frame output alike (with one slice a 100k-atom frame has only two items per resident wave)
    if (cull && ctx->opt_rdf_sj != 0 && ctx->opt_rdf_jsplit <= 0) {
        // ... when the lists are long. A short reach (coordination cutoffs: a handful of neighbour tiles per tile) leaves
        // a slice one tile or none, and every item pays its set-up (counter, boxes, context) for it: round 4 measured
        // 4.72 -> 2.93 ms per 64 C3 frames for CN alone with ONE slice (tools/ab_pair.py rdf_jsplit=4,2,1 C3 cn).
        // Expected list length: the share of tile pairs within reach (as for the culling decision above) x tiles / 2.
        const double V = p.h_box[0] * p.h_box[1] * p.h_box[2];
        const double edge = 0.5 * (std::cbrt((double)TILE * V / (double)p.ni) + std::cbrt((double)TILE * V / (double)p.nj));
        const double reach = std::sqrt(p.rc2) + 0.8 * edge;
        const double share = std::min(1.0, 4.18879 * reach * reach * reach / V);
        const double list_est = share * (double)nTj * (p.tri ? 0.5 : 1.0);
        jsplit = std::min(list_est >= 12.0 ? 4 : list_est >= 6.0 ? 2 : 1, max_list);
    }
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
    // CN tables of the scalar-j rows (one pass, all rows): word index of every row's split bin | cutoff^2 per row
    const int cn_rows = p.n_cn ? (ordered ? ord_rows : p.n_cls + 1) : 0;
    const size_t cn_fw = ((size_t)cn_rows + 1) & ~size_t(1);
    const size_t cn_b = p.n_cn ? (cn_fw + 2 * (size_t)cn_rows) * 4 : 0;
    float cn_reach = 0.f;
    const size_t cls_b = ((size_t)p.n_ti * p.n_tj + 63) & ~size_t(63);
    // edges and the class table of every pass: one pinned staging buffer, one H2D copy
    // (displaced rows: A | B as ints behind the CN tables, for pack_w of the sort pre-pass)
    const size_t disp_off = (edges_b + (size_t)n_pass * cls_b + cn_b + 7) & ~size_t(7);
    const size_t disp_b = ordered && displaced ? ((size_t)p.n_ti + p.n_tj) * 4 : 0;
    const size_t tab_b = disp_off + disp_b + 8;
    // The common case of the scalar-j sweep — one class pass, frame-summed rows — needs two more small things that a
    // C2 step paid a copy / a fill of their own for (round 5: ~12 us each with the gaps around them): the row map of
    // derive_rdf_kernel (results left on the device) rides behind the tables in the same copy, and the row sums sit
    // behind the flag words so that ONE fill empties both.
    const bool sj_path = cull && ctx->opt_rdf_sj != 0;
    const bool one_sum = sj_path && n_pass == 1 && !p.per_frame;
    const int sj_rows1 = ordered ? ord_rows : p.n_cls + 1;
    const size_t rows1_b = one_sum ? (size_t)sj_rows1 * (size_t)(p.nbins + 1 + (p.n_cn ? 1 : 0)) * 8 : 0;
    const bool map_rides = one_sum && p.dev_out != nullptr;
    const size_t tab_al = (tab_b + 15) & ~size_t(15);
    const size_t map_b = map_rides ? ((size_t)sj_rows1 + 2 * (size_t)p.n_rel) * 4 : 0;
    MD_WS(d_tab, unsigned char, WS_TABLES, tab_al + map_b);
    MD_PIN(h_tab, unsigned char, tab_al + map_b + 32);
    {
        double *e = reinterpret_cast<double *>(h_tab);
        std::copy(p.edges, p.edges + p.nbins + 1, e);
        e[p.nbins + 1] = std::numeric_limits<double>::infinity();
        if (p.n_cn) {
            int *fl = reinterpret_cast<int *>(h_tab + edges_b + (size_t)n_pass * cls_b);
            double *c2r = reinterpret_cast<double *>(fl + cn_fw);
            std::fill(fl, fl + cn_fw, -1);
            double reach = 0.0;
            for (int r = 0; r < cn_rows; ++r) {
                const int cl = ordered ? row_cls[r] : (r < p.n_cls ? r : -1);
                const double c2 = cl >= 0 ? p.cn_c2_cls[cl] : 0.0;
                c2r[r] = c2;
                if (!(c2 > 0.0)) continue;
                // split bin kc: edges[kc] <= c2 < edges[kc + 1]; none when the cutoff sits exactly on an edge
                const int kc = (int)(std::upper_bound(p.edges, p.edges + p.nbins + 1, c2) - p.edges) - 1;
                if (p.edges[kc] == c2) continue;
                fl[r] = r * (p.nbins + 1) + kc;
                const double top = kc + 1 <= p.nbins ? std::sqrt(p.edges[kc + 1]) : std::sqrt(p.rc2);
                reach = std::max(reach, std::min(top, std::sqrt(p.rc2)));
            }
            cn_reach = (float)((reach + 1e-3) * 1.00001);
        }
        for (int pass = 0; pass < n_pass; ++pass) {
            const int c0 = pass * cls_per_pass;
            const int nc = (p.n_cls - c0) < cls_per_pass ? (p.n_cls - c0) : cls_per_pass;
            unsigned char *t = h_tab + edges_b + (size_t)pass * cls_b;
            for (size_t k = 0; k < (size_t)p.n_ti * p.n_tj; ++k) {
                const int c = p.cls[k];  // (pass-local ids are < 250: they fit the byte, 0xFF = other pass)
                t[k] = (c >= c0 && c < c0 + nc) ? (unsigned char)(c - c0) : 0xFF;
            }
        }
    }
    RowDisp disp_i, disp_j;
    if (disp_b) {
        int *h_d = reinterpret_cast<int *>(h_tab + disp_off);
        std::copy(p.disp_a.begin(), p.disp_a.end(), h_d);
        std::copy(p.disp_b.begin(), p.disp_b.end(), h_d + p.n_ti);
        const int *d_d = reinterpret_cast<const int *>(d_tab + disp_off);
        // atom-atom: one sorted set plays both roles (low word A, addend B); atoms x sites: the i set's addend and the j
        // set's low word are never read
        disp_i.lo = d_d;
        disp_i.hi = p.tri ? d_d + p.n_ti : d_d;
        disp_j.lo = disp_j.hi = d_d + p.n_ti;
    }
    if (map_rides) {
        int *h_map = reinterpret_cast<int *>(h_tab + tab_al);
        for (int r = 0; r < sj_rows1; ++r) h_map[r] = ordered ? row_cls[r] : (r < p.n_cls ? r : -1);
        for (int kl = 0; kl < p.n_rel; ++kl) {
            h_map[sj_rows1 + kl] = p.rel_cls[kl];
            h_map[sj_rows1 + p.n_rel + kl] = p.rel_mult[kl];
        }
    }
    {
        const int rcc = mdhip_copy_small(ctx, d_tab, h_tab, tab_al + map_b, hipMemcpyHostToDevice);
        if (rcc) return rcc;
    }
    MD_WS(d_misc, unsigned long long, WS_MISC, 64 + rows1_b);
    MD_HIP(hipMemsetAsync(d_misc, 0, 64 + rows1_b, ctx->stream));

    // ---- culled path: Morton sort, tile boxes, neighbour-tile lists (once, shared by all class passes) ----
    const double *k_xi = p.d_xi, *k_xj = p.d_xj;
    const int *k_ti = p.d_ti, *k_tj = p.d_tj;
    long long k_ti_fs = p.ti_fs, k_tj_fs = p.tj_fs;
    const unsigned short *d_list = nullptr;
    const int *d_list_cnt = nullptr;
    const float4 *d_gsph = nullptr, *d_wsph = nullptr, *d_gsph4 = nullptr;
    const double4 *d_aos = nullptr;
    // what the host half of the passes accumulates (shared with the completion step of a deferred batch)
    struct BatchAcc {
        double total_ms = 0.0, prep_ms = 0.0;
        int launches = 0;
        unsigned long long ov = 0;
        bool prep_timed = false;
        KernelTimer prep;
        explicit BatchAcc(const KernelTimer &t) : prep(t) {}
    };
    std::shared_ptr<BatchAcc> acc;
    const double4 *d_aos_j = nullptr;
    const float *d_rel = nullptr;
    const double *d_cen = nullptr;
    if (cull) {
        const long long N = p.ni;
        const bool want_soa = ctx->opt_rdf_sj == 0;  // only the LDS-tile kernel (atom-atom) reads the SoA copy
        MD_WS(d_l, unsigned short, WS_LIST, (size_t)F * nTi * (p.tri ? nTi : nTj) * 2);
        MD_WS(d_lc, int, WS_LISTCNT, (size_t)F * nTi * 4);
        KernelTimer ptimer(ctx, 1, true);  // second event pair: collected with the pair kernel's
        SortedSet si, sj_set;
        const int slot_i[5] = {WS_SORT_AOS, WS_BBOX, WS_GSPH, WS_WSPH, WS_GSPH4};
        // (the bin-guess addend near + type * row_len and the tile-relative f32 records belong to the j set)
        int rc = cull_prepare_set(ctx, F, p.d_xi, p.d_ti, (long long)p.ti_fs, p.d_box, N, nTi, p.n_ti,
                                  ordered ? near_ord : 0.f, ordered ? p.nbins + 1 : 0, disp_i, want_soa,
                                  pk && p.tri ? rel_block : 0, pk_rows ? 1 : 0, pk ? 1 : 0, slot_i, si);
        if (rc) return rc;
        if (p.tri) {
            sj_set = si;
        } else {
            const int slot_j[5] = {WS_SORT_AOS_J, WS_BBOX_J, WS_GSPH_J, WS_WSPH_J, WS_GSPH4_J};
            rc = cull_prepare_set(ctx, F, p.d_xj, p.d_tj, (long long)p.tj_fs, p.d_box, p.nj, nTj, p.n_ti,
                                  ordered ? near_ord : 0.f, ordered ? p.nbins + 1 : 0, disp_j, false, pk ? rel_block : 0,
                                  pk_rows ? 1 : 0, pk ? 1 : 0, slot_j, sj_set);
            if (rc) return rc;
        }
        launch_cull_lists(ctx->stream, p.tri, F, si.bbox, sj_set.bbox, nTi, nTj, p.d_box,
                          p.rc2 * (1.0 + 1e-9) + 1e-9, d_l, d_lc);
        ptimer.stop();
        MD_HIP(hipGetLastError());
        acc = std::make_shared<BatchAcc>(ptimer);
        acc->prep_timed = true;
        d_gsph = sj_set.gs;    // 8-atom boxes of the j set (LDS-tile kernel)
        d_gsph4 = sj_set.gs4;  // 4-atom boxes of the j set (scalar-j kernel)
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
        d_rel = sj_set.rel;
        d_cen = sj_set.cen;
    }

    // (total_ms: the pair kernel alone; the culling pre-pass is reported separately)
    if (!acc) acc = std::make_shared<BatchAcc>(KernelTimer(ctx, 1, true));
    const int n_cls_all = p.n_cls, nbins_all = p.nbins;
    // the end of the batch: what the passes found goes to the caller and to the context's registers
    auto finalize = [ctx, acc, overflow, n_pass]() {
        // with several passes every in-cutoff overflow pair is seen once per pass
        *overflow = acc->ov / (uint64_t)n_pass;
        ctx->last_ms = acc->total_ms;
        ctx->last_launches = acc->launches;
        ctx->last_aux_ms = acc->prep_ms;
        return MDHIP_OK;
    };
    bool deferred = false;
    for (int pass = 0; pass < n_pass; ++pass) {
        const int c0 = pass * cls_per_pass;
        const int nc = (p.n_cls - c0) < cls_per_pass ? (p.n_cls - c0) : cls_per_pass;
        const size_t words = (size_t)nc * p.nbins;
        const size_t acc_frames = p.per_frame ? (size_t)F : (size_t)slots;
        MD_WS(d_hist, unsigned long long, WS_HIST, (acc_frames + 1) * words * 8);
        // (the scalar-j kernels keep their histograms in LDS and store them to their own slices: nothing of theirs is
        // added to d_hist, so it need not be emptied for them — one fill less per call)
        if (!(cull && ctx->opt_rdf_sj != 0)) MD_HIP(hipMemsetAsync(d_hist, 0, acc_frames * words * 8, ctx->stream));

        PairArgs a;
        a.xi = k_xi;
        a.xj = k_xj;
        a.ti = k_ti;
        a.tj = k_tj;
        a.list = d_list;
        a.list_cnt = d_list_cnt;
        a.gsph = d_gsph;
        a.gsph4 = d_gsph4;
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
        a.row_mul = ordered && displaced ? 1 : p.n_tj;
        a.n_rows_ord = ord_rows;
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

        a.near = pk_rows ? near_pk_f : near_ord;
        a.rel = d_rel;
        a.cen = d_cen;
        a.s_cap = s_cap;
        a.rc2hi = rc2hi;
        a.cut_lo = cut_lo;
        a.n_cn = p.n_cn;
        a.cn_tab = reinterpret_cast<const unsigned *>(d_tab + edges_b + (size_t)n_pass * cls_b);
        a.cn_reach = cn_reach;
        const bool sj = cull && ctx->opt_rdf_sj != 0;  // wave-independent sweep with scalar loads of the j atoms
        const bool persist = sj && !p.per_frame && ctx->opt_rdf_sj != 2;  // resident grid + per-XCD work counters
        a.work = reinterpret_cast<unsigned *>(d_misc + 4);
        const size_t lds = pk_rows ? lds_bytes_sj_pk_rows(p.nbins, nc, p.n_ti, p.n_tj, p.n_cn)
                           : pk    ? lds_bytes_sj_pk(p.nbins, ord_rows, p.n_cn, big)
                           : ordered ? ord_b
                           : sj    ? lds_bytes_sj(p.nbins, nc, p.n_ti, p.n_tj, mode_cn)
                           : fast ? lds_bytes_fast(p.nbins, nc, p.n_ti, p.n_tj)
                                  : lds_bytes(p.nbins, nc, p.n_ti, p.n_tj);
        if (sj) {
            a.guard_off = (unsigned)((lds + 15) & ~size_t(15));
            // one neighbour tile adds at most 64 x 256 to any one word of a block: 2^32 / 2^14 tiles, with margin
            a.guard_tiles = ctx->opt_rdf_guard > 0 ? (unsigned)ctx->opt_rdf_guard : 250000u;
        }
        const size_t lds_launch = sj ? ((lds + 15) & ~size_t(15)) + 16 : lds;
        const char *kname = "";
        const int sj_mode = pk && ctx->opt_rdf_pk != 2 ? (pk_rows ? 5 : 3) + (cut_guard ? 1 : 0) : ordered ? 2 : mode_cn ? 1 : 0;
        const bool big_launch = big && sj_mode >= 3 && sj_mode <= 4;
        const int bs = sj ? sj_block_threads(sj_mode, big_launch) : TILE;  // threads per block
        const int wpb = bs / 64;                                // independent waves per block (scalar-j kernels)
        PairKernel kern = sj ? sj_kernel(sj_mode, persist, p.n_cn > 0, big_launch, &kname)
                             : dense_kernel(fast, p.tri, mode_cn, fast && cull, &kname);
        ctx->last_kernel = kname;
        if (lds_launch > 65536)
            MD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
        long long launch_grid = grid;
        if (sj) {
            int per_cu = 0;
            MD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), bs,
                                                                lds_launch));
            if (per_cu < 1) per_cu = 1;
            const long long capacity = (long long)per_cu * ctx->cu_count;
            // blocks' worth of wave items per frame (a frame has nTi * 4 * jsplit wave items)
            const long long block_items = ((long long)nTi * 4 * jsplit + wpb - 1) / wpb;
            a.fpb = 1;
            if (persist) {
                launch_grid = std::min(capacity, F * block_items + 8);
                launch_grid = (launch_grid + 7) / 8 * 8;
                // (the work counters: the first pass finds them empty, d_misc was zeroed whole at the top of the batch)
                if (pass > 0) MD_HIP(hipMemsetAsync(d_misc + 4, 0, 32, ctx->stream));
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
        const int sj_rows = ordered ? ord_rows : nc + 1;
        const int cn_len = p.n_cn ? 1 : 0;  // one split counter per row, behind all histogram rows
        const int sj_words = sj_rows * (p.nbins + 1 + cn_len);
        unsigned long long *d_rows = nullptr;
        if (sj) {
            MD_WS(d_sl, unsigned, WS_SLICES, (size_t)launch_grid * sj_words * 4);
            if (one_sum && rows1_b == (size_t)sj_words * 8) {
                d_rows = d_misc + 8;  // (emptied with the flag words at the top of the batch)
            } else {
                d_rows = (unsigned long long *)mdhip_ws(ctx, WS_ROWS, out_frames * (size_t)sj_words * 8);
                if (!d_rows) return MDHIP_ENOMEM;
                if (!p.per_frame) MD_HIP(hipMemsetAsync(d_rows, 0, (size_t)sj_words * 8, ctx->stream));
            }
            a.slices = d_sl;
        }
        KernelTimer timer(ctx);
        hipLaunchKernelGGL(kern, dim3((unsigned)launch_grid), dim3(bs), lds_launch, ctx->stream, a);
        if (sj) {
            const unsigned gy = p.per_frame ? (unsigned)F : (unsigned)std::min<long long>(64, launch_grid);
            launch_merge_slices(ctx->stream, a.slices, sj_words, launch_grid, p.per_frame, a.blocks_per_frame, gy,
                                d_rows);
        }
        timer.stop();
        MD_HIP(hipGetLastError());

        const bool can_defer = defer != nullptr && sj && n_pass == 1;
        // the launch's flags (pinned): [0] deferred pairs lost, [1] work-loop assertion, [2] overflow guard
        auto check_flags = [ctx](const uint64_t *hlost) {
            if (hlost[2]) return SPLIT_BATCH;
            if (hlost[0] || hlost[1])
                return mdhip_fail(ctx, MDHIP_EHIP, "pair_hist: internal check failed (%llu deferred pairs lost, work loop %llu)",
                                  (unsigned long long)hlost[0], (unsigned long long)hlost[1]);
            return MDHIP_OK;
        };
        auto collect_times = [acc, timer]() {
            acc->total_ms += timer.collect();
            ++acc->launches;
            if (acc->prep_timed) {  // the pre-pass ran ahead of the first pass on the same stream: its events are complete
                acc->prep_ms = acc->prep.collect();
                acc->prep_timed = false;
            }
        };
        if (sj && p.dev_out && n_pass == 1 && !p.per_frame) {
            // outputs stay on the device: rows -> full | part | overflow by derive_rdf_kernel (added to dev_out)
            // (the row map came with the tables, in the copy at the top of the batch: map_rides)
            if (!map_rides || sj_rows != sj_rows1)
                return mdhip_fail(ctx, MDHIP_EHIP, "pair_hist: internal: the row map of the device-resident path is missing");
            const int *d_map = reinterpret_cast<const int *>(d_tab + tab_al);
            launch_derive_rdf(ctx->stream, d_rows, sj_rows, p.nbins, d_map, p.n_rel, d_map + sj_rows,
                              d_map + sj_rows + p.n_rel, d_misc + 3, p.dev_out);
            MD_HIP(hipGetLastError());
            uint64_t *hlost = reinterpret_cast<uint64_t *>(h_tab + tab_al + ((map_b + 7) & ~size_t(7)));
            {
                const int rcc = mdhip_copy_small(ctx, hlost, d_misc + 1, 24, hipMemcpyDeviceToHost);
                if (rcc) return rcc;
            }
            auto fin = [check_flags, collect_times, hlost]() {
                const int rcf = check_flags(hlost);
                if (rcf) return rcf;
                collect_times();
                return (int)MDHIP_OK;
            };
            if (can_defer) {
                defer->defer([fin, finalize, redo]() {
                    const int rcf = fin();
                    if (rcf == SPLIT_BATCH) {
                        *redo = true;
                        return (int)MDHIP_OK;
                    }
                    return rcf ? rcf : finalize();
                });
                deferred = true;
                continue;
            }
            MD_HIP(mdhip_stream_wait(ctx));
            const int rcf = fin();
            if (rcf) return rcf;
            continue;
        }
        if (sj) {
            // D2H of the row sums (pinned staging), then rows -> classes and the overflow words on the host
            MD_PIN(hall, uint64_t, (out_frames * (size_t)sj_words + 16) * 8);
            uint64_t *hrows = hall + 8;
            uint64_t *hlost = hrows + out_frames * (size_t)sj_words;  // [0] queue overflow, [1] work-loop assertion
            if (d_rows == d_misc + 8 && out_frames == 1) {
                // the row sums sit behind the flag words (one_sum): flags and rows leave in ONE copy
                hlost = hall + 1;
                const int rcc = mdhip_copy_small(ctx, hall, d_misc, (8 + (size_t)sj_words) * 8, hipMemcpyDeviceToHost);
                if (rcc) return rcc;
            } else {
                int rcc = mdhip_copy_small(ctx, hrows, d_rows, out_frames * (size_t)sj_words * 8, hipMemcpyDeviceToHost);
                if (!rcc) rcc = mdhip_copy_small(ctx, hlost, d_misc + 1, 24, hipMemcpyDeviceToHost);
                if (rcc) return rcc;
            }
            std::vector<uint64_t> *Hp = &H, *Hsplit = p.Hsplit;
            auto fin = [check_flags, collect_times, acc, hrows, hlost, Hp, Hsplit, out_frames, sj_words, sj_rows, cn_len,
                        ordered, nc, c0, n_cls_all, nbins_all, cls = ordered ? row_cls : std::vector<int>()]() {
                const int rcf = check_flags(hlost);
                if (rcf) return rcf;
                collect_times();
                const int row_len = nbins_all + 1;
                for (size_t fr = 0; fr < out_frames && cn_len; ++fr)
                    for (int r = 0; r < sj_rows; ++r) {
                        const uint64_t *src = hrows + fr * (size_t)sj_words + (size_t)sj_rows * row_len + (size_t)r;
                        const int cl = ordered ? cls[r] : (r < nc ? c0 + r : -1);
                        if (cl >= 0) (*Hsplit)[fr * n_cls_all + cl] += src[0];
                    }
                for (size_t fr = 0; fr < out_frames; ++fr)
                    for (int r = 0; r < sj_rows; ++r) {
                        const uint64_t *src = hrows + fr * (size_t)sj_words + (size_t)r * row_len;
                        acc->ov += src[nbins_all];
                        // ordered rows (ti, tj) -> class of the unordered pair; class rows of this pass -> c0 + r, the
                        // extra row holds the pairs whose class belongs to another pass
                        const int cl = ordered ? (int)cls[r] : (r < nc ? c0 + r : -1);
                        if (cl < 0) continue;
                        uint64_t *dst = &(*Hp)[(fr * n_cls_all + cl) * nbins_all];
                        for (int k = 0; k < nbins_all; ++k) dst[k] += src[k];
                    }
                return (int)MDHIP_OK;
            };
            if (can_defer) {
                defer->defer([fin, finalize, redo]() {
                    const int rcf = fin();
                    if (rcf == SPLIT_BATCH) {
                        *redo = true;
                        return (int)MDHIP_OK;
                    }
                    return rcf ? rcf : finalize();
                });
                deferred = true;
                continue;
            }
            MD_HIP(mdhip_stream_wait(ctx));
            const int rcf = fin();
            if (rcf) return rcf;
            continue;
        }

        unsigned long long *d_final = d_hist;
        if (!p.per_frame && slots > 1) {
            d_final = d_hist + (size_t)slots * words;
            launch_reduce_slots(ctx->stream, d_hist, d_final, (int)words, slots);
            MD_HIP(hipGetLastError());
        }
        // D2H (pinned staging) into the right class rows; the overflow word rides along with the last pass
        MD_PIN(tmp, uint64_t, (out_frames * words + 1) * 8);
        MD_HIP(hipMemcpyAsync(tmp, d_final, out_frames * words * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (pass == n_pass - 1)
            MD_HIP(hipMemcpyAsync(tmp + out_frames * words, d_misc, 8, hipMemcpyDeviceToHost, ctx->stream));
        MD_HIP(mdhip_stream_wait(ctx));
        if (pass == n_pass - 1) acc->ov = tmp[out_frames * words];  // (d_misc[0] accumulates over the passes)
        collect_times();
        for (size_t fr = 0; fr < out_frames; ++fr)
            memcpy(&H[(fr * p.n_cls + c0) * p.nbins], &tmp[fr * words], words * 8);
    }
    if (deferred) return MDHIP_OK;
    return finalize();
}

// Copies the frames [f0, f0 + n) of the host-resident inputs of `p` to their device buffers on the copy stream and
// records `ev` behind them.
static int stage_batch_async(mdhip_ctx *ctx, const PairProblem &p, int64_t f0, int64_t n, hipEvent_t ev)
{
    // (pageable sources through the context's page-locked ring: mdhip_h2d_any — the halves alternate with the batches)
    const int half = ev == ctx->copy_ev[1] ? 1 : 0;
    if (p.h_xi) {
        const int rc = mdhip_h2d_any(ctx, const_cast<double *>(p.d_xi) + (size_t)f0 * 3 * p.ni, p.h_xi + (size_t)f0 * 3 * p.ni,
                                     (size_t)n * 3 * p.ni * 8, ctx->copy_stream, half);
        if (rc) return rc;
    }
    if (p.h_xj) {
        const int rc = mdhip_h2d_any(ctx, const_cast<double *>(p.d_xj) + (size_t)f0 * 3 * p.nj, p.h_xj + (size_t)f0 * 3 * p.nj,
                                     (size_t)n * 3 * p.nj * 8, ctx->copy_stream, half);
        if (rc) return rc;
    }
    MD_HIP(hipEventRecord(ev, ctx->copy_stream));
    return MDHIP_OK;
}

static int pair_hist_run_range(mdhip_ctx *ctx, const PairProblem &p, int64_t f0, int64_t n, std::vector<uint64_t> &H,
                               uint64_t *overflow, std::vector<uint64_t> *Hsplit, CallScope *defer = nullptr,
                               bool *redo = nullptr);

// Splits the frames into batches so that the culled path's workspace (sorted copy, keys, cell counts, boxes,
// neighbour-tile lists) stays within ~2 GiB and a launch's grid.y within 65535, and merges the batches.
// Host-resident coordinates (p.h_xi): the batches are also the unit of the host-to-device staging — the copy of batch
// k + 1 is issued on the copy stream before batch k is swept, so that it runs under that sweep (page-locked sources:
// a DMA the host does not wait for; pageable sources: the runtime stages them synchronously, so the copy simply comes
// first, as without this scheme). A shorter first batch lets the sweep start early.
// `defer` / `redo`: see pair_hist_run_batch — used when the frames are ONE batch whose coordinates are on the device.
int pair_hist_run(mdhip_ctx *ctx, const PairProblem &p, std::vector<uint64_t> &H, uint64_t *overflow,
                  CallScope *defer = nullptr, bool *redo = nullptr)
{
    const int64_t F = p.n_frames;
    const int64_t nT = (p.ni + TILE - 1) / TILE;
    const double per_frame_b = 54.0 * (double)p.ni + 4.0 * MORTON_CELLS + 2.0 * (double)nT * (double)nT + 1024.0;
    int64_t batch = (int64_t)(2147483648.0 / per_frame_b);
    if (batch < 1) batch = 1;
    if (batch > 32768) batch = 32768;
    // whole rounds of the spatial sort, which runs one block per frame and CU (its cell counters fill the LDS): 367 C3
    // frames a batch sorted in two rounds of 1.1 ms where 256 take one
    if (batch > ctx->cu_count && ctx->cu_count > 0) batch = batch / ctx->cu_count * ctx->cu_count;
    if (ctx->opt_rdf_batch > 0) batch = ctx->opt_rdf_batch;
    const bool staged = p.h_xi || p.h_xj;
    // the batches: [f0, f0 + n)
    std::vector<std::pair<int64_t, int64_t>> parts;
    {
        int64_t f0 = 0;
        if (staged && F >= 32) {
            // first batch: a quarter of the frames (whole multiples of 8: frames are dealt to the 8 XCDs). Its sweep has
            // to last as long as the copy of the rest — the sweep consumes frames ~3.7x slower than PCIe delivers them
            // at C2 and at C3 size (61 against 230 frames per ms at 10k atoms) — or the second sweep waits for data
            // (a sixth measured 3.94 ms per C2 step with a 0.18 ms stall)
            int64_t b0 = std::max<int64_t>(8, (F / 4) / 8 * 8);
            b0 = std::min(b0, batch);
            parts.emplace_back(0, b0);
            f0 = b0;
        }
        for (; f0 < F; f0 += batch) parts.emplace_back(f0, std::min<int64_t>(batch, F - f0));
    }
    const size_t row = (size_t)p.n_cls * p.nbins;
    const size_t row_cn = (size_t)p.n_cls;
    if (staged) {
        const int rc = stage_batch_async(ctx, p, parts[0].first, parts[0].second, ctx->copy_ev[0]);
        if (rc) return rc;
    }
    if (parts.size() == 1 && !staged) return pair_hist_run_range(ctx, p, 0, F, H, overflow, p.Hsplit, defer, redo);
    H.assign((p.per_frame ? (size_t)F : 1) * row, 0);
    if (p.Hsplit) p.Hsplit->assign((p.per_frame ? (size_t)F : 1) * row_cn, 0);
    *overflow = 0;
    double ms = 0.0, aux = 0.0;
    int launches = 0;
    std::vector<uint64_t> part, part_cn;
    for (size_t k = 0; k < parts.size(); ++k) {
        const int64_t f0 = parts[k].first, n = parts[k].second;
        if (staged) {
            // this batch's frames must have landed before the launch stream touches them; the next batch's copy goes
            // out now, ahead of this batch's kernels
            MD_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_ev[k & 1], 0));
            if (k + 1 < parts.size()) {
                const int rc = stage_batch_async(ctx, p, parts[k + 1].first, parts[k + 1].second, ctx->copy_ev[(k + 1) & 1]);
                if (rc) return rc;
            }
        }
        uint64_t ov = 0;
        const int rc = pair_hist_run_range(ctx, p, f0, n, part, &ov, p.Hsplit ? &part_cn : nullptr);
        if (rc) return rc;
        *overflow += ov;
        ms += ctx->last_ms;
        aux += ctx->last_aux_ms;
        launches += ctx->last_launches;
        if (p.per_frame)
            std::copy(part.begin(), part.end(), H.begin() + (size_t)f0 * row);
        else
            for (size_t q = 0; q < row; ++q) H[q] += part[q];
        if (p.Hsplit) {
            if (p.per_frame)
                std::copy(part_cn.begin(), part_cn.end(), p.Hsplit->begin() + (size_t)f0 * row_cn);
            else
                for (size_t q = 0; q < row_cn; ++q) (*p.Hsplit)[q] += part_cn[q];
        }
    }
    ctx->last_ms = ms;
    ctx->last_aux_ms = aux;
    ctx->last_launches = launches;
    return MDHIP_OK;
}

// The frames [f0, f0 + n) of `p` (their coordinates are on the device): one batch, or — when a block's 32-bit histogram
// words might wrap — halves of it, recursively. Results as pair_hist_run_batch's.
static int pair_hist_run_range(mdhip_ctx *ctx, const PairProblem &p, int64_t f0, int64_t n, std::vector<uint64_t> &H,
                               uint64_t *overflow, std::vector<uint64_t> *Hsplit, CallScope *defer, bool *redo)
{
    PairProblem q = p;
    q.h_xi = q.h_xj = nullptr;
    q.Hsplit = Hsplit;
    q.n_frames = n;
    q.d_xi = p.d_xi + (size_t)f0 * 3 * p.ni;
    q.d_xj = p.d_xj + (size_t)f0 * 3 * p.nj;
    q.d_ti = p.d_ti + (size_t)f0 * p.ti_fs;
    q.d_tj = p.d_tj + (size_t)f0 * p.tj_fs;
    q.d_box = p.d_box + (size_t)f0 * 3;
    q.h_box = p.h_box + (size_t)f0 * 3;
    const int rc1 = pair_hist_run_batch(ctx, q, H, overflow, defer, redo);
    if (rc1 != SPLIT_BATCH) return rc1;
    if (n == 1)
        return mdhip_fail(ctx, MDHIP_ELIMIT, "pair_hist: one frame can overflow the 32-bit block histograms "
                                              "(%lld x %lld atoms)", (long long)p.ni, (long long)p.nj);
    // (device-resident sums, dev_out, are left untouched by a flagged launch)
    const size_t row = (size_t)p.n_cls * p.nbins, row_cn = (size_t)p.n_cls;
    H.assign((p.per_frame ? (size_t)n : 1) * row, 0);
    if (Hsplit) Hsplit->assign((p.per_frame ? (size_t)n : 1) * row_cn, 0);
    *overflow = 0;
    double ms = 0.0, aux = 0.0;
    int launches = 0;
    std::vector<uint64_t> part, part_cn;
    const int64_t half = (n + 1) / 2;
    for (int64_t h0 = 0; h0 < n; h0 += half) {
        uint64_t ov = 0;
        const int rc = pair_hist_run_range(ctx, p, f0 + h0, std::min<int64_t>(half, n - h0), part, &ov,
                                           Hsplit ? &part_cn : nullptr);
        if (rc) return rc;
        *overflow += ov;
        ms += ctx->last_ms;
        aux += ctx->last_aux_ms;
        launches += ctx->last_launches;
        if (p.per_frame)
            std::copy(part.begin(), part.end(), H.begin() + (size_t)h0 * row);
        else
            for (size_t k = 0; k < row; ++k) H[k] += part[k];
        if (Hsplit) {
            if (p.per_frame)
                std::copy(part_cn.begin(), part_cn.end(), Hsplit->begin() + (size_t)h0 * row_cn);
            else
                for (size_t k = 0; k < row_cn; ++k) (*Hsplit)[k] += part_cn[k];
        }
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

// Only labels that some relation names need histogram rows of their own: every other label shares ONE index (its
// pairs all belong to the class "no relation asks for this"). Shrinks the type count the kernels see — e.g. a
// 9-type electrolyte with relations over 3 types runs with 4 — which is what decides whether the table-free
// ordered-row layout (and with it the packed-f32 sweep) fits LDS. `col` = the relation column that refers to this
// atom set (-1: both, the triangular case). uniq: sorted distinct labels (in) -> named labels (out); idx: remapped.
int merge_unnamed_labels(std::vector<int32_t> &uniq, std::vector<int32_t> &idx, int n_rel, const int32_t *rel, int col)
{
    std::vector<int32_t> named;
    std::vector<int32_t> rank(uniq.size(), -1);
    for (size_t k = 0; k < uniq.size(); ++k) {
        bool hit = false;
        for (int kl = 0; kl < n_rel && !hit; ++kl)
            hit = (col != 1 && rel[2 * kl] == uniq[k]) || (col != 0 && rel[2 * kl + 1] == uniq[k]);
        if (hit) {
            rank[k] = (int32_t)named.size();
            named.push_back(uniq[k]);
        }
    }
    const int32_t other = (int32_t)named.size();
    const bool has_other = named.size() < uniq.size();
    for (auto &v : idx) v = rank[(size_t)v] >= 0 ? rank[(size_t)v] : other;
    uniq.swap(named);
    return (int)uniq.size() + (has_other ? 1 : 0);
}

// Relations -> classes. Triangular: unordered type pairs {a,b}; rectangular: ordered (atom type,
// site type). rel_cls[kl] = class id, or -1 when a label does not occur in the data (the reference
// then counts nothing); the last class collects every pair no relation asks for.
// ui / uj hold the labels that have an index of their own (0 .. size-1); n_ti / n_tj may be one larger: the
// index shared by all labels no relation names (merge_unnamed_labels).
void build_classes(bool tri, const std::vector<int32_t> &ui, const std::vector<int32_t> &uj, int n_ti, int n_tj,
                   int n_rel, const int32_t *rel, std::vector<int> &cls, std::vector<int> &rel_cls,
                   int &n_cls)
{
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
    for (size_t k = 0; k < cls.size(); ++k) cls[k] = map[k] < 0 ? next : map[k];
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
    double bin_size;
    int per_frame;
    unsigned long long *dev_out = nullptr;  // see PairProblem::dev_out
    const double *cn_rc2 = nullptr;           // mdhip_rdf_cn_atomic: coordination cutoff^2 per relation, host [n_rel]
    std::vector<uint64_t> *Hsplit = nullptr;  // see PairProblem::Hsplit
    std::vector<double> *cn_c2_cls = nullptr; // out: the cutoff^2 of every class
};

// Stages everything, runs the kernel and returns class histograms + the relation->class map.
// `defer` / `redo`: see pair_hist_run_batch (H, rel_cls, *overflow, j.Hsplit and *redo then live in the entry point's
// heap state; what a deferred step needs from this function's locals is copied into it).
int run_job(mdhip_ctx *ctx, const RelJob &j, std::vector<uint64_t> &H, std::vector<int> &rel_cls,
            int &n_cls, uint64_t *overflow, CallScope *defer = nullptr, bool *redo = nullptr)
{
    std::vector<int32_t> ui, idx_i, uj, idx_j;
    const size_t n_lab_i = j.lab_i_fs ? (size_t)j.F * j.ni : (size_t)j.ni;
    compact_labels(j.lab_i, n_lab_i, ui, idx_i);
    if (!j.tri) compact_labels(j.lab_j, (size_t)j.nj, uj, idx_j);
    const int n_ti = merge_unnamed_labels(ui, idx_i, j.n_rel, j.rel, j.tri ? -1 : 0);
    const int n_tj = j.tri ? n_ti : merge_unnamed_labels(uj, idx_j, j.n_rel, j.rel, 1);
    const std::vector<int32_t> &ujr = j.tri ? ui : uj;
    if ((size_t)n_ti * (size_t)n_tj > 16384)
        return mdhip_fail(ctx, MDHIP_ELIMIT, "too many distinct types named by relations (%d x %d)", n_ti, n_tj);

    PairProblem p;
    p.tri = j.tri;
    build_classes(j.tri, ui, ujr, n_ti, n_tj, j.n_rel, j.rel, p.cls, rel_cls, n_cls);
    p.n_cls = n_cls;
    std::vector<int> rel_mult(j.n_rel, 1);
    if (j.dev_out) {
        for (int kl = 0; kl < j.n_rel; ++kl) rel_mult[kl] = (j.tri && j.rel[2 * kl] == j.rel[2 * kl + 1]) ? 2 : 1;
        p.dev_out = j.dev_out;
        p.n_rel = j.n_rel;
        p.rel_cls = rel_cls.data();
        p.rel_mult = rel_mult.data();
    }
    p.n_ti = n_ti;
    p.n_tj = n_tj;
    if (ctx->opt_rdf_disp != 0) p.disp_rows = displace_rows(n_ti, n_tj, p.cls, p.disp_a, p.disp_b, p.disp_cls);

    int rc;
    // Host-resident coordinates are staged batch by batch under the sweeps (pair_hist_run); the guard drains the copy
    // stream on every way out, so that no copy still reads the caller's arrays after this call has returned.
    struct CopyGuard {
        mdhip_ctx *c;
        bool on = true;
        ~CopyGuard()
        {
            if (on && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        }
    } copy_guard{ctx};
    const bool overlap = ctx->opt_h2d_overlap != 0 && (!j.xi_dev || (!j.tri && !j.xj_dev));
    if (overlap && !ctx->copy_stream) {
        MD_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        MD_HIP(hipEventCreateWithFlags(&ctx->copy_ev[0], hipEventDisableTiming));
        MD_HIP(hipEventCreateWithFlags(&ctx->copy_ev[1], hipEventDisableTiming));
        MD_HIP(hipEventCreateWithFlags(&ctx->stage_ev[0], hipEventDisableTiming));
        MD_HIP(hipEventCreateWithFlags(&ctx->stage_ev[1], hipEventDisableTiming));
    }
    // An ASYNCHRONOUS atom-atom call on host-resident frames: the whole trajectory is copied on the copy stream into the
    // staging buffer the call before this one does not use — i.e. under that call's kernels — and the sweep then runs as
    // on resident frames (one batch, its host half deferred). The caller's array is read until the call completes (the
    // contract of the *_async entry points; a pageable source is staged by the runtime before hipMemcpyAsync returns).
    int stage_buf = -1;
    if (ctx->completing > 0 && (!j.xi_dev || (!j.tri && !j.xj_dev)))
        MD_HIP(mdhip_stream_wait(ctx));  // a re-run from a completion step: the staging buffers may feed queued kernels
    if (overlap && !j.xi_dev && j.tri && defer != nullptr) {
        stage_buf = ctx->stage_flip;
        ctx->stage_flip ^= 1;
        const size_t xb = (size_t)j.F * 3 * j.ni * 8;
        double *d_x = (double *)mdhip_ws(ctx, stage_buf ? WS_XYZ_I2 : WS_XYZ_I, xb);
        if (!d_x) return MDHIP_ENOMEM;
        if (ctx->stage_used[stage_buf]) MD_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->stage_ev[stage_buf], 0));
        {
            const int rch = mdhip_h2d_any(ctx, d_x, j.xi, xb, ctx->copy_stream, stage_buf);
            if (rch) return rch;
        }
        MD_HIP(hipEventRecord(ctx->copy_ev[0], ctx->copy_stream));
        MD_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_ev[0], 0));
        p.d_xi = d_x;
        copy_guard.on = false;
    } else if (overlap && !j.xi_dev) {
        MD_WS(d_x, double, WS_XYZ_I, (size_t)j.F * 3 * j.ni * 8);
        // (a workspace buffer that was just re-allocated may still be read by nothing: mdhip_ws synchronised the stream)
        p.d_xi = d_x;
        p.h_xi = j.xi;
    } else {
        p.d_xi = (const double *)mdhip_stage(ctx, WS_XYZ_I, j.xi, (size_t)j.F * 3 * j.ni * 8, j.xi_dev, &rc);
        if (rc) return rc;
    }
    // compact types of both sets and the box lengths: one pinned staging buffer (owned by the context, so the
    // asynchronous copies need no sync before this function's vectors go away)
    const size_t ti_b = (idx_i.size() * 4 + 63) & ~size_t(63), tj_b = (idx_j.size() * 4 + 63) & ~size_t(63);
    const size_t box_b = (size_t)j.F * 3 * 8;
    MD_PIN(h_in, unsigned char, ti_b + tj_b + box_b);
    memcpy(h_in, idx_i.data(), idx_i.size() * 4);
    memcpy(h_in + ti_b, idx_j.data(), idx_j.size() * 4);
    memcpy(h_in + ti_b + tj_b, j.box, box_b);
    // ONE copy for the three tables (round 5: three copies cost a C2 step ~12 us each, the gaps between a copy and the
    // kernels around it included): types of set i | types of set j | box lengths, behind each other in one buffer
    MD_WS(d_in, unsigned char, WS_TYPE_I, ti_b + tj_b + box_b);
    {
        const int rcc = mdhip_copy_small(ctx, d_in, h_in, ti_b + tj_b + box_b, hipMemcpyHostToDevice);
        if (rcc) return rcc;
    }
    int *d_ti = reinterpret_cast<int *>(d_in);
    p.d_ti = d_ti;
    p.ti_fs = j.lab_i_fs;
    if (j.tri) {
        p.d_xj = p.d_xi;
        p.d_tj = p.d_ti;
        p.tj_fs = p.ti_fs;
        p.nj = j.ni;
    } else {
        if (overlap && !j.xj_dev) {
            MD_WS(d_xjw, double, WS_XYZ_J, (size_t)j.F * 3 * j.nj * 8);
            p.d_xj = d_xjw;
            p.h_xj = j.xj;
        } else {
            p.d_xj = (const double *)mdhip_stage(ctx, WS_XYZ_J, j.xj, (size_t)j.F * 3 * j.nj * 8, j.xj_dev, &rc);
            if (rc) return rc;
        }
        p.d_tj = reinterpret_cast<int *>(d_in + ti_b);
        p.tj_fs = 0;
        p.nj = j.nj;
    }
    p.d_box = reinterpret_cast<double *>(d_in + ti_b + tj_b);
    p.h_box = j.box;
    p.n_frames = j.F;
    p.ni = j.ni;
    p.nbins = j.nbins;
    p.edges = j.edges;
    p.rc2 = j.rc2;
    p.gscale = j.gscale;
    p.bin_size = j.bin_size;
    p.per_frame = j.per_frame;
    if (j.cn_rc2) {
        // one cutoff per class: relations that name the same pair of types with different cutoffs take two sweeps
        std::vector<double> &c2 = *j.cn_c2_cls;
        c2.assign(n_cls, 0.0);
        std::vector<char> seen(n_cls, 0);
        for (int kl = 0; kl < j.n_rel; ++kl) {
            const int cl = rel_cls[kl];
            if (cl < 0) continue;
            const double v = j.cn_rc2[kl] > 0.0 ? j.cn_rc2[kl] : 0.0;
            if (seen[cl] && c2[cl] != v) return CN_UNFUSED;
            seen[cl] = 1;
            c2[cl] = v;
        }
        p.n_cn = 1;
        p.cn_c2_cls = c2.data();
        p.Hsplit = j.Hsplit;
    }
    rc = pair_hist_run(ctx, p, H, overflow, defer, redo);
    if (stage_buf >= 0 && (rc == MDHIP_OK || rc == CN_UNFUSED)) {
        MD_HIP(hipEventRecord(ctx->stage_ev[stage_buf], ctx->stream));  // behind the last kernels that read the buffer
        ctx->stage_used[stage_buf] = true;
    }
    return rc;
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

double mdhip_pk_error_bound(double r_cut, double bin_size, int nbins, int n_rows, double s_cap, double l_max)
{
    return pk_error_bound(r_cut, bin_size, nbins, n_rows, s_cap, l_max);
}

int mdhip_row_displacement(int n_ti, int n_tj, const int32_t *cls, int32_t *a, int32_t *b, int32_t *row_cls, int *n_rows)
{
    if (n_ti < 1 || n_tj < 1 || !cls || !a || !b || !row_cls || !n_rows) return MDHIP_EINVAL;
    std::vector<int> c(cls, cls + (size_t)n_ti * n_tj), va, vb, vr;
    for (int v : c)
        if (v < 0) return MDHIP_EINVAL;
    const int rows = displace_rows(n_ti, n_tj, c, va, vb, vr);
    *n_rows = rows;
    if (rows > 0) {
        std::copy(va.begin(), va.end(), a);
        std::copy(vb.begin(), vb.end(), b);
        std::copy(vr.begin(), vr.end(), row_cls);
    }
    return MDHIP_OK;
}

// What an atom-atom entry point keeps on the heap while its batch may still be in flight (asynchronous calls): the
// class histograms the batch's completion step fills, and what the entry point's own step needs to turn them into the
// reference's outputs.
struct PairState {
    std::vector<uint64_t> H, Hsplit;
    std::vector<int> rel_cls;
    std::vector<double> c2_cls, edges;
    std::vector<int32_t> rel;
    int n_cls = 0;
    uint64_t ov = 0;
    bool redo = false;  // the launch raised the overflow guard of the 32-bit block histograms: run again, synchronously
};

static const double *state_edges(PairState &st, const double *edges, double bin_size, int nbins)
{
    if (edges) {
        st.edges.assign(edges, edges + nbins + 1);
    } else {
        st.edges.resize((size_t)nbins + 1);
        mdhip_bin_edges(bin_size, nbins, st.edges.data());
    }
    return st.edges.data();
}

static void atomic_job(RelJob &j, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device, const int32_t *type,
                       int64_t type_frame_stride, const double *box, int n_rel, const int32_t *rel, double r_cut_sq,
                       double bin_size, int nbins, const double *edges, int per_frame)
{
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
    j.gscale = bin_size > 0.0 ? (float)(1.0 / bin_size) : 0.f;
    j.bin_size = bin_size;
    j.per_frame = per_frame;
}

int mdhip_rdf_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                     int on_device, const int32_t *type, int64_t type_frame_stride,
                     const double *box, int n_rel, const int32_t *rel, double r_cut_sq,
                     double bin_size, int nbins, const double *edges, int per_frame,
                     uint64_t *hist_full, uint64_t *hist_part, uint64_t *overflow)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
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
    if (n_frames == 0 || n_atoms < 2) return cs.end();

    auto st = std::make_shared<PairState>();
    RelJob j{};
    atomic_job(j, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq, bin_size, nbins,
               state_edges(*st, edges, bin_size, nbins), per_frame);
    st->rel.assign(rel, rel + 2 * (size_t)n_rel);
    rc = run_job(ctx, j, st->H, st->rel_cls, st->n_cls, &st->ov, cs.async() ? &cs : nullptr, &st->redo);
    if (rc) return rc;
    cs.defer([=]() {
        if (st->redo)  // (xyz, type, box are the caller's: valid and unchanged until the call has completed)
            return mdhip_rdf_atomic(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel,
                                    st->rel.data(), r_cut_sq, bin_size, nbins, st->edges.data(), per_frame, hist_full,
                                    hist_part, overflow);
        if (overflow) *overflow = st->ov;
        const int n_cls = st->n_cls;
        const int32_t *rl = st->rel.data();
        // rdf_full[bin] += 2 per pair (rdf_cn.py:85-86); rdf_part: +1 per unordered {a,b} pair, +2 when a == b
        for (size_t f = 0; f < out_frames; ++f) {
            const uint64_t *Hf = &st->H[f * n_cls * nbins];
            uint64_t *full = hist_full + f * nbins;
            for (int c = 0; c < n_cls; ++c)
                for (int b = 0; b < nbins; ++b) full[b] += 2 * Hf[(size_t)c * nbins + b];
            for (int kl = 0; kl < n_rel; ++kl) {
                if (st->rel_cls[kl] < 0) continue;
                const uint64_t mult = rl[2 * kl] == rl[2 * kl + 1] ? 2 : 1;
                uint64_t *part = hist_part + (f * n_rel + kl) * nbins;
                const uint64_t *row = Hf + (size_t)st->rel_cls[kl] * nbins;
                for (int b = 0; b < nbins; ++b) part[b] = mult * row[b];
            }
        }
        return (int)MDHIP_OK;
    });
    return cs.end();
}

int mdhip_rdf_atomic_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                         const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                         const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                         uint64_t *out_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(nbins >= 1 && bin_size > 0.0, "nbins and bin_size must be positive");
    MD_REQUIRE(type_frame_stride == 0 || type_frame_stride == n_atoms, "type_frame_stride must be 0 or n_atoms");
    MD_REQUIRE(out_dev, "NULL output");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t words = (size_t)(1 + n_rel) * nbins + 1;
    MD_HIP(hipMemsetAsync(out_dev, 0, words * 8, ctx->stream));
    if (n_frames == 0 || n_atoms < 2) return cs.end();
    auto st = std::make_shared<PairState>();
    RelJob j{};
    atomic_job(j, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq, bin_size, nbins,
               state_edges(*st, edges, bin_size, nbins), 0);
    j.dev_out = reinterpret_cast<unsigned long long *>(out_dev);
    st->rel.assign(rel, rel + 2 * (size_t)n_rel);
    rc = run_job(ctx, j, st->H, st->rel_cls, st->n_cls, &st->ov, cs.async() ? &cs : nullptr, &st->redo);
    if (rc) return rc;
    cs.defer([=]() {
        if (st->redo)  // (device-resident sums are left untouched by a flagged launch)
            return mdhip_rdf_atomic_dev(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel,
                                        st->rel.data(), r_cut_sq, bin_size, nbins, st->edges.data(), out_dev);
        // batches that did not run as one scalar-j pass (small frames, class passes) came back as host class
        // histograms: add them in (rare path; a D2H / H2D round trip of the output words)
        const std::vector<uint64_t> &H = st->H;
        bool any = st->ov != 0;
        for (size_t k = 0; k < H.size() && !any; ++k) any = H[k] != 0;
        if (!any) return (int)MDHIP_OK;
        CallScope fix(ctx);
        const int n_cls = st->n_cls;
        const int32_t *rl = st->rel.data();
        MD_PIN(out, uint64_t, words * 8);
        MD_HIP(hipMemcpyAsync(out, out_dev, words * 8, hipMemcpyDeviceToHost, ctx->stream));
        MD_HIP(mdhip_stream_wait(ctx));
        for (int c = 0; c < n_cls; ++c)
            for (int b = 0; b < nbins; ++b) out[b] += 2 * H[(size_t)c * nbins + b];
        for (int kl = 0; kl < n_rel; ++kl) {
            if (st->rel_cls[kl] < 0) continue;
            const uint64_t mult = rl[2 * kl] == rl[2 * kl + 1] ? 2 : 1;
            for (int b = 0; b < nbins; ++b)
                out[(size_t)(1 + kl) * nbins + b] += mult * H[(size_t)st->rel_cls[kl] * nbins + b];
        }
        out[words - 1] += st->ov;
        MD_HIP(hipMemcpyAsync(out_dev, out, words * 8, hipMemcpyHostToDevice, ctx->stream));
        return fix.end();
    });
    return cs.end();
}

// A finished array of counts goes to the caller: host memory at once, device memory through a copy of its own.
static int deliver_counts(mdhip_ctx *ctx, const uint64_t *src, size_t n, uint64_t *dst, int dst_on_device)
{
    if (!dst_on_device) {
        if (dst != src) memcpy(dst, src, n * 8);
        return MDHIP_OK;
    }
    return mdhip_deliver_to_device(ctx, dst, src, n * 8);
}

// One sweep for the histograms AND the coordination counts. hist_full / hist_part / overflow may be
// NULL (coordination counts only: mdhip_cn_atomic runs its cutoffs through here with a coarse 64-bin histogram whose
// cutoff is the largest coordination cutoff). Returns MDHIP_OK, an error, or CN_UNFUSED when this call has to take the
// two-sweep route (a cutoff beyond r_cut, two cutoffs for one class, a geometry the packed sweep does not take).
// The work is issued into the caller's scope `cs`; its host half is a completion step of that call. cn: host, or
// device memory (cn_on_device; frame-summed only).
static int fused_rdf_cn(CallScope &cs, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                        const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                        const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                        const double *cn_r_cut_sq, int per_frame, uint64_t *hist_full, uint64_t *hist_part,
                        uint64_t *overflow, uint64_t *cn, int cn_on_device)
{
    mdhip_ctx *ctx = cs.ctx;
    bool fused = n_rel > 0 && n_frames > 0 && n_atoms >= 2;
    for (int kl = 0; kl < n_rel && fused; ++kl) fused = !(cn_r_cut_sq[kl] > r_cut_sq);
    if (!fused) return CN_UNFUSED;
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    auto st = std::make_shared<PairState>();
    RelJob j{};
    atomic_job(j, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq, bin_size, nbins,
               state_edges(*st, edges, bin_size, nbins), per_frame);
    st->rel.assign(rel, rel + 2 * (size_t)n_rel);
    auto cuts = std::make_shared<std::vector<double>>(cn_r_cut_sq, cn_r_cut_sq + n_rel);
    j.cn_rc2 = cuts->data();
    j.Hsplit = &st->Hsplit;
    j.cn_c2_cls = &st->c2_cls;
    const int rc = run_job(ctx, j, st->H, st->rel_cls, st->n_cls, &st->ov, cs.async() ? &cs : nullptr, &st->redo);
    if (rc != MDHIP_OK) return rc;  // an error, or CN_UNFUSED
    cs.defer([=]() {
        if (st->redo) {
            // the same sweep again, inside a synchronous call of its own (which halves the batch where it has to)
            CallScope again(ctx);
            const int rc2 = fused_rdf_cn(again, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel,
                                         st->rel.data(), r_cut_sq, bin_size, nbins, st->edges.data(), cuts->data(),
                                         per_frame, hist_full, hist_part, overflow, cn, cn_on_device);
            if (rc2 != MDHIP_OK)
                return rc2 == CN_UNFUSED ? mdhip_fail(ctx, MDHIP_EHIP, "pair_hist: the one-sweep path refused its own re-run") : rc2;
            return again.end();
        }
        const std::vector<uint64_t> &H = st->H, &Hsplit = st->Hsplit;
        const double *use_edges = st->edges.data();
        const int32_t *rl = st->rel.data();
        const int n_cls = st->n_cls;
        if (overflow) *overflow = st->ov;
        if (hist_full) std::fill(hist_full, hist_full + out_frames * nbins, (uint64_t)0);
        std::vector<uint64_t> cn_host(cn_on_device ? out_frames * (size_t)n_rel : 0);
        uint64_t *cn_out = cn_on_device ? cn_host.data() : cn;
        for (size_t f = 0; f < out_frames; ++f) {
            const uint64_t *Hf = &H[f * n_cls * nbins];
            if (hist_full) {
                uint64_t *full = hist_full + f * nbins;
                for (int c = 0; c < n_cls; ++c)
                    for (int b = 0; b < nbins; ++b) full[b] += 2 * Hf[(size_t)c * nbins + b];
            }
            for (int kl = 0; kl < n_rel; ++kl) {
                uint64_t *part = hist_part ? hist_part + (f * n_rel + kl) * nbins : nullptr;
                uint64_t s = 0;
                const uint64_t mult = rl[2 * kl] == rl[2 * kl + 1] ? 2 : 1;
                if (st->rel_cls[kl] < 0) {
                    if (part) std::fill(part, part + nbins, (uint64_t)0);
                } else {
                    const int cl = st->rel_cls[kl];
                    const uint64_t *row = Hf + (size_t)cl * nbins;
                    if (part)
                        for (int b = 0; b < nbins; ++b) part[b] = mult * row[b];
                    const double c2 = st->c2_cls[cl];
                    if (c2 > 0.0) {
                        // bins below the split bin are inside the cutoff (exact edges); the split bin's share was
                        // counted by the exact chain (a cutoff inside the overflow bin, index nbins: all bins plus
                        // the overflow pairs below the cutoff)
                        const int kc = (int)(std::upper_bound(use_edges, use_edges + nbins + 1, c2) - use_edges) - 1;
                        for (int b = 0; b < kc && b < nbins; ++b) s += row[b];
                        s += Hsplit[f * n_cls + cl];
                    }
                }
                cn_out[f * n_rel + kl] = mult * s;
            }
        }
        return deliver_counts(ctx, cn_out, out_frames * (size_t)n_rel, cn, cn_on_device);
    });
    return MDHIP_OK;
}

static int cn_atomic_impl(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                          const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                          const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn, int cn_on_device);

int mdhip_rdf_cn_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                        const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                        const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                        const double *cn_r_cut_sq, int per_frame, uint64_t *hist_full, uint64_t *hist_part,
                        uint64_t *overflow, uint64_t *cn)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(nbins >= 1 && bin_size > 0.0, "nbins and bin_size must be positive");
    MD_REQUIRE(type_frame_stride == 0 || type_frame_stride == n_atoms, "type_frame_stride must be 0 or n_atoms");
    MD_REQUIRE(hist_full && (n_rel == 0 || (hist_part && cn_r_cut_sq && cn)), "NULL output or cutoff array");
    rc = fused_rdf_cn(cs, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                      bin_size, nbins, edges, cn_r_cut_sq, per_frame, hist_full, hist_part, overflow, cn, 0);
    if (rc == MDHIP_OK) return cs.end();
    if (rc != CN_UNFUSED) return rc;
    // this call does not run as one packed sweep — two sweeps (calls of their own, complete on return), same integers
    rc = mdhip_rdf_atomic(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                          bin_size, nbins, edges, per_frame, hist_full, hist_part, overflow);
    if (rc) return rc;
    rc = cn_atomic_impl(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel,
                        cn_r_cut_sq, per_frame, cn, 0);
    if (rc) return rc;
    return cs.end();
}

static int cn_atomic_impl(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                          const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                          const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn, int cn_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(n_rel == 0 || (r_cut_sq && cn), "NULL cutoff or output");
    MD_REQUIRE(type_frame_stride == 0 || type_frame_stride == n_atoms,
               "type_frame_stride must be 0 or n_atoms");
    MD_REQUIRE(!cn_on_device || !per_frame, "a device result buffer holds the frame-summed counts");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    rc = mdhip_zero_result(ctx, cn, out_frames * (size_t)n_rel * 8, cn_on_device);
    if (rc) return rc;
    if (n_frames == 0 || n_atoms < 2 || n_rel == 0) return cs.end();
    std::vector<double> edges;
    std::vector<int> rank;
    cn_edges(n_rel, r_cut_sq, edges, rank);
    const int nbins = (int)edges.size() - 1;
    if (nbins == 0) return cs.end();
    if (ctx->opt_cn_pk != 0) {
        // Option cn_pk (default): the packed-f32 sweep with a coarse histogram whose cutoff is the largest coordination
        // cutoff; the counts are the exact bins below each class's split bin plus the split-bin pairs the exact chain
        // finds inside — the same integers as the f64 edge-table kernel below, which is the route for
        // everything the packed sweep does not take (small frames, two cutoffs for one class, ...) and for cn_pk = 0.
        const double c_max = std::sqrt(edges.back());
        const int nb = 64;
        rc = fused_rdf_cn(cs, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel,
                          edges.back(), c_max / nb, nb, nullptr, r_cut_sq, per_frame, nullptr, nullptr, nullptr, cn,
                          cn_on_device);
        if (rc == MDHIP_OK) return cs.end();
        if (rc != CN_UNFUSED) return rc;
    }
    RelJob j{};
    atomic_job(j, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, edges.back(), 0.0, nbins,
               edges.data(), per_frame);
    std::vector<uint64_t> H;
    std::vector<int> rel_cls;
    int n_cls = 0;
    uint64_t ov = 0;
    rc = run_job(ctx, j, H, rel_cls, n_cls, &ov);  // (edge-table kernel: completes inside)
    if (rc) return rc;
    std::vector<uint64_t> cn_host(out_frames * (size_t)n_rel, 0);
    for (size_t f = 0; f < out_frames; ++f)
        for (int kl = 0; kl < n_rel; ++kl) {
            if (rel_cls[kl] < 0) continue;
            const uint64_t mult = rel[2 * kl] == rel[2 * kl + 1] ? 2 : 1;
            const uint64_t *row = &H[(f * n_cls + rel_cls[kl]) * nbins];
            uint64_t s = 0;
            for (int b = 0; b < rank[kl]; ++b) s += row[b];
            cn_host[f * n_rel + kl] = mult * s;
        }
    if (cn_on_device) {
        rc = mdhip_h2d_small(ctx, cn, cn_host.data(), cn_host.size() * 8);
        if (rc) return rc;
    } else {
        memcpy(cn, cn_host.data(), cn_host.size() * 8);
    }
    return cs.end();
}

int mdhip_cn_atomic(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                    int on_device, const int32_t *type, int64_t type_frame_stride,
                    const double *box, int n_rel, const int32_t *rel, const double *r_cut_sq,
                    int per_frame, uint64_t *cn)
{
    return cn_atomic_impl(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                          per_frame, cn, 0);
}

int mdhip_cn_atomic_dev(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                        const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                        const int32_t *rel, const double *r_cut_sq, uint64_t *cn_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    MD_REQUIRE(n_rel == 0 || cn_dev, "NULL output");
    return cn_atomic_impl(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq, 0,
                          cn_dev, 1);
}

/* ---- asynchronous twins (results complete after mdhip_sync / mdhip_wait; inputs must stay valid until then) ---- */
int mdhip_rdf_atomic_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                           const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                           const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                           int per_frame, uint64_t *hist_full, uint64_t *hist_part, uint64_t *overflow)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_rdf_atomic(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                            bin_size, nbins, edges, per_frame, hist_full, hist_part, overflow);
}

int mdhip_rdf_atomic_dev_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                               const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                               const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                               uint64_t *out_dev)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_rdf_atomic_dev(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                                bin_size, nbins, edges, out_dev);
}

int mdhip_rdf_cn_atomic_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                              const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                              const int32_t *rel, double r_cut_sq, double bin_size, int nbins, const double *edges,
                              const double *cn_r_cut_sq, int per_frame, uint64_t *hist_full, uint64_t *hist_part,
                              uint64_t *overflow, uint64_t *cn)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return mdhip_rdf_cn_atomic(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                               bin_size, nbins, edges, cn_r_cut_sq, per_frame, hist_full, hist_part, overflow, cn);
}

int mdhip_cn_atomic_async(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz, int on_device,
                          const int32_t *type, int64_t type_frame_stride, const double *box, int n_rel,
                          const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn, int cn_on_device)
{
    if (!ctx) return MDHIP_EINVAL;
    AsyncCall mark(ctx);
    return cn_atomic_impl(ctx, n_frames, n_atoms, xyz, on_device, type, type_frame_stride, box, n_rel, rel, r_cut_sq,
                          per_frame, cn, cn_on_device ? 1 : 0);
}

int mdhip_rdf_sites(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                    int xyz_on_device, const int32_t *type, int64_t n_sites, const double *sites,
                    int sites_on_device, const int32_t *site_type, const double *box, int n_rel,
                    const int32_t *rel, double r_cut_sq, double bin_size, int nbins,
                    const double *edges, int per_frame, uint64_t *hist_part, uint64_t *overflow)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);  // (atoms x sites: completes inside)
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
    if (n_frames == 0 || n_atoms == 0 || n_sites == 0 || n_rel == 0) return cs.end();
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
    j.bin_size = bin_size;
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
    return cs.end();
}

int mdhip_cn_sites(mdhip_ctx *ctx, int64_t n_frames, int64_t n_atoms, const double *xyz,
                   int xyz_on_device, const int32_t *type, int64_t n_sites, const double *sites,
                   int sites_on_device, const int32_t *site_type, const double *box, int n_rel,
                   const int32_t *rel, const double *r_cut_sq, int per_frame, uint64_t *cn)
{
    if (!ctx) return MDHIP_EINVAL;
    CallScope cs(ctx);  // (atoms x sites: completes inside)
    int rc = check_common(ctx, n_frames, n_atoms, xyz, type, box, n_rel, rel);
    if (rc) return rc;
    MD_REQUIRE(n_sites >= 0 && n_sites < (1LL << 31), "bad n_sites");
    MD_REQUIRE(n_frames == 0 || n_sites == 0 || (sites && site_type), "NULL site arrays");
    MD_REQUIRE(n_rel == 0 || (r_cut_sq && cn), "NULL cutoff or output");
    MD_HIP(hipSetDevice(ctx->device));
    const size_t out_frames = per_frame ? (size_t)n_frames : 1;
    std::fill(cn, cn + out_frames * n_rel, (uint64_t)0);
    if (n_frames == 0 || n_atoms == 0 || n_sites == 0 || n_rel == 0) return cs.end();
    std::vector<double> edges;
    std::vector<int> rank;
    cn_edges(n_rel, r_cut_sq, edges, rank);
    const int nbins = (int)edges.size() - 1;
    if (nbins == 0) return cs.end();
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
    return cs.end();
}

}  // extern "C"
