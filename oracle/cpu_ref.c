/*
 * oracle/cpu_ref.c — single-thread C restatement of the reference's hot loops.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/cpu_ref.py for the rule): used by
 * tests/ as a checker at sizes numpy is too slow for, and by bench.py's
 * cpu_baseline leg ("port", 1 core) as a stand-in for the reference's
 * numba-JIT speed, which cannot be measured in this environment.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC  (oracle/Makefile)
 * -ffp-contract=off matters: the reference never fuses a*b+c, and a fused rsq
 * can move a pair across a bin or cutoff edge.
 *
 * Parity pin: checked against the tests/golden npz fixtures (outputs of the real
 * reference) by tests/test_oracle_golden.py.
 * Citations: file:line under /root/reference/mdproptools/.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* structural/rdf_cn.py:44-57: d = head - other; one shift by -sign(d)*L iff |d| > L/2 (strict). */
static inline double wrap1(double d, double L, double half)
{
    if (d > half)
        d = d - L;
    else if (d < -half)
        d = d + L; /* d - (-1)*L */
    return d;
}

static inline double pair_rsq(double xi, double yi, double zi, double xj, double yj, double zj,
                              const double *L, const double *H)
{
    double dx = wrap1(xi - xj, L[0], H[0]);
    double dy = wrap1(yi - yj, L[1], H[1]);
    double dz = wrap1(zi - zj, L[2], H[2]);
    return (dx * dx + dy * dy) + dz * dz; /* rdf_cn.py:56, left to right */
}

/*
 * structural/rdf_cn.py:72-97 (_rdf_loop), upper triangle i < j.
 * xyz: SoA planes [3][n]; type: labels as in the dump; rel: [n_rel][2] labels.
 * full[bin] += 2; part[kl][bin] += 1 per (head a, other b) and per (head b, other a).
 * bin = trunc(sqrt(rsq)/ddr) (rdf_cn.py:68,85). bin == nbins is dropped and counted.
 */
/* Head rows i0 <= i < i1 only (the reference's outer loop, rdf_cn.py:82, restricted): a bounded sample of a
 * large frame for bench.py's cpu_baseline, and what lets tests split a frame over host threads. */
void oracle_rdf_pairs_rows(int64_t n, int64_t i0, int64_t i1, const double *xyz, const int32_t *type, int n_rel,
                           const int32_t *rel, const double *lengths, double rc2, double ddr,
                           int nbins, uint64_t *full, uint64_t *part, uint64_t *overflow)
{
    const double *x = xyz, *y = xyz + n, *z = xyz + 2 * n;
    double H[3] = {lengths[0] / 2, lengths[1] / 2, lengths[2] / 2};
    if (i1 > n - 1)
        i1 = n - 1;
    for (int64_t i = i0 < 0 ? 0 : i0; i < i1; ++i) {
        const double xi = x[i], yi = y[i], zi = z[i];
        const int32_t ti = type[i];
        for (int64_t j = i + 1; j < n; ++j) {
            double rsq = pair_rsq(xi, yi, zi, x[j], y[j], z[j], lengths, H);
            if (!(rsq < rc2))
                continue;
            int64_t b = (int64_t)(sqrt(rsq) / ddr);
            if (b >= nbins) {
                ++*overflow;
                continue;
            }
            full[b] += 2;
            const int32_t tj = type[j];
            for (int kl = 0; kl < n_rel; ++kl) {
                const int32_t a = rel[2 * kl], c = rel[2 * kl + 1];
                if (ti == a && tj == c)
                    part[(int64_t)kl * nbins + b] += 1;
                if (ti == c && tj == a)
                    part[(int64_t)kl * nbins + b] += 1;
            }
        }
    }
}

void oracle_rdf_pairs(int64_t n, const double *xyz, const int32_t *type, int n_rel,
                      const int32_t *rel, const double *lengths, double rc2, double ddr,
                      int nbins, uint64_t *full, uint64_t *part, uint64_t *overflow)
{
    oracle_rdf_pairs_rows(n, 0, n - 1, xyz, type, n_rel, rel, lengths, rc2, ddr, nbins, full, part, overflow);
}

/* structural/rdf_cn.py:100-119 (_cn_loop): per-relation cutoff rc2[kl]; head rows i0 <= i < i1. */
void oracle_cn_pairs_rows(int64_t n, int64_t i0, int64_t i1, const double *xyz, const int32_t *type, int n_rel,
                          const int32_t *rel, const double *lengths, const double *rc2, uint64_t *cn)
{
    const double *x = xyz, *y = xyz + n, *z = xyz + 2 * n;
    double H[3] = {lengths[0] / 2, lengths[1] / 2, lengths[2] / 2};
    double rcmax = 0;
    for (int kl = 0; kl < n_rel; ++kl)
        if (rc2[kl] > rcmax)
            rcmax = rc2[kl];
    if (i1 > n - 1)
        i1 = n - 1;
    for (int64_t i = i0 < 0 ? 0 : i0; i < i1; ++i) {
        const double xi = x[i], yi = y[i], zi = z[i];
        const int32_t ti = type[i];
        for (int64_t j = i + 1; j < n; ++j) {
            double rsq = pair_rsq(xi, yi, zi, x[j], y[j], z[j], lengths, H);
            if (!(rsq < rcmax))
                continue;
            const int32_t tj = type[j];
            for (int kl = 0; kl < n_rel; ++kl) {
                if (!(rsq < rc2[kl]))
                    continue;
                const int32_t a = rel[2 * kl], c = rel[2 * kl + 1];
                if (ti == a && tj == c)
                    cn[kl] += 1;
                if (ti == c && tj == a)
                    cn[kl] += 1;
            }
        }
    }
}

void oracle_cn_pairs(int64_t n, const double *xyz, const int32_t *type, int n_rel,
                     const int32_t *rel, const double *lengths, const double *rc2, uint64_t *cn)
{
    oracle_cn_pairs_rows(n, 0, n - 1, xyz, type, n_rel, rel, lengths, rc2, cn);
}

/* structural/rdf_cn.py:122-141 (_rdf_mol_loop): atoms x sites, +1 when (atom a, site b). */
void oracle_rdf_rect(int64_t n, const double *xyz, const int32_t *type, int64_t m,
                     const double *sxyz, const int32_t *stype, int n_rel, const int32_t *rel,
                     const double *lengths, double rc2, double ddr, int nbins, uint64_t *part,
                     uint64_t *overflow)
{
    const double *x = xyz, *y = xyz + n, *z = xyz + 2 * n;
    const double *sx = sxyz, *sy = sxyz + m, *sz = sxyz + 2 * m;
    double H[3] = {lengths[0] / 2, lengths[1] / 2, lengths[2] / 2};
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ti = type[i];
        for (int64_t j = 0; j < m; ++j) {
            double rsq = pair_rsq(x[i], y[i], z[i], sx[j], sy[j], sz[j], lengths, H);
            if (!(rsq < rc2))
                continue;
            int64_t b = (int64_t)(sqrt(rsq) / ddr);
            if (b >= nbins) {
                ++*overflow;
                continue;
            }
            for (int kl = 0; kl < n_rel; ++kl)
                if (ti == rel[2 * kl] && stype[j] == rel[2 * kl + 1])
                    part[(int64_t)kl * nbins + b] += 1;
        }
    }
}

/* structural/rdf_cn.py:144-162 (_cn_mol_loop). */
void oracle_cn_rect(int64_t n, const double *xyz, const int32_t *type, int64_t m,
                    const double *sxyz, const int32_t *stype, int n_rel, const int32_t *rel,
                    const double *lengths, const double *rc2, uint64_t *cn)
{
    const double *x = xyz, *y = xyz + n, *z = xyz + 2 * n;
    const double *sx = sxyz, *sy = sxyz + m, *sz = sxyz + 2 * m;
    double H[3] = {lengths[0] / 2, lengths[1] / 2, lengths[2] / 2};
    for (int64_t i = 0; i < n; ++i) {
        const int32_t ti = type[i];
        for (int64_t j = 0; j < m; ++j) {
            double rsq = pair_rsq(x[i], y[i], z[i], sx[j], sy[j], sz[j], lengths, H);
            for (int kl = 0; kl < n_rel; ++kl)
                if (rsq < rc2[kl] && ti == rel[2 * kl] && stype[j] == rel[2 * kl + 1])
                    cn[kl] += 1;
        }
    }
}

/*
 * dynamical/diffusion.py:212-218: for frame pairs (t0,t1), sum over the
 * entities of each contiguous group of (r(t1)-r(t0))^2 per axis and of
 * (dx2+dy2)+dz2. r: [F][3][E] already in output units. out: [P][G][4] sums.
 */
void oracle_msd_pairs(int64_t n_ent, const double *r, int n_pairs, const int32_t *pairs,
                      int n_groups, const int64_t *group_off, double *out)
{
    for (int p = 0; p < n_pairs; ++p) {
        const double *a = r + (int64_t)pairs[2 * p] * 3 * n_ent;
        const double *b = r + (int64_t)pairs[2 * p + 1] * 3 * n_ent;
        for (int g = 0; g < n_groups; ++g) {
            double s[4] = {0, 0, 0, 0};
            for (int64_t e = group_off[g]; e < group_off[g + 1]; ++e) {
                double dx = b[e] - a[e], dy = b[n_ent + e] - a[n_ent + e],
                       dz = b[2 * n_ent + e] - a[2 * n_ent + e];
                double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
                s[0] += dx2;
                s[1] += dy2;
                s[2] += dz2;
                s[3] += (dx2 + dy2) + dz2;
            }
            memcpy(out + ((int64_t)p * n_groups + g) * 4, s, sizeof s);
        }
    }
}

/* dynamical/viscosity.py:103-108: c[k] = sum_{t<n-k} a[t+k]*b[t] / (n-k), k < n_lags. */
void oracle_xcorr_direct(int64_t n, const double *a, const double *b, int64_t n_lags, double *out)
{
    for (int64_t k = 0; k < n_lags; ++k) {
        double s = 0;
        for (int64_t t = 0; t + k < n; ++t)
            s += a[t + k] * b[t];
        out[k] = s / (double)(n - k);
    }
}

/*
 * Full lag x origin MSD (superset of dynamical/diffusion.py:225-238, which keeps one lag): for lag k = 0..max_lag
 * the mean over origins t0 < F - k and over the entities of each group of (r(t0+k) - r(t0))^2 per axis and of
 * (dx2+dy2)+dz2. r: [F][3][E]; lags: the n_lags lags to evaluate (any subset); out [n_lags][G][4].
 */
void oracle_lag_msd(int64_t n_frames, int64_t n_ent, const double *r, int n_lags, const int32_t *lags,
                    int n_groups, const int64_t *group_off, double *out)
{
    for (int q = 0; q < n_lags; ++q) {
        const int64_t k = lags[q];
        for (int g = 0; g < n_groups; ++g) {
            double s[4] = {0, 0, 0, 0};
            for (int64_t t0 = 0; t0 + k < n_frames; ++t0) {
                const double *a = r + t0 * 3 * n_ent, *b = r + (t0 + k) * 3 * n_ent;
                for (int64_t e = group_off[g]; e < group_off[g + 1]; ++e) {
                    double dx = b[e] - a[e], dy = b[n_ent + e] - a[n_ent + e],
                           dz = b[2 * n_ent + e] - a[2 * n_ent + e];
                    double dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
                    s[0] += dx2;
                    s[1] += dy2;
                    s[2] += dz2;
                    s[3] += (dx2 + dy2) + dz2;
                }
            }
            const double cnt = (double)(n_frames - k) * (double)(group_off[g + 1] - group_off[g]);
            for (int c = 0; c < 4; ++c)
                out[((int64_t)q * n_groups + g) * 4 + c] = cnt > 0 ? s[c] / cnt : 0.0;
        }
    }
}
