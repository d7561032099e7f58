"""
oracle/cpu_ref.py — CPU restatement (numpy) of the reference's hot-path algorithms.

TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (mdproptools_amd/) never
does and fails loudly when libmdhip.so is missing.

Parity pin: every function here is checked against golden vectors produced by
importing the real reference in the build container (oracle/make_golden.py ->
tests/golden/*.npz; tests/test_oracle_golden.py), and the diffusion chain is
additionally pinned by the reference's only published known answer
(examples/mg_tfsi_dme_analysis.ipynb cell 17).

All citations are file:line under /root/reference/mdproptools/.
Everything is float64; products and sums are separate IEEE operations (numpy
never contracts them), which is what the reference's numba/numpy code does.
"""

import numpy as np

# ----------------------------------------------------------------------------
# R1-R5: all-pairs minimum-image distance, cutoff filter, binning, counting
# ----------------------------------------------------------------------------


def min_image_rsq(head_xyz, other_xyz, lengths):
    """
    structural/rdf_cn.py:44-57 (_calc_rsq). d = head - other per axis; ONE
    conditional shift by -sign(d)*L when |d| > L/2 (strict); rsq =
    (dx*dx + dy*dy) + dz*dz. `head_xyz` [..., 3] broadcasts against
    `other_xyz` [..., 3].
    """
    d = head_xyz - other_xyz
    for ax in range(3):
        L = lengths[ax]
        c = d[..., ax]
        cond = (c > L / 2) | (c < -L / 2)
        c[cond] = c[cond] - np.sign(c[cond]) * L
    return d[..., 0] ** 2 + d[..., 1] ** 2 + d[..., 2] ** 2


def bin_index(rsq, ddr):
    """structural/rdf_cn.py:68,85: trunc( sqrt(rsq) / ddr ) (a division, not a reciprocal multiply)."""
    return (np.sqrt(rsq) / ddr).astype(np.int64)


def _pair_blocks(n, block):
    for i0 in range(0, n, block):
        yield i0, min(n, i0 + block)


def rdf_pairs(data, relation_matrix, lengths, r_cut, ddr, nbins, block=256):
    """
    structural/rdf_cn.py:72-97 (_rdf_loop) over the upper triangle i < j.

    data [N,4] = [type, x, y, z]; relation_matrix [R,2] ints.
    Returns (rdf_full int64[nbins], rdf_part int64[R,nbins], overflow int):
      rdf_full[bin] += 2 per in-cutoff pair (rdf_cn.py:85-86);
      rdf_part[kl][bin] += 1 if head is a and other is b, and += 1 if head is
      b and other is a (rdf_cn.py:87-96) — so a == b pairs count 2.
    Pairs with rsq < r_cut**2 whose bin index equals nbins (SURVEY.md fact 7;
    the reference indexes out of bounds there) are dropped and counted in
    `overflow`.
    """
    data = np.asarray(data, dtype=np.float64)
    rel = np.asarray(relation_matrix, dtype=np.int64).reshape(-1, 2)
    n = data.shape[0]
    types = data[:, 0].astype(np.int64)
    xyz = data[:, 1:4]
    full = np.zeros(nbins, dtype=np.int64)
    part = np.zeros((len(rel), nbins), dtype=np.int64)
    overflow = 0
    rc2 = r_cut ** 2
    for i0, i1 in _pair_blocks(n - 1, block):
        js = np.arange(i0 + 1, n)
        rsq = min_image_rsq(xyz[i0:i1, None, :], xyz[None, js, :].copy(), lengths)
        keep = (rsq < rc2) & (js[None, :] > np.arange(i0, i1)[:, None])
        ii, jj = np.nonzero(keep)
        b = bin_index(rsq[ii, jj], ddr)
        ok = b < nbins
        overflow += int((~ok).sum())
        ii, jj, b = ii[ok], jj[ok], b[ok]
        full += 2 * np.bincount(b, minlength=nbins)
        th = types[i0:i1][ii]
        to = types[js][jj]
        for kl, (a, c) in enumerate(rel):
            m1 = (th == a) & (to == c)
            m2 = (th == c) & (to == a)
            part[kl] += np.bincount(b[m1], minlength=nbins)
            part[kl] += np.bincount(b[m2], minlength=nbins)
    return full, part, overflow


def cn_pairs(data, relation_matrix, lengths, r_cut_list, block=256):
    """
    structural/rdf_cn.py:100-119 (_cn_loop): per relation its own cutoff,
    cn[kl] += #pairs(rsq < r_cut[kl]**2) with the same a/b double test.
    """
    data = np.asarray(data, dtype=np.float64)
    rel = np.asarray(relation_matrix, dtype=np.int64).reshape(-1, 2)
    n = data.shape[0]
    types = data[:, 0].astype(np.int64)
    xyz = data[:, 1:4]
    cn = np.zeros(len(rel), dtype=np.int64)
    rc2 = [rc ** 2 for rc in r_cut_list]
    rc2max = max(rc2)
    for i0, i1 in _pair_blocks(n - 1, block):
        js = np.arange(i0 + 1, n)
        rsq = min_image_rsq(xyz[i0:i1, None, :], xyz[None, js, :].copy(), lengths)
        keep = (rsq < rc2max) & (js[None, :] > np.arange(i0, i1)[:, None])
        ii, jj = np.nonzero(keep)
        r = rsq[ii, jj]
        th = types[i0:i1][ii]
        to = types[js][jj]
        for kl, (a, c) in enumerate(rel):
            inside = r < rc2[kl]
            cn[kl] += int((inside & (th == a) & (to == c)).sum())
            cn[kl] += int((inside & (th == c) & (to == a)).sum())
    return cn


def rdf_mol_pairs(atom_data, mol_data, relation_matrix, lengths, r_cut, ddr, nbins, block=512):
    """
    structural/rdf_cn.py:122-141 (_rdf_mol_loop): every atom (type a) against
    every molecule site (mol type b), +1 per in-cutoff pair; the atom's own
    molecule is not excluded. Returns (rdf_part int64[R,nbins], overflow).
    """
    atom_data = np.asarray(atom_data, dtype=np.float64)
    mol_data = np.asarray(mol_data, dtype=np.float64)
    rel = np.asarray(relation_matrix, dtype=np.int64).reshape(-1, 2)
    at = atom_data[:, 0].astype(np.int64)
    mt = mol_data[:, 0].astype(np.int64)
    part = np.zeros((len(rel), nbins), dtype=np.int64)
    overflow = 0
    rc2 = r_cut ** 2
    for i0, i1 in _pair_blocks(atom_data.shape[0], block):
        rsq = min_image_rsq(
            atom_data[i0:i1, None, 1:4], mol_data[None, :, 1:4].copy(), lengths
        )
        ii, jj = np.nonzero(rsq < rc2)
        b = bin_index(rsq[ii, jj], ddr)
        ok = b < nbins
        overflow += int((~ok).sum())
        ii, jj, b = ii[ok], jj[ok], b[ok]
        th = at[i0:i1][ii]
        to = mt[jj]
        for kl, (a, c) in enumerate(rel):
            part[kl] += np.bincount(b[(th == a) & (to == c)], minlength=nbins)
    return part, overflow


def cn_mol_pairs(atom_data, mol_data, relation_matrix, lengths, r_cut_list, block=512):
    """structural/rdf_cn.py:144-162 (_cn_mol_loop)."""
    atom_data = np.asarray(atom_data, dtype=np.float64)
    mol_data = np.asarray(mol_data, dtype=np.float64)
    rel = np.asarray(relation_matrix, dtype=np.int64).reshape(-1, 2)
    at = atom_data[:, 0].astype(np.int64)
    mt = mol_data[:, 0].astype(np.int64)
    cn = np.zeros(len(rel), dtype=np.int64)
    rc2 = [rc ** 2 for rc in r_cut_list]
    for i0, i1 in _pair_blocks(atom_data.shape[0], block):
        rsq = min_image_rsq(
            atom_data[i0:i1, None, 1:4], mol_data[None, :, 1:4].copy(), lengths
        )
        for kl, (a, c) in enumerate(rel):
            sel = rsq[np.ix_(at[i0:i1] == a, mt == c)]
            cn[kl] += int((sel < rc2[kl]).sum())
    return cn


# ----------------------------------------------------------------------------
# R6, R7, M3: molecule membership, wrapped/unwrapped centres of mass, altered ids
# ----------------------------------------------------------------------------


def molecule_layout(num_mols, num_atoms_per_mol):
    """
    Membership implied by sorted-id order (type-major, then molecule, then
    atom): structural/rdf_cn.py:222-230, common/com_mols.py:31-42.
    Returns (mol_type int64[N], mol_id int64[N], seg_offsets int64[M+1],
    seg_type int64[M]).
    """
    mol_type, mol_id, offsets, seg_type = [], [], [0], []
    for t, (nm, na) in enumerate(zip(num_mols, num_atoms_per_mol)):
        for m in range(nm):
            mol_type.extend([t + 1] * na)
            mol_id.extend([m + 1] * na)
            offsets.append(offsets[-1] + na)
            seg_type.append(t + 1)
    return (
        np.asarray(mol_type, dtype=np.int64),
        np.asarray(mol_id, dtype=np.int64),
        np.asarray(offsets, dtype=np.int64),
        np.asarray(seg_type, dtype=np.int64),
    )


def mol_com_dot(xyz, atom_mass, seg_offsets):
    """
    structural/rdf_cn.py:233-238 (_define_mol_cols) and dynamical/diffusion.py:83-89:
    com = (mass @ xyz) / mass.sum() per segment. xyz [N,3], atom_mass [N].
    """
    m = len(seg_offsets) - 1
    out = np.zeros((m, 3))
    for s in range(m):
        lo, hi = seg_offsets[s], seg_offsets[s + 1]
        out[s] = atom_mass[lo:hi] @ xyz[lo:hi] / atom_mass[lo:hi].sum()
    return out


def calc_com(attrs, atom_mass, seg_offsets, charge=None):
    """
    common/com_mols.py:58-60 (calc_com): attr*mass, segment sums, divide by
    the segment mass. attrs [N,K]. Returns (com [M,K], seg_mass [M], seg_q [M] or None).
    """
    w = attrs * atom_mass[:, None]
    starts = seg_offsets[:-1]
    seg_mass = np.add.reduceat(atom_mass, starts)
    com = np.add.reduceat(w, starts, axis=0) / seg_mass[:, None]
    seg_q = None if charge is None else np.add.reduceat(charge, starts)
    return com, seg_mass, seg_q


def calc_atom_type(ids, num_mols, num_atoms):
    """
    structural/rdf_cn.py:197-215 (_calc_atom_type): atom id -> index of the
    atom inside its molecule type (1-based), offset by the atom counts of the
    preceding molecule types. ids: float64 or int array of LAMMPS atom ids.
    """
    ids = np.asarray(ids, dtype=np.float64)
    num_atoms = np.asarray(num_atoms)
    totals = np.multiply(num_mols, num_atoms)
    cutoff = np.cumsum(totals)
    out = ids.copy()
    done = np.zeros(ids.shape, dtype=bool)
    for i, c in enumerate(cutoff):
        sel = (~done) & (ids <= c)
        v = np.mod(ids[sel] - c, num_atoms[i])  # Python % semantics (rdf_cn.py:209)
        v[v == 0] = num_atoms[i]
        if i > 0:
            v = v + np.sum(num_atoms[:i])
        out[sel] = v
        done |= sel
    return out


# ----------------------------------------------------------------------------
# R8: densities and per-frame normalisation (host side in the product too)
# ----------------------------------------------------------------------------


def shell_volume(bin_size, nbins):
    """structural/rdf_cn.py:312-318."""
    return (
        4 / 3 * np.pi * bin_size ** 3
        * (np.arange(1, nbins + 1) ** 3 - np.arange(nbins) ** 3)
    )


def normalize_rdf(hist_full, hist_part, n_ref, volume, ref_counts, obj_counts,
                  partial_relations, bin_size):
    """
    structural/rdf_cn.py:288-291,319-328: g_full = h / (N * rho * shell),
    g_part[kl] = h / (N_a * rho_b * shell), rho_b = count_b / V.
    hist_full may be None (molecular variant). n_ref = number of objects.
    """
    nbins = hist_part.shape[1]
    sv = shell_volume(bin_size, nbins)
    g_full = None
    if hist_full is not None:
        rho = n_ref / volume
        g_full = hist_full / (n_ref * rho * sv)
    n_a = np.array([ref_counts[a] for a in partial_relations[0]], dtype=np.float64)
    rho_b = np.array([obj_counts[b] / volume for b in partial_relations[1]])
    g_part = hist_part / (n_a[:, None] * rho_b[:, None] * sv[None, :])
    return g_full, g_part


# ----------------------------------------------------------------------------
# M1, M2, drift: mean-square displacement
# ----------------------------------------------------------------------------


def msd_single_origin(r, origin=0):
    """
    dynamical/diffusion.py:212-217: per entity, per frame (r(t) - r(origin))**2
    per axis, msd = (dx2 + dy2) + dz2. r [F,E,3] -> [F,E,4].
    """
    d2 = (r - r[origin][None]) ** 2
    return np.concatenate([d2, d2.sum(axis=2, keepdims=True)], axis=2)


def msd_group_mean(msd_all, group_offsets):
    """dynamical/diffusion.py:218: mean over entities of each group. -> [F,G,4]."""
    out = []
    for g in range(len(group_offsets) - 1):
        lo, hi = group_offsets[g], group_offsets[g + 1]
        out.append(msd_all[:, lo:hi, :].mean(axis=1))
    return np.stack(out, axis=1)


def msd_fixed_lag(r, tao):
    """
    dynamical/diffusion.py:225-237: keep frames [::tao]; per entity and axis
    (x_k - x_{k-1})**2 over consecutive kept frames; the three axis columns
    are means over the n-1 windows, but the "msd" column is the sum over
    windows divided by n (the NaN first row sums to 0 before the mean).
    r [F,E,3] -> [E,4].
    """
    kept = r[::tao]
    n = kept.shape[0]
    d2 = (kept[1:] - kept[:-1]) ** 2
    axes = d2.mean(axis=0)
    tot = d2.sum(axis=2).sum(axis=0) / n
    return np.concatenate([axes, tot[:, None]], axis=1)


def remove_type_drift(r, ent_mass, group_offsets):
    """
    dynamical/diffusion.py:83-96: per group (molecule type) mass-weighted COM
    at each frame minus the same at frame 0, subtracted from every entity of
    that group. r [F,E,3], ent_mass [E].
    """
    out = r.copy()
    for g in range(len(group_offsets) - 1):
        lo, hi = group_offsets[g], group_offsets[g + 1]
        m = ent_mass[lo:hi]
        com = np.einsum("e,fek->fk", m, r[:, lo:hi, :]) / m.sum()
        out[:, lo:hi, :] -= (com - com[0])[:, None, :]
    return out


def lag_msd_full(r, max_lag):
    """
    Superset (not in the reference): msd[lag] = mean over t0 and entities of
    |r(t0+lag) - r(t0)|**2, per axis and total. r [F,E,3] -> [max_lag+1,4].
    """
    F = r.shape[0]
    out = np.zeros((max_lag + 1, 4))
    for lag in range(1, max_lag + 1):
        d2 = (r[lag:] - r[: F - lag]) ** 2
        out[lag, :3] = d2.mean(axis=(0, 1))
        out[lag, 3] = d2.sum(axis=2).mean()
    return out


# ----------------------------------------------------------------------------
# M4: OLS through the origin
# ----------------------------------------------------------------------------


def ols_origin(t, y):
    """
    dynamical/diffusion.py:323-329 uses statsmodels OLS without intercept;
    closed form: slope = Sxy/Sxx, bse = sqrt(RSS/(n-1)/Sxx), R2 = 1 - RSS/Syy.
    """
    t = np.asarray(t, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    sxx = float(t @ t)
    slope = float(t @ y) / sxx
    res = y - slope * t
    rss = float(res @ res)
    return slope, np.sqrt(rss / (len(t) - 1) / sxx), 1.0 - rss / float(y @ y)


# ----------------------------------------------------------------------------
# G1-G4: charge flux, correlation, running integral
# ----------------------------------------------------------------------------


def charge_flux(vel, charge, atom_mass, seg_offsets, seg_type, n_types,
                vel_conv, charge_conv):
    """
    dynamical/_conductivity.py:11-35: COM velocity per molecule (calc_com),
    SI conversion of v and q, then J[k, type] = sum_mol q_mol * v_com,k.
    vel [N,3] -> [3, n_types].
    """
    vcom, _, q = calc_com(vel, atom_mass, seg_offsets, charge)
    vcom = vcom * vel_conv
    q = q * charge_conv
    out = np.zeros((3, n_types))
    for t in range(n_types):
        sel = seg_type == t + 1
        for k in range(3):
            out[k, t] = np.dot(vcom[sel, k], q[sel])
    return out


def xcorr_fft(a, b):
    """
    dynamical/conductivity.py:109-114 (correlate) and viscosity.py:111-115:
    c[k] = sum_t a[t+k]*b[t] / (n-k) via zero-padded length-2n FFT.
    """
    n = len(a)
    al = np.concatenate((a, np.zeros(n)))
    bl = np.concatenate((b, np.zeros(n)))
    c = np.fft.ifft(np.fft.fft(al) * np.conjugate(np.fft.fft(bl))).real
    return c[:n] / (np.arange(n) + 1)[::-1]


def xcorr_direct(a, b):
    """
    dynamical/viscosity.py:103-108 ("brute_force"): the same estimator by
    direct summation, c[k] = sum_{t<n-k} a[t+k]*b[t] / (n-k).
    """
    n = len(a)
    full = np.correlate(a, b, "full")
    return full[n - 1:] / np.arange(n, 0, -1, dtype="float")


def cumtrapz(y, dx, leading_zero=False):
    """
    dynamical/viscosity.py:151, conductivity.py:231: I[k] = sum_{m<k}
    (y[m]+y[m+1])/2*dx, sequential order. n-1 points, or n with I[0]=0.
    """
    inc = dx * (y[1:] + y[:-1]) / 2.0
    out = np.cumsum(inc)
    return np.concatenate(([0.0], out)) if leading_zero else out


# ----------------------------------------------------------------------------
# neighbour-shell residence autocorrelation
# ----------------------------------------------------------------------------


def shell_indicator(k_xyz, l_xyz, lengths, rc2_lo, rc2_hi, same):
    """
    dynamical/residence_time.py:100-106 for one frame: h[i][j] = (rsq > lo^2) & (rsq <= hi^2) with the
    reference's single-wrap rsq (rdf_cn.py:44-57, called with num_of_ids=0); for a relation of a type with
    itself the diagonal is cleared (residence_time.py:103-104). k_xyz [Nk,3], l_xyz [Nl,3] -> bool [Nk,Nl].
    """
    rsq = min_image_rsq(k_xyz[:, None, :], l_xyz[None, :, :], lengths)
    h = (rsq > rc2_lo) & (rsq <= rc2_hi)
    if same:
        np.fill_diagonal(h, False)
    return h


def residence_counts(h_frames):
    """
    Exact numerators of residence_time.py:124-131: counts[k] = sum over (i, j) and t of h(t) h(t+k), the
    quantity `acovf(column, demean=False, unbiased=True)` estimates times (n - k). h_frames: bool [F,Nk,Nl].
    """
    h = np.asarray(h_frames, dtype=np.int64)
    n = h.shape[0]
    return np.array([int((h[: n - k] * h[k:]).sum()) for k in range(n)], dtype=np.int64)


def residence_autocorr(h_frames):
    """
    residence_time.py:111-136: mean over all (central atom, column) series of the unbiased autocovariance,
    normalised by its lag-0 value. In exact arithmetic: c[k] = counts[k] / (n - k) / (Nk * Nl); corr = c / c[0].
    """
    counts = residence_counts(h_frames).astype(np.float64)
    n, nk, nl = np.asarray(h_frames).shape
    c = counts / (n - np.arange(n)) / float(nk * nl)
    return c / c[0]
