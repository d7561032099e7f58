"""
GPU parity tests proper: every kernel is called through the C-ABI (ctypes) and compared with the
oracle and with the golden vectors recorded from the real reference.
Integer work (histograms, counts) must be bit-exact; floating point within the stated tolerance.
"""
import numpy as np
import pytest

from conftest import sorted_frame
from oracle import cpu_ref as O
from oracle import cref as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from mdproptools_amd import backend

    return backend


def soa(data):
    """[N,4] = [type,x,y,z] -> (xyz [1,3,N], types [N])."""
    return np.ascontiguousarray(data[:, 1:4].T)[None], data[:, 0].astype(np.int32)


# ------------------------------------------------------------------ R3 / R4: atom-atom
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rdf_cn_golden_synth(B, g_synth, tag):
    g = g_synth
    data, L, rel = g[tag + "_data"], g[tag + "_lengths"], g[tag + "_rel"]
    r_cut, ddr = float(g[tag + "_r_cut"]), float(g[tag + "_ddr"])
    nb = int(r_cut / ddr)
    xyz, ty = soa(data)
    full, part, ov = B.rdf_loop(xyz, ty, L[None], rel, r_cut, ddr, nb)
    assert ov == 0
    np.testing.assert_array_equal(full[0].astype(np.int64), g[tag + "_full"])
    np.testing.assert_array_equal(part[0].astype(np.int64), g[tag + "_part"])
    cn = B.cn_loop(xyz, ty, L[None], rel, list(g[tag + "_cn_cut"]))
    np.testing.assert_array_equal(cn[0].astype(np.int64), g[tag + "_cn"])


def test_rdf_cn_golden_c1(B, g_c1):
    g = g_c1
    frames = [sorted_frame(fr) for fr in g["frames"]]
    xyz = np.stack([np.ascontiguousarray(fr[:, 2:5].T) for fr in frames])
    box = g["bounds"][:, :, 1] - g["bounds"][:, :, 0]
    ty = frames[0][:, 1].astype(np.int32)
    full, part, ov = B.rdf_loop(xyz, ty, box, g["rdf_def_rel"].T, 20, 0.05, 400)
    assert ov == 0
    # (round 6: the example's nine types / five relations run on the table-free packed sweep through displaced rows —
    # 15 rows instead of 36 — not on the class-row kernel `<5, .>`)
    assert "<3," in B.default_context().last_kernel_name(), B.default_context().last_kernel_name()
    np.testing.assert_array_equal(full.astype(np.int64), g["rdf_def_full"])
    np.testing.assert_array_equal(part.astype(np.int64), g["rdf_def_part"])
    assert int(full[0].sum()) == 30926986  # SURVEY.md known answer
    cn = B.cn_loop(xyz, ty, box, g["cn_def_rel"].T, list(g["cn_def_cut"]))
    np.testing.assert_array_equal(cn.astype(np.int64), g["cn_def_raw"])
    # altered ids: 32 pseudo-types, 2 relations (class table 32x32)
    alt = O.calc_atom_type(frames[0][:, 0], g["num_mols"], g["num_atoms_per_mol"]).astype(np.int32)
    full, part, ov = B.rdf_loop(xyz, alt, box, g["rdf_alt_rel"].T, 20, 0.05, 400)
    np.testing.assert_array_equal(full.astype(np.int64), g["rdf_alt_full"])
    np.testing.assert_array_equal(part.astype(np.int64), g["rdf_alt_part"])
    cn = B.cn_loop(xyz, alt, box, g["rdf_alt_rel"].T, list(g["cn_alt_cut"]))
    np.testing.assert_array_equal(cn.astype(np.int64), g["cn_alt_raw"])
    # frame-summed output equals the sum of the per-frame outputs
    fsum, psum, _ = B.rdf_loop(xyz, ty, box, g["rdf_def_rel"].T, 20, 0.05, 400, per_frame=False)
    np.testing.assert_array_equal(fsum.astype(np.int64), g["rdf_def_full"].sum(axis=0))
    np.testing.assert_array_equal(psum.astype(np.int64), g["rdf_def_part"].sum(axis=0))


def _random_case(rng, n, n_types, L, stray=True):
    xyz = rng.uniform(0, 1, (3, n)) * np.asarray(L)[:, None] + 1.75
    if stray and n > 8:
        idx = rng.choice(n, n // 10, replace=False)
        xyz[:, idx] += rng.integers(-2, 3, (3, len(idx))) * np.asarray(L)[:, None]
    ty = rng.integers(1, n_types + 1, n).astype(np.int32)
    return xyz, ty


@pytest.mark.parametrize("n", [2, 3, 63, 255, 256, 257, 511, 513, 1025])
def test_rdf_ragged_sizes_vs_c_oracle(B, n):
    rng = np.random.default_rng(100 + n)
    L = np.array([17.0, 19.5, 18.25])
    xyz, ty = _random_case(rng, n, 3, L)
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3], [3, 1], [2, 2]])
    r_cut, ddr = 8.0, 0.04
    nb = int(r_cut / ddr)
    full, part, ov = B.rdf_loop(xyz[None], ty, L[None], rel, r_cut, ddr, nb)
    cf, cp, cov = C.rdf_pairs(xyz, ty, rel, L, r_cut * r_cut, ddr, nb)
    assert ov == cov
    np.testing.assert_array_equal(full[0], cf)
    np.testing.assert_array_equal(part[0], cp)
    cuts = [1.5, 2.5, 3.75, 8.0, 2.5, 0.9]
    cn = B.cn_loop(xyz[None], ty, L[None], rel, cuts)
    np.testing.assert_array_equal(cn[0], C.cn_pairs(xyz, ty, rel, L, [c * c for c in cuts]))


def test_rdf_empty_and_single(B):
    L = np.array([[10.0, 10.0, 10.0]])
    rel = np.array([[1, 1]])
    for n in (0, 1):
        xyz = np.zeros((1, 3, n))
        full, part, ov = B.rdf_loop(xyz, np.ones(n, np.int32), L, rel, 4.0, 0.1, 40)
        assert full.sum() == 0 and part.sum() == 0 and ov == 0
    full, part, ov = B.rdf_loop(np.zeros((0, 3, 5)), np.ones(5, np.int32), np.zeros((0, 3)), rel, 4.0, 0.1, 40)
    assert full.shape == (0, 40)


def test_rdf_geometry_independence(B):
    """Integer results must not depend on launch geometry: j-splits, replica slots, variant."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(5)
    L = np.array([30.0, 30.0, 30.0])
    F, n = 21, 1500  # 21 frames: frame groups of 8 XCD shares with ragged ends
    xyz = np.stack([_random_case(rng, n, 4, L)[0] for _ in range(F)])
    ty = (1 + np.arange(n) % 4).astype(np.int32)
    rel = np.array([(a, b) for a in range(1, 5) for b in range(a, 5)])
    box = np.tile(L, (F, 1))
    ref = None
    for jsplit, slots, variant, fpb, batch in [(0, 16, 0, 0, 0), (1, 1, 1, 1, 0), (2, 3, 0, 0, 4), (3, 8, 1, 3, 0),
                                               (0, 16, 1, 0, 5), (1, 2, 1, 64, 0)]:
        ctx = Context(0)
        ctx.set_option("rdf_jsplit", jsplit)
        ctx.set_option("rdf_slots", slots)
        ctx.set_option("rdf_variant", variant)
        ctx.set_option("rdf_fpb", fpb)
        ctx.set_option("rdf_batch", batch)
        ctx.set_option("rdf_cull", 0)
        for per_frame in (True, False):
            full, part, ov = B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 240, per_frame=per_frame, ctx=ctx)
            if per_frame:
                full, part = full.sum(axis=0), part.sum(axis=0)
            if ref is None:
                ref = (full, part)
                cf = sum(C.rdf_pairs(xyz[f], ty, rel, L, 144.0, 0.05, 240)[0] for f in range(F))
                np.testing.assert_array_equal(full, cf)
            np.testing.assert_array_equal(full, ref[0])
            np.testing.assert_array_equal(part, ref[1])
        ctx.close()


def test_rdf_overflow_bin_is_dropped_and_counted(B):
    """a pair with rsq just below 20**2 lands in bin 400 == nbins."""
    r = np.sqrt(np.nextafter(400.0, 0.0))
    xyz = np.array([[[1.0, 1.0 + r]], [[1.0, 1.0]], [[1.0, 1.0]]]).reshape(1, 3, 2)
    d = xyz[0, 0, 0] - xyz[0, 0, 1]
    rsq = d * d
    L = np.array([[100.0, 100.0, 100.0]])
    full, part, ov = B.rdf_loop(xyz, np.array([1, 1], np.int32), L, np.array([[1, 1]]), 20, 0.05, 400)
    if rsq < 400.0 and int(np.sqrt(rsq) / 0.05) == 400:
        assert ov == 1 and full.sum() == 0
    else:  # the constructed distance rounded elsewhere; it must then be counted normally
        assert ov == 0


def test_rdf_per_frame_types_and_varying_box(B):
    rng = np.random.default_rng(11)
    F, n = 3, 400
    boxes = np.array([[20.0, 21.0, 22.0], [20.5, 21.5, 22.5], [19.0, 23.0, 21.0]])
    xyz = np.stack([rng.uniform(0, 1, (3, n)) * boxes[f][:, None] for f in range(F)])
    ty = rng.integers(1, 4, (F, n)).astype(np.int32)
    rel = np.array([[1, 2], [3, 3], [2, 2]])
    full, part, ov = B.rdf_loop(xyz, ty, boxes, rel, 9.0, 0.1, 90)
    for f in range(F):
        cf, cp, _ = C.rdf_pairs(xyz[f], ty[f], rel, boxes[f], 81.0, 0.1, 90)
        np.testing.assert_array_equal(full[f], cf)
        np.testing.assert_array_equal(part[f], cp)


def test_rdf_many_relations_multi_pass(B):
    """45 relations x 2000 bins do not fit LDS in one pass: class rows are processed in passes."""
    rng = np.random.default_rng(12)
    n, L = 900, np.array([24.0, 24.0, 24.0])
    xyz = rng.uniform(0, 1, (3, n)) * L[:, None]
    ty = rng.integers(1, 10, n).astype(np.int32)
    rel = np.array([(a, b) for a in range(1, 10) for b in range(a, 10)])
    full, part, ov = B.rdf_loop(xyz[None], ty, L[None], rel, 10.0, 0.005, 2000)
    cf, cp, cov = C.rdf_pairs(xyz, ty, rel, L, 100.0, 0.005, 2000)
    assert ov == cov
    np.testing.assert_array_equal(full[0], cf)
    np.testing.assert_array_equal(part[0], cp)


# ------------------------------------------------------------------ R5 / R6: atoms x COM sites
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_rdf_cn_mol_golden_synth(B, g_synth, tag):
    g = g_synth
    data, mol, L, rel = g[tag + "_data"], g[tag + "_mol"], g[tag + "_lengths"], g[tag + "_mol_rel"]
    r_cut, ddr = float(g[tag + "_r_cut"]), float(g[tag + "_ddr"])
    nb = int(r_cut / ddr)
    xyz, ty = soa(data)
    sxyz, st = soa(mol)
    part, ov = B.rdf_mol_loop(xyz, ty, sxyz, st, L[None], rel, r_cut, ddr, nb)
    assert ov == 0
    np.testing.assert_array_equal(part[0].astype(np.int64), g[tag + "_mol_part"])
    cn = B.cn_mol_loop(xyz, ty, sxyz, st, L[None], rel, list(g[tag + "_mol_cn_cut"]))
    np.testing.assert_array_equal(cn[0].astype(np.int64), g[tag + "_mol_cn"])


def test_molecular_c1_com_and_hist(B, g_c1):
    g = g_c1
    fr = sorted_frame(g["frames"][0])
    L = (g["bounds"][0][:, 1] - g["bounds"][0][:, 0])[None]
    _, _, off, seg_type = O.molecule_layout(g["num_mols"], g["num_atoms_per_mol"])
    amass = g["mass"][fr[:, 1].astype(np.int64) - 1]
    xyz = np.ascontiguousarray(fr[:, 2:5].T)[None]
    com, seg_mass, _ = B.segment_com(xyz, amass, off)
    ref_com = g["mol_com"][0]
    np.testing.assert_allclose(com[0].T, ref_com[:, 1:4], rtol=1e-13, atol=0)
    ty = fr[:, 1].astype(np.int32)
    # integer parity given the reference's own COM sites ...
    sites = np.ascontiguousarray(ref_com[:, 1:4].T)[None]
    part, ov = B.rdf_mol_loop(xyz, ty, sites, seg_type.astype(np.int32), L, g["mol_rel"].T, 20, 0.05, 400)
    np.testing.assert_array_equal(part[0].astype(np.int64), g["mol_rdf_part"][0])
    cn = B.cn_mol_loop(xyz, ty, sites, seg_type.astype(np.int32), L, g["mol_rel"].T, list(g["mol_cn_cut"]))
    np.testing.assert_array_equal(cn[0].astype(np.int64), g["mol_cn_raw"][0])
    # ... and with the device-computed COM (sums in another order: equal here, documented as rtol 1e-13 on COM)
    part2, _ = B.rdf_mol_loop(xyz, ty, com, seg_type.astype(np.int32), L, g["mol_rel"].T, 20, 0.05, 400)
    assert np.abs(part2[0].astype(np.int64) - g["mol_rdf_part"][0]).sum() <= 2


def test_rect_ragged_vs_c_oracle(B):
    rng = np.random.default_rng(21)
    L = np.array([15.0, 16.0, 17.0])
    for n, m in [(1, 1), (300, 7), (257, 513), (700, 256)]:
        xyz, ty = _random_case(rng, n, 3, L)
        sx, st = _random_case(rng, m, 2, L, stray=False)
        rel = np.array([[1, 1], [2, 2], [3, 1], [1, 2], [1, 1]])
        part, ov = B.rdf_mol_loop(xyz[None], ty, sx[None], st, L[None], rel, 7.0, 0.07, 100)
        cp, cov = C.rdf_rect(xyz, ty, sx, st, rel, L, 49.0, 0.07, 100)
        assert ov == cov
        np.testing.assert_array_equal(part[0], cp)
        cuts = [2.0, 3.0, 6.5, 1.0, 4.0]
        cn = B.cn_mol_loop(xyz[None], ty, sx[None], st, L[None], rel, cuts)
        np.testing.assert_array_equal(cn[0], C.cn_rect(xyz, ty, sx, st, rel, L, [c * c for c in cuts]))


# ------------------------------------------------------------------ M1-M3, G1 on the 1146-atom golden
def _small(g):
    cols = list(g["columns"])
    fr = np.stack([sorted_frame(f, cols.index("id")) for f in g["frames"]])
    pick = lambda names: np.ascontiguousarray(  # noqa: E731
        fr[:, :, [cols.index(c) for c in names]].transpose(0, 2, 1))
    return cols, fr, pick


def test_msd_allatom_golden(B, g_small):
    g = g_small
    cols, fr, pick = _small(g)
    r = pick(("xu", "yu", "zu"))
    F, _, E = r.shape
    pairs = [(0, t) for t in range(F)]
    sums, pe = B.msd_pairs(r, pairs, [0, E], scale=1e-10, per_entity=True)
    np.testing.assert_allclose(pe.reshape(F * E, 4), g["aa_msd_all"][:, 2:6], rtol=1e-12, atol=0)
    np.testing.assert_allclose(sums[:, 0, :] / E, g["aa_msd"][:, 1:5], rtol=1e-10, atol=0)
    win = B.msd_windows(r, 4, scale=1e-10)
    n_kept = len(range(0, F, 4))
    expect = g["aa_msd_int"][:, 1:5]
    np.testing.assert_allclose(win[:, :3] / (n_kept - 1), expect[:, :3], rtol=1e-10, atol=0)
    np.testing.assert_allclose(win[:, 3] / n_kept, expect[:, 3], rtol=1e-10, atol=0)


@pytest.mark.parametrize("E", [1, 777, 4096])
def test_msd_pairs_column_layout_equals_row_layout(B, E):
    """mdhip_msd_pairs_cols (the four per-entity values as column blocks, here rows of a wider block) stores the same
    doubles as the row layout of mdhip_msd_pairs, with the same sums; odd and even entity counts, groups."""
    rng = np.random.default_rng(E)
    F = 6
    r = rng.normal(0, 20, (F, 3, E))
    pairs = [(0, t) for t in range(F)] + [(3, 1)]
    goff = [0, E] if E < 10 else [0, E // 3, E]
    sums, pe = B.msd_pairs(r, pairs, goff, scale=1e-10, per_entity=True)
    block = np.full((6, len(pairs) * E + 5), -1.0)
    cols = block[1:5, :len(pairs) * E]
    sums_c = B.msd_pairs_cols(r, pairs, goff, cols, scale=1e-10)
    np.testing.assert_array_equal(sums_c, sums)
    np.testing.assert_array_equal(cols, pe.reshape(-1, 4).T)
    assert (block[0] == -1).all() and (block[5] == -1).all() and (block[:, len(pairs) * E:] == -1).all()
    tight = np.empty((4, len(pairs) * E))
    B.msd_pairs_cols(r, pairs, goff, tight, scale=1e-10)
    np.testing.assert_array_equal(tight, cols)
    with pytest.raises(ValueError):
        B.msd_pairs_cols(r, pairs, goff, np.empty((4, 3)), scale=1.0)
    # device-resident columns with a stride (the C-ABI's cols_on_device form), device-resident input
    import ctypes as C

    import torch

    from mdproptools_amd._lib import default_context, ptr

    ctx = default_context()
    n_col = len(pairs) * E
    dcols = torch.full((4, n_col + 3), -1.0, dtype=torch.float64, device="cuda")
    rd = torch.from_numpy(r).cuda()
    pr = np.asarray(pairs, dtype=np.int32)
    go = np.asarray(goff, dtype=np.int64)
    sums_d = np.zeros((len(pairs), len(goff) - 1, 4))
    ctx.check(ctx.lib.mdhip_msd_pairs_cols(ctx.h, F, E, C.c_void_p(rd.data_ptr()), 1, 1e-10, len(pairs), ptr(pr, C.c_int32),
                                           len(goff) - 1, ptr(go, C.c_int64), ptr(sums_d), C.c_void_p(dcols.data_ptr()),
                                           n_col + 3, 1))
    np.testing.assert_array_equal(sums_d, sums)
    got = dcols.cpu().numpy()
    np.testing.assert_array_equal(got[:, :n_col], cols)
    assert (got[:, n_col:] == -1).all()


def test_segment_com_ragged_and_long_segments(B):
    """Staged kernel (runs of whole segments through LDS) and the per-lane fallback (a segment longer than
    the LDS stage) against the oracle, on ragged segment sizes 1..40, 4-atom and 16-atom molecules."""
    rng = np.random.default_rng(9)
    for sizes in (rng.integers(1, 41, 700), np.full(900, 4), np.full(300, 16),
                  np.concatenate([rng.integers(1, 9, 50), [3000], rng.integers(1, 9, 50)])):
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        n = int(off[-1])
        F = 5
        attr = rng.normal(0, 30, (F, 3, n))
        mass = rng.uniform(1, 40, n)
        q = rng.normal(0, 1, n)
        com, seg_mass, seg_q = B.segment_com(attr, mass, off, atom_q=q)
        want = np.add.reduceat(attr * mass, off[:-1], axis=2) / np.add.reduceat(mass, off[:-1])
        np.testing.assert_allclose(com, want, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(seg_mass, np.add.reduceat(mass, off[:-1]), rtol=1e-14)
        np.testing.assert_allclose(seg_q, np.add.reduceat(q, off[:-1]), rtol=1e-12, atol=1e-14)
        M = len(sizes)
        st = (np.arange(M) * 3 // M).astype(np.int32)
        j = B.charge_flux(attr, mass, q, off, st, 3, 1e5, 1.6e-19)
        jm = (want * 1e5) * (np.add.reduceat(q, off[:-1]) * 1.6e-19)  # [F,3,M]
        jw = np.stack([np.stack([jm[:, k, st == t].sum(axis=1) for t in range(3)]) for k in range(3)])
        np.testing.assert_allclose(j, jw, rtol=1e-9, atol=1e-25)


def test_segment_kernels_agree_for_any_plane_count(B):
    """The one-(run, frame)-per-block kernel (seg_frame 1, default) and the software-pipelined staged kernel (0) do the
    same products and additions in the same order: bit-identical centres and fluxes, for 1 .. 7 attribute planes (plane
    groups of three, the last one partial), odd and even atom counts (the 16-byte load path needs even ones), many
    frames (more steps than one grid row holds is not reachable here; the loop is covered by the frame count)."""
    ctx = B.default_context()
    rng = np.random.default_rng(19)
    try:
        for sizes in (rng.integers(1, 30, 400), np.full(257, 4), np.concatenate([np.full(130, 16), [3]])):
            off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            n, M = int(off[-1]), len(sizes)
            mass = rng.uniform(1, 40, n)
            q = rng.normal(0, 1, n)
            st = (np.arange(M) * 2 // M).astype(np.int32)
            for K in (1, 2, 3, 4, 7):
                attr = rng.normal(0, 30, (9, K, n))
                got = {}
                for mode in (1, 0):
                    ctx.set_option("seg_frame", mode)
                    com, _, _ = B.segment_com(attr, mass, off)
                    assert ctx.last_kernel_name().startswith("segment_frame_kernel" if mode else "segment_staged_kernel")
                    got[mode] = com
                np.testing.assert_array_equal(got[1], got[0])
                want = np.add.reduceat(attr * mass, off[:-1], axis=2) / np.add.reduceat(mass, off[:-1])
                np.testing.assert_allclose(got[1], want, rtol=1e-13, atol=1e-13)
            vel = rng.normal(0, 5, (9, 3, n))
            flux = {}
            for mode in (1, 0):
                ctx.set_option("seg_frame", mode)
                flux[mode] = B.charge_flux(vel, mass, q, off, st, 2, 1e5, 1.6e-19)
            np.testing.assert_array_equal(flux[1], flux[0])
    finally:
        ctx.set_option("seg_frame", 1)


def test_com_msd_and_charge_flux_golden(B, g_small):
    g = g_small
    cols, fr, pick = _small(g)
    nm, na = g["num_mols"], g["num_atoms_per_mol"]
    _, _, off, seg_type = O.molecule_layout(nm, na)
    amass = g["mass"][fr[0][:, cols.index("type")].astype(np.int64) - 1]
    com, seg_mass, _ = B.segment_com(pick(("xu", "yu", "zu")), amass, off)
    ref = g["calc_com_xu"]
    np.testing.assert_allclose(com[0].T, ref[:, 2:5], rtol=1e-13)
    np.testing.assert_allclose(seg_mass, ref[:, 5], rtol=1e-14)
    F, _, M = com.shape
    goff = np.concatenate([[0], np.cumsum(nm)])
    sums, pe = B.msd_pairs(com, [(0, t) for t in range(F)], goff, scale=1e-10, per_entity=True)
    np.testing.assert_allclose(pe.reshape(F * M, 4), g["com_msd_all"][:, 3:7], rtol=1e-9, atol=0)
    mean = sums / np.diff(goff)[None, :, None]
    np.testing.assert_allclose(mean.reshape(F, -1), g["com_msd"][:, 1:], rtol=1e-9, atol=0)
    q = fr[0][:, cols.index("q")]
    j = B.charge_flux(pick(("vx", "vy", "vz")), amass, q, off, (seg_type - 1).astype(np.int32), 3,
                      10 ** -10 / 10 ** -15, 1.602176634 * 10 ** -19)
    np.testing.assert_allclose(j, g["cond_j"], rtol=1e-9, atol=1e-25)


def test_msd_ragged_groups_vs_oracle(B):
    rng = np.random.default_rng(31)
    F, E = 7, 2500
    r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(0, 50, (1, 3, E))
    goff = [0, 1, 1030, 1030, 2500]  # a single-entity group, an empty group, ragged chunk ends
    pairs = [(0, 0), (0, 6), (2, 5), (6, 1)]
    sums = B.msd_pairs(r, pairs, goff, scale=1e-10)
    expect = C.msd_pairs(r * 1e-10, pairs, goff)
    np.testing.assert_allclose(sums, expect, rtol=1e-12, atol=0)
    win = B.msd_windows(r, 3, scale=1e-10)
    kept = (r * 1e-10)[::3]
    d2 = (kept[1:] - kept[:-1]) ** 2
    np.testing.assert_allclose(win[:, :3], d2.sum(axis=0).T, rtol=1e-12)
    np.testing.assert_allclose(win[:, 3], d2.sum(axis=1).sum(axis=0), rtol=1e-12)


def test_lag_msd_vs_oracle(B):
    rng = np.random.default_rng(32)
    for F, E, goff in [(40, 70, [0, 30, 70]), (300, 5, [0, 5]), (2100, 3, [0, 1, 3])]:
        r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(0, 50, (1, 3, E))
        out = B.lag_msd(r, F - 1, goff, scale=1.0)
        for g in range(len(goff) - 1):
            sub = r[:, :, goff[g]:goff[g + 1]].transpose(0, 2, 1)
            expect = O.lag_msd_full(sub, F - 1)
            np.testing.assert_allclose(out[:, g, :], expect, rtol=1e-10, atol=1e-12)


def test_lag_msd_fft_variant(B):
    """The autocorrelation-theorem path (lag_variant 2) against the oracle and the difference kernel: random
    walks far from the origin with ragged / empty / single-entity groups, a scale factor, max_lag < F - 1;
    agreement within the bound the library reports (and within 1e-9 here). Variant 3 hands a ballistic
    trajectory, whose bound is too loose, back to the difference kernel."""
    ctx = B.default_context()
    rng = np.random.default_rng(33)
    try:
        for F, E, goff, max_lag in [(40, 70, [0, 30, 30, 70], 39), (300, 5, [0, 5], 299), (2100, 3, [0, 1, 3], 2099),
                                    (1000, 400, [0, 150, 400], 250), (17, 1, [0, 1], 16), (5, 3, [0, 3], 4),
                                    (2, 2, [0, 2], 1), (100, 9, [0, 9], 0), (6000, 2, [0, 2], 5999),
                                    (9000, 2, [0, 1, 2], 7000), (700, 20, [0, 20], 300), (130, 6, [0, 6], 120)]:
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-500, 500, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, max_lag, goff, scale=0.5)
            assert ctx.last_rel_bound() == 0.0
            for variant in (2, 4):   # fused LDS transform / batched global transforms (fft_pow2.hip)
                ctx.set_option("lag_variant", variant)
                fft = B.lag_msd(r, max_lag, goff, scale=0.5)
                bound = ctx.last_rel_bound()
                assert (ctx.last_kernel_name() == "lag_msd_fft") == (variant == 4)
                assert ctx.last_kernel_name() in ("lag_msd_fft", "msd_power_lds_kernel", "msd_power_w12_kernel", "msd_power_w1_kernel") and (bound > 0.0) == (max_lag > 0) and bound < 1e-9, (F, bound)
                assert fft.shape == exact.shape and (fft[0] == 0.0).all()
                nz = exact > 0
                rel = np.abs(fft[nz] - exact[nz]) / exact[nz]
                assert not nz.any() or rel.max() <= max(bound, 1e-13), (F, rel.max(), bound)
                assert (fft[~nz] == 0.0).all()          # the empty group and lag 0
                for g in range(len(goff) - 1):
                    if goff[g + 1] > goff[g]:
                        sub = 0.5 * r[:, :, goff[g]:goff[g + 1]].transpose(0, 2, 1)
                        np.testing.assert_allclose(fft[:, g, :], O.lag_msd_full(sub, max_lag), rtol=1e-9, atol=1e-12)
        # ballistic: <x^2> / MSD(1) ~ F^2, the bound exceeds 1e-10 and variant 3 falls back
        v = rng.normal(0, 1, (3, 50))
        traj = rng.uniform(0, 50, (1, 3, 50)) + v[None] * np.arange(4000)[:, None, None]
        ctx.set_option("lag_variant", 2)
        B.lag_msd(traj, 3999, [0, 50])
        assert ctx.last_rel_bound() > 1e-10
        ctx.set_option("lag_variant", 3)
        out3 = B.lag_msd(traj, 3999, [0, 50])
        b3, name3 = ctx.last_rel_bound(), ctx.last_kernel_name()
        ctx.set_option("lag_variant", 1)
        exact3 = B.lag_msd(traj, 3999, [0, 50])
        if name3.startswith("lag_msd_"):  # the whole call from the difference kernel
            assert b3 == 0.0
            np.testing.assert_array_equal(out3, exact3)
        else:  # (round 6) MSD ~ k^2: only the first and last lags miss the bound — those from the difference form
            assert "lag_low_lags_kernel" in name3 and 0.0 < b3 <= 1e-10, (name3, b3)
            nz3 = exact3 > 0
            assert (np.abs(out3[nz3] - exact3[nz3]) / exact3[nz3]).max() <= b3
        # diffusive data: variant 3 keeps the FFT answer
        r = np.cumsum(rng.normal(0, 0.1, (500, 3, 64)), axis=0)
        ctx.set_option("lag_variant", 3)
        B.lag_msd(r, 499, [0, 64])
        assert ctx.last_kernel_name() == "msd_power_w1_kernel" and 0.0 < ctx.last_rel_bound() <= 1e-10
    finally:
        ctx.set_option("lag_variant", 1)


def test_lag_msd_fft_every_transform_size(B):
    """The fused kernels (lag_fft_kernel 3 = the round-5 kernel for padded length 12288 where it applies (F + max_lag in
    (8192, 12288]), else as 2 = first pass from registers + wave-private sub-transforms where that applies, 1 = block-wide
    passes, 0 = round-2 kernel) over series lengths on both sides of every transform size 2^9 .. 2^13 and 3 * 2^12, full and
    truncated lag ranges (F > N only happens with those), tiny and empty groups: each within the bound the library reports
    against the difference kernel (hence within twice that of each other)."""
    ctx = B.default_context()
    rng = np.random.default_rng(5)
    cases = [(F, F - 1) for F in (257, 511, 513, 1024, 1025, 2047, 2049, 3000, 4096, 4097, 5000, 5121, 6143, 6144, 6145,
                                  8191, 8192)]
    cases += [(9000, 7000), (12000, 4000), (16000, 300), (6000, 2100), (1000, 20), (6200, 6000), (8000, 4288), (4200, 4100)]
    cases += [(2050, 2049), (2500, 2499), (3071, 3070), (3072, 3071), (3073, 3072), (2600, 1800), (1536, 1535), (1537, 600),
              (1535, 1534), (1800, 400)]  # the 12288-point kernel's short range and its lower edge
    try:
        for F, max_lag in cases:
            E = int(rng.integers(3, 24))
            cut = int(rng.integers(0, E + 1))
            goff = [0, cut, cut, E]
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-500, 500, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, max_lag, goff, scale=0.5)
            ctx.set_option("lag_variant", 2)
            nz = exact > 0
            outs = []
            # round 6: one wave per series below the 12288-point kernel's range (F <= 1536, F + max_lag <= 3072; the fused
            # block-wide kernels this test walks stay behind `lag_w1` 0)
            ctx.set_option("lag_w1", 1)
            w1 = B.lag_msd(r, max_lag, goff, scale=0.5)
            w1_bound = ctx.last_rel_bound()
            takes_w1 = F <= 1536 and F + max_lag <= 3072 and (F < 1536 or F + max_lag <= 2048)
            assert (ctx.last_kernel_name() == "msd_power_w1_kernel") == takes_w1, (F, max_lag, ctx.last_kernel_name())
            assert (np.abs(w1[nz] - exact[nz]) / exact[nz]).max() <= w1_bound and (w1[0] == 0.0).all(), (F, max_lag, w1_bound)
            ctx.set_option("lag_w1", 0)
            for kern in (3, 2, 1, 0):
                ctx.set_option("lag_fft_kernel", kern)
                fft = B.lag_msd(r, max_lag, goff, scale=0.5)
                bound = ctx.last_rel_bound()
                # (round 6: also from 1536 frames on where F + max_lag lies in (2048, 8192] — the SHORT instance below 3072)
                w12 = kern == 3 and (F + 1) // 2 <= 3072 and ((8192 < F + max_lag <= 12288 and F >= 3072) or
                                                              (2048 < F + max_lag <= 8192 and F >= 1536))
                assert ctx.last_kernel_name() == ("msd_power_w12_kernel" if w12 else "msd_power_lds_kernel") and bound > 0.0
                assert (fft[0] == 0.0).all()
                rel = np.abs(fft[nz] - exact[nz]) / exact[nz]
                assert rel.max() <= bound, (F, max_lag, kern, rel.max(), bound)
                outs.append((fft, bound))
            for o, b in outs[1:]:
                assert (np.abs(o[nz] - outs[0][0][nz]) / exact[nz]).max() <= b + outs[0][1], (F, max_lag)
    finally:
        ctx.set_option("lag_variant", 1)
        ctx.set_option("lag_fft_kernel", 3)
        ctx.set_option("lag_w1", -1)


def test_lag_variant3_repairs_the_few_lags_that_miss_the_bound(B):
    """Round 6: the spectral path's error is the same absolute amount at every lag, so relative to the MSD it is largest at
    the first lags (small displacement) and the last ones (few origins). When the 1e-10 bound is missed at no more than 24
    lags per end, the default (lag_variant 3) recomputes THOSE lags from the difference form (lag_low_lags_kernel /
    lag_high_lags_kernel, csrc/msd.hip) and lets the rest of the spectral result stand, instead of handing the whole call
    to the O(F^2) difference kernel: long random walks through the residue-class and the batched paths, a walk riding on a
    slow oscillation through the fused kernel. The repaired rows equal the difference kernel's to 1e-12, every other row is
    the spectral path's bit for bit, the reported bound is that of the rows that stood (<= 1e-10); host, device and
    asynchronous results agree; `lag_ends` 0 restores the whole-call fallback; solid-like data still takes it."""
    import torch

    ctx = B.default_context()
    rng = np.random.default_rng(91)
    cases = []
    for F, E, goff in ((24576, 30, [0, 10, 30]), (30000, 20, [0, 20])):
        cases.append((np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E)), goff))
    F, E = 5000, 60
    t = np.arange(F)[:, None, None]
    for amp in (9.0, 17.0):  # (one lag per end misses the bound / about ten do)
        cases.append((np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + amp * np.sin(2 * np.pi * t / F + rng.uniform(0, 6.28, (1, 3, E))),
                      [0, 25, 60]))
    try:
        for r, goff in cases:
            F = r.shape[0]
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, F - 1, goff)
            ctx.set_option("lag_variant", 2)
            spec = B.lag_msd(r, F - 1, goff)
            assert ctx.last_rel_bound() > 1e-10, (F, ctx.last_rel_bound())  # (else the case tests nothing)
            ctx.set_option("lag_variant", 3)
            got = B.lag_msd(r, F - 1, goff)
            bound = ctx.last_rel_bound()
            assert "lag_low_lags_kernel" in ctx.last_kernel_name() and 0.0 < bound <= 1e-10, (F, ctx.last_kernel_name(), bound)
            changed = np.where(np.any(got != spec, axis=(1, 2)))[0]
            assert len(changed) and all(k <= 24 or k >= F - 24 for k in changed), (F, changed)
            np.testing.assert_allclose(got[changed], exact[changed], rtol=1e-12)
            nz = exact > 0
            assert (np.abs(got[nz] - exact[nz]) / exact[nz]).max() <= bound, (F, bound)
            out = torch.empty((F, len(goff) - 1, 4), dtype=torch.float64, device="cuda")
            B.lag_msd(torch.from_numpy(r).cuda(), F - 1, goff, out=out, async_=True).wait()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(out.cpu().numpy(), got)
            ctx.set_option("lag_ends", 0)
            whole = B.lag_msd(r, F - 1, goff)
            assert ctx.last_kernel_name().startswith("lag_msd_") and ctx.last_rel_bound() == 0.0
            np.testing.assert_array_equal(whole, exact)
            ctx.set_option("lag_ends", -1)
        # solid-like motion (an oscillation: the MSD returns to ~0 at every multiple of the period): the bound fails at lags
        # all over the range — the whole call goes to the difference kernel, as before
        F, E = 3000, 12
        r = 0.3 * np.sin(2 * np.pi * np.arange(F)[:, None, None] / 50.0 + rng.uniform(0, 6.28, (1, 3, E))) + rng.uniform(-5, 5, (1, 3, E))
        ctx.set_option("lag_variant", 3)
        got = B.lag_msd(r, F - 1, [0, E])
        assert ctx.last_kernel_name().startswith("lag_msd_") and ctx.last_rel_bound() == 0.0
    finally:
        ctx.set_option("lag_variant", 1)
        ctx.set_option("lag_ends", -1)


def test_lag_msd_short_series_one_wave_per_series(B):
    """Round 6: trajectories of at most 1536 frames (F + max_lag <= 3072) run msd_power_w1_kernel — padded length 1024, 2048 or
    3072, one wave per series, residue classes of the spectrum. Every padded length on both sides of its limits,
    full and truncated lags, tiny frame counts, groups that start mid-tile / are empty / hold one entity, more series than
    one block's twelve waves, several batches: within the reported bound of the exact-difference kernel, equal to the C
    oracle at rtol 1e-9, zero at lag 0, reproducible bit for bit, device result = host result."""
    import torch

    ctx = B.default_context()
    rng = np.random.default_rng(41)
    cases = [(2, 1, 3), (3, 2, 5), (17, 16, 40), (300, 299, 1000), (512, 511, 77), (513, 510, 30), (513, 512, 30), (700, 699, 13),
             (1000, 999, 2500), (1024, 1023, 50), (1025, 1022, 50), (1025, 1024, 50), (1400, 1399, 130), (1535, 1534, 64),
             (1536, 500, 20), (1200, 1847, 0), (1100, 900, 33), (1536, 1535, 9)]
    try:
        for F, max_lag, E in cases:
            if max_lag >= F:  # (marks a truncated case written as (F, F + max_lag limit, .): keep it legal)
                max_lag = F - 1
            E = E or 25
            cuts = sorted(rng.integers(0, E + 1, 2).tolist())
            goff = [0, cuts[0], cuts[0], cuts[1], E]
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-80, 80, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, max_lag, goff, scale=0.3)
            ctx.set_option("lag_variant", 2)
            ctx.set_option("lag_batch_mb", int(rng.choice([-1, 1])))
            got = B.lag_msd(r, max_lag, goff, scale=0.3)
            bound = ctx.last_rel_bound()
            takes_w1 = F <= 1536 and F + max_lag <= 3072 and (F < 1536 or F + max_lag <= 2048)
            assert (ctx.last_kernel_name() == "msd_power_w1_kernel") == takes_w1, (F, max_lag, ctx.last_kernel_name())
            nz = exact > 0
            if nz.any():
                assert (np.abs(got[nz] - exact[nz]) / exact[nz]).max() <= bound, (F, max_lag, E, bound)
            assert (got[~nz] == 0.0).all() and (got[0] == 0.0).all()
            lags = np.unique(np.concatenate([np.arange(0, min(max_lag + 1, 30)), rng.integers(0, max_lag + 1, 20)]))
            want = C.lag_msd(r * 0.3, lags, goff)
            nzl = want > 0
            if nzl.any():
                assert (np.abs(got[lags][nzl] - want[nzl]) / want[nzl]).max() <= max(bound, 1e-11), (F, max_lag)
            assert np.array_equal(B.lag_msd(r, max_lag, goff, scale=0.3), got)
            out = torch.empty((max_lag + 1, 4, 4), dtype=torch.float64, device="cuda")
            B.lag_msd(torch.from_numpy(r).cuda(), max_lag, goff, scale=0.3, out=out)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(out.cpu().numpy(), got)
        # more than 16 groups: the block-wide kernels keep the call (they take every segment in ONE launch; this path
        # launches per segment)
        F, E = 600, 90
        goff = list(range(0, 91, 5))  # 18 groups
        r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0)
        ctx.set_option("lag_variant", 1)
        exact = B.lag_msd(r, F - 1, goff)
        ctx.set_option("lag_variant", 2)
        got = B.lag_msd(r, F - 1, goff)
        assert ctx.last_kernel_name() == "msd_power_lds_kernel"
        nz = exact > 0
        assert (np.abs(got[nz] - exact[nz]) / exact[nz]).max() <= ctx.last_rel_bound()
    finally:
        ctx.set_option("lag_variant", 1)
        ctx.set_option("lag_batch_mb", -1)


def test_lag_msd_long_series_finish_on_the_device(B):
    """Series beyond the fused kernels (F + max_lag > 16 384: trajectories of 10^4+ frames, diffusion.py:207-238) run the
    batched transforms and — round 6 — finish on the device like the fused kernels: the means equal the exact-difference
    kernel's within the reported bound and the C oracle's at rtol 1e-9, a device result with its status word equals the
    host result bit for bit, the asynchronous twin equals the synchronous call, no device-to-host copy of Q / the
    correlations is left (the call reports the bound through the same status word the multi-GPU step all-reduces)."""
    import torch

    ctx = B.default_context()
    rng = np.random.default_rng(77)
    try:
        for F, max_lag, E in ((9001, 9000, 7), (12000, 8000, 5), (20000, 19999, 3)):
            goff = [0, 2, 2, E]
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, max_lag, goff, scale=0.5)
            ctx.set_option("lag_variant", 2)
            fft = B.lag_msd(r, max_lag, goff, scale=0.5)
            bound = ctx.last_rel_bound()
            # (the bound is relative to the SMALLEST |MSD sum| over the lags: with max_lag ~ F the last lags hold one or two
            # origins of two entities, so it is far looser here than at C4 — what is tested is that the result respects it)
            # (round 6: up to F + max_lag = 24 576 the residue-class kernel, beyond it the batched transforms)
            want_kernel = ("msd_power_w12p_kernel" if F + max_lag <= 24576 and F <= 12288 else
                           "msd_power_w12p_kernel + msd_power_w12o_kernel" if F + max_lag <= 49152 and F <= 24576 else "lag_msd_fft")
            assert ctx.last_kernel_name() == want_kernel and 0.0 < bound < 1e-4, (ctx.last_kernel_name(), bound)
            assert (fft[0] == 0.0).all()
            nz = exact > 0
            rel = np.abs(fft[nz] - exact[nz]) / exact[nz]
            assert rel.max() <= bound, (F, max_lag, rel.max(), bound)
            lags = np.unique(np.concatenate([np.arange(0, 40), rng.integers(0, max_lag + 1, 60), [max_lag]]))
            want = C.lag_msd(r * 0.5, lags, goff)
            nzl = want > 0
            assert (np.abs(fft[lags][nzl] - want[nzl]) / want[nzl]).max() <= max(bound, 1e-11)
            np.testing.assert_allclose(fft[lags[:40]], want[:40], rtol=1e-9, atol=0)  # (small lags: thousands of origins)
            dev_r = torch.from_numpy(r).cuda()
            out = torch.empty((max_lag + 1, 3, 4), dtype=torch.float64, device="cuda")
            st = torch.full((1,), -1.0, dtype=torch.float64, device="cuda")
            B.lag_msd(dev_r, max_lag, goff, scale=0.5, out=out, async_=True, status_out=st).wait()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(out.cpu().numpy(), fft)
            assert float(st.item()) == bound
            again = B.lag_msd(r, max_lag, goff, scale=0.5, async_=True).wait()
            np.testing.assert_array_equal(again, fft)
            # round 6: the first pass reads the series in place and |X|^2 is reduced from the packed transform
            # (`lag_batched_fuse` 1) — the same operations in the same order as the padded copy + half spectra of round 2
            # (`lag_batched_fuse` 0): identical bits. The default (2: two passes, the second fused with the reduction) runs
            # another butterfly network and adds the rows in another order: equal within the bounds.
            try:
                ctx.set_option("lag_residue", 0)
                ctx.set_option("lag_batched_fuse", 2)
                two = B.lag_msd(r, max_lag, goff, scale=0.5)
                two_bound = ctx.last_rel_bound()
                assert ctx.last_kernel_name() == "lag_msd_fft"
                assert (np.abs(two[nz] - fft[nz]) / exact[nz]).max() <= bound + two_bound
                assert (np.abs(two[nz] - exact[nz]) / exact[nz]).max() <= two_bound
                ctx.set_option("lag_batched_fuse", 0)
                old = B.lag_msd(r, max_lag, goff, scale=0.5)
                ctx.set_option("lag_batched_fuse", 1)
                mid = B.lag_msd(r, max_lag, goff, scale=0.5)
                mid_bound = ctx.last_rel_bound()
            finally:
                ctx.set_option("lag_batched_fuse", -1)
                ctx.set_option("lag_residue", -1)
            np.testing.assert_array_equal(old, mid)
            assert (np.abs(mid[nz] - fft[nz]) / exact[nz]).max() <= bound + mid_bound
            assert (np.abs(mid[nz] - exact[nz]) / exact[nz]).max() <= mid_bound
        # the series are centred on the mean of ~512 SAMPLED frames (col_sum_sample_kernel): any constant is right as long as
        # S1 and the correlations use the same one. A motion whose period is the sampling stride (F / 512 = 19 frames) must
        # not alias into the offset (the samples are jittered), and the bound, made of the values actually used, holds.
        F, E = 10000, 24
        t = np.arange(F)[:, None, None]
        r = 3.0 * np.sin(2 * np.pi * t / 19.0 + rng.uniform(0, 6.28, (1, 3, E))) + np.cumsum(rng.normal(0, 0.02, (F, 3, E)), axis=0)
        ctx.set_option("lag_variant", 1)
        exact = B.lag_msd(r, F - 1, [0, E])
        nz = exact > 0
        ctx.set_option("lag_variant", 2)
        try:
            bounds = {}
            for sample in (-1, 0, 64):
                ctx.set_option("lag_mean_sample", sample)
                fft = B.lag_msd(r, F - 1, [0, E])
                bounds[sample] = ctx.last_rel_bound()
                assert (np.abs(fft[nz] - exact[nz]) / exact[nz]).max() <= bounds[sample], (sample, bounds)
            assert bounds[-1] < 1.5 * bounds[0], bounds  # (the sampled mean costs the bound next to nothing)
        finally:
            ctx.set_option("lag_mean_sample", -1)
        # more series than one batch holds / than the fused pass has rows per split: groups that straddle batches, splits and
        # blocks — both long-series paths (the residue-class kernel; the batched transforms), small batches forced
        for F, E, goff, mb in ((8200, 700, [0, 1, 130, 700], -1), (8193, 300, [0, 300], 1), (12288, 40, [0, 7, 7, 40], 1),
                               (11111, 130, [0, 64, 65, 130], 2), (12289, 70, [0, 3, 70], 1), (24576, 9, [0, 9], -1),
                               (17001, 150, [0, 50, 150], 3)):
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, F - 1, goff)
            nz = exact > 0
            ctx.set_option("lag_variant", 2)
            try:
                ctx.set_option("lag_batch_mb", mb)
                long_form = "msd_power_w12p_kernel + msd_power_w12o_kernel"  # (padded length 49 152: F > 12 288)
                for residue, name in ((1, "msd_power_w12p_kernel"), (2, "msd_power_w12r_kernel"), (0, "lag_msd_fft")):
                    if F > 12288 and residue:
                        name = long_form
                    ctx.set_option("lag_residue", residue)
                    # (`lag_overlap` 2: the CU-partitioned streams whatever the size — the transposition of batch k + 1 beside
                    # the transform kernel of batch k, two buffers; 0: one stream)
                    for overlap in (2, 0) if residue else (0,):
                        ctx.set_option("lag_overlap", overlap)
                        fft = B.lag_msd(r, F - 1, goff)
                        bound = ctx.last_rel_bound()
                        assert ctx.last_kernel_name() == name, (F, ctx.last_kernel_name())
                        assert (np.abs(fft[nz] - exact[nz]) / exact[nz]).max() <= bound, (F, E, residue, overlap, bound)
                        assert (fft[~nz] == 0.0).all()
                        assert np.array_equal(B.lag_msd(r, F - 1, goff), fft), (F, E, residue, overlap)
            finally:
                ctx.set_option("lag_batch_mb", -1)
                ctx.set_option("lag_residue", -1)
                ctx.set_option("lag_overlap", -1)
    finally:
        ctx.set_option("lag_variant", 1)


@pytest.mark.parametrize("mode", [1, 2])
def test_lag_msd_direct_read_option(B, mode):
    """`lag_direct` 1: the power kernel reads [F][3][E] itself (clusters of 16 blocks on adjacent columns, no transposed
    copy; kept as an option, it measured slower). `lag_direct` 2: the clusters transpose their tiles
    inside the kernel through a ring in device memory, handed from block to block (device-scope stores, per-tile ready
    counters). Either way the results must sit within the reported bound of the exact-difference kernel like the
    transposed path's — groups that start and end off the 16-column tiles, a group smaller than a tile, an empty group,
    more groups than clusters (falls back to the transposed path), entity counts off the tile width, odd column counts,
    trajectories beyond 5120 frames (eight staging units per lane and tile instead of five)."""
    ctx = B.default_context()
    rng = np.random.default_rng(15)
    try:
        for F, E, goff in ((4500, 700, [0, 700]), (5000, 333, [0, 5, 5, 141, 333]), (4100, 64, [0, 1, 64]),
                           (4200, 90, list(range(0, 91, 3))), (5120, 171, [0, 100, 171]), (2100, 257, [0, 257]),
                           (6001, 120, [0, 120]), (8192, 70, [0, 33, 70]), (7000, 45, [0, 45])):
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-100, 100, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, F - 1, goff, scale=0.7)
            ctx.set_option("lag_variant", 2)
            ctx.set_option("lag_direct", 0)
            ref = B.lag_msd(r, F - 1, goff, scale=0.7)
            b0 = ctx.last_rel_bound()
            ctx.set_option("lag_direct", mode)
            got = B.lag_msd(r, F - 1, goff, scale=0.7)
            b1 = ctx.last_rel_bound()
            again = B.lag_msd(r, F - 1, goff, scale=0.7)
            nz = exact > 0
            assert (np.abs(got[nz] - exact[nz]) / exact[nz]).max() <= b1, (F, E)
            assert (np.abs(got[nz] - ref[nz]) / exact[nz]).max() <= b0 + b1, (F, E)
            assert (got[0] == 0.0).all()
            assert np.array_equal(got, again), (F, E)  # (the hand-off between blocks leaves nothing to chance)
    finally:
        ctx.set_option("lag_variant", -1)
        ctx.set_option("lag_direct", -1)


def test_lag_msd_cluster_stall_falls_back(B):
    """`lag_direct` 3 = the in-kernel transposition with ONE cluster member withholding its signals (what a grid that is
    not resident as a whole looks like to the others): the polls run out (~1 s), the kernel raises its stall word and
    runs to its end, and the host repeats the call over the transposed copy — same numbers as `lag_direct` 0."""
    ctx = B.default_context()
    rng = np.random.default_rng(16)
    F, E = 4300, 900
    r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0)
    try:
        ctx.set_option("lag_variant", 2)
        ctx.set_option("lag_direct", 0)
        ref = B.lag_msd(r, F - 1, [0, 400, E], scale=1.0)
        ctx.set_option("lag_direct", 3)
        got = B.lag_msd(r, F - 1, [0, 400, E], scale=1.0)
        assert np.array_equal(got, ref)
        assert "repeated over the transposed copy" in ctx.last_kernel_name()
        pend = B.lag_msd(r, F - 1, [0, 400, E], scale=1.0, async_=True)  # (the repeat runs inside the completion step)
        assert np.array_equal(pend.wait(), ref)
    finally:
        ctx.set_option("lag_variant", -1)
        ctx.set_option("lag_direct", -1)


# ------------------------------------------------------------------ G2-G4
def test_xcorr_golden(B, g_acf):
    g = g_acf
    p = g["pressure"]
    fft = B.xcorr(p, method=B.XCORR_FFT)
    direct = B.xcorr(p, method=B.XCORR_DIRECT)
    for k in range(3):
        tol = 1e-10 * g["acf_wkt"][k][0]
        np.testing.assert_allclose(fft[k], g["acf_wkt"][k], rtol=0, atol=tol)
        np.testing.assert_allclose(direct[k], g["acf_brute"][k], rtol=0, atol=tol)
    j = g["flux"]
    c01 = B.xcorr(j[0, 0], j[0, 1], method=B.XCORR_FFT)
    np.testing.assert_allclose(c01, g["corr_01"], rtol=0, atol=1e-10 * abs(g["corr_01"]).max())
    d01 = B.xcorr(j[0, 0], j[0, 1], method=B.XCORR_DIRECT)
    np.testing.assert_allclose(d01, g["corr_01"], rtol=0, atol=1e-10 * abs(g["corr_01"]).max())
    # all 27 (i, j, k) correlations in one call, summed as correlate_charge_flux does (conductivity.py:207-213)
    a = np.stack([j[k, i] for i in range(3) for jj in range(3) for k in range(3)])
    b = np.stack([j[k, jj] for i in range(3) for jj in range(3) for k in range(3)])
    c = B.xcorr(a, b, method=B.XCORR_FFT).reshape(3, 9, -1).sum(axis=1)
    tot = np.vstack([c, c.sum(axis=0, keepdims=True)])
    np.testing.assert_allclose(tot, g["tot_flux"], rtol=0, atol=1e-10 * abs(g["tot_flux"]).max())


@pytest.mark.parametrize("n", [1, 2, 9, 2047, 2048, 2049, 5000, 12345])
def test_xcorr_direct_sizes_vs_oracle(B, n):
    rng = np.random.default_rng(n)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    out = B.xcorr(a, b, method=B.XCORR_DIRECT)
    expect = C.xcorr_direct(a, b)
    np.testing.assert_allclose(out, expect, rtol=0, atol=1e-12 * max(1.0, abs(expect).max()) * np.sqrt(n))
    f = B.xcorr(a, b, method=B.XCORR_FFT)
    np.testing.assert_allclose(f, expect, rtol=0, atol=1e-11 * max(1.0, abs(expect).max()) * np.sqrt(n))
    half = B.xcorr(a, b, method=B.XCORR_DIRECT, n_lags=max(1, n // 2))
    np.testing.assert_allclose(half, expect[: max(1, n // 2)], rtol=0, atol=1e-12 * max(1.0, abs(expect).max()) * np.sqrt(n))


def test_cumtrapz_golden_and_sizes(B, g_acf):
    g = g_acf
    dt = g["flux_time"][1] - g["flux_time"][0]
    out = B.cumtrapz(g["tot_flux"], dt, leading_zero=True)
    np.testing.assert_allclose(out, g["integral"], rtol=1e-9, atol=1e-12 * abs(g["integral"]).max())
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 2048, 2049, 2050, 100001):
        y = rng.standard_normal((2, n))
        for lead in (False, True):
            res = B.cumtrapz(y, 0.37, leading_zero=lead)
            for s in range(2):
                exp = O.cumtrapz(y[s], 0.37, leading_zero=lead) if n > 1 else (np.zeros(1) if lead else np.zeros(0))
                np.testing.assert_allclose(res[s], exp, rtol=1e-9, atol=1e-12 * max(1.0, abs(exp).max() if len(exp) else 1.0))


@pytest.mark.parametrize("shape", [(5, 1_000_001), (1, 9_000_000), (3, 4_196_000)])
def test_cumtrapz_across_launch_boundaries(B, shape):
    """More than 2048 tiles: the scan takes several launches, a series continues from the launch before it through the
    carry words and the two sets of totals alternate (with one set of carry words the last tile's store
    raced with the other tiles' loads whenever a continued series filled a whole launch). Integer-valued samples whose
    partial sums stay below 2^53 make every summation order exact: the result must equal numpy's cumsum bit for bit,
    and repeats must be bit-identical."""
    s, n = shape
    rng = np.random.default_rng(17 + n % 97)
    y = rng.integers(-1000, 1001, size=(s, n)).astype(np.float64)
    inc = y[:, 1:] + y[:, :-1]  # dx = 2: inc = 2 * (y0 + y1) / 2, exact
    exp = np.cumsum(inc, axis=1)
    first = B.cumtrapz(y, 2.0)
    np.testing.assert_array_equal(first, exp)
    for rep in range(3):
        np.testing.assert_array_equal(B.cumtrapz(y, 2.0), first)
    lead = B.cumtrapz(y, 2.0, leading_zero=True)
    np.testing.assert_array_equal(lead[:, 1:], exp)
    assert not lead[:, 0].any()
    # and a float case against the oracle to rounding
    z = rng.standard_normal((s, n))
    got = B.cumtrapz(z, 0.37)
    for k in range(s):
        e = O.cumtrapz(z[k], 0.37)
        np.testing.assert_allclose(got[k], e, rtol=1e-9, atol=1e-12 * max(1.0, abs(e).max()))


# ------------------------------------------------------------------ full-size properties (BASELINE C2)
def test_c2_full_size_properties(B):
    """N=10 000, F=200 at full size: frame 0 against the C oracle, and size-independent identities."""
    import torch
    from mdproptools_amd import synth

    cfg = synth.rdf_config("C2")
    n, F, L = cfg["n_atoms"], cfg["n_frames"], cfg["box_len"]
    xyz = synth.rdf_frames(n, range(F), L, cfg["seed_offset"])
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    d = torch.from_numpy(xyz).cuda()
    full, part, ov = B.rdf_loop(d, ty, box, rel, cfg["r_cut"], cfg["bin_size"], 400)
    cf, cp, cov = C.rdf_pairs(xyz[0], ty, rel, box[0], 400.0, 0.05, 400)
    np.testing.assert_array_equal(full[0], cf)
    np.testing.assert_array_equal(part[0], cp)
    # every pair belongs to exactly one unordered type pair: sum of partials (a == b rows already doubled,
    # a != b rows counted once per unordered pair) reproduces the full histogram
    mult = np.array([1 if a == b else 2 for a, b in rel], dtype=np.uint64)
    np.testing.assert_array_equal((part * mult[None, :, None]).sum(axis=1), full)
    # frame-summed call == sum over frames of the per-frame call
    fs, ps, ovs = B.rdf_loop(d, ty, box, rel, cfg["r_cut"], cfg["bin_size"], 400, per_frame=False)
    np.testing.assert_array_equal(fs, full.sum(axis=0))
    np.testing.assert_array_equal(ps, part.sum(axis=0))
    assert ovs == ov
    # relabelling invariance: a permutation of the atoms leaves every histogram unchanged
    perm = np.random.default_rng(0).permutation(n)
    f2, p2, _ = B.rdf_loop(np.ascontiguousarray(xyz[:2][:, :, perm]), ty[perm], box[:2], rel, 20.0, 0.05, 400)
    np.testing.assert_array_equal(f2, full[:2])
    np.testing.assert_array_equal(p2, part[:2])
    # ideal gas: g(r) = 1 within counting noise once averaged over 200 frames (r > 2 A)
    sv = O.shell_volume(0.05, 400)
    gr = full.sum(axis=0) / F / (n * (n / L ** 3) * sv)
    assert abs(gr[40:].mean() - 1.0) < 2e-3


def test_c3_geometry_full_atoms_properties(B):
    """BASELINE C3 geometry at full N (100 000 atoms, L = 104 A, r_cut 20 A, 400 bins), 6 frames: the packed-f32
    sweep against the all-f64 sweep, and the size-independent identities (sum of partials = full, frame-summed =
    sum of per-frame, ideal gas g(r) = 1)."""
    import torch
    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context

    cfg = synth.rdf_config("C3")
    n, L, F = cfg["n_atoms"], cfg["box_len"], 6
    xyz = synth.rdf_frames(n, range(F), L, cfg["seed_offset"])
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    d = torch.from_numpy(xyz).cuda()
    out = {}
    for v in (0, 1):
        ctx = Context(0)
        ctx.set_option("rdf_pk", v)
        out[v] = B.rdf_loop(d, ty, box, rel, cfg["r_cut"], cfg["bin_size"], 400, ctx=ctx)
        assert ("<3," in ctx.last_kernel_name()) == bool(v)
        if v:
            fs, ps, ovs = B.rdf_loop(d, ty, box, rel, cfg["r_cut"], cfg["bin_size"], 400, per_frame=False, ctx=ctx)
        ctx.close()
    full, part, ov = out[1]
    np.testing.assert_array_equal(full, out[0][0])
    np.testing.assert_array_equal(part, out[0][1])
    assert ov == out[0][2] == ovs
    np.testing.assert_array_equal(fs, full.sum(axis=0))
    np.testing.assert_array_equal(ps, part.sum(axis=0))
    mult = np.array([1 if a == b else 2 for a, b in rel], dtype=np.uint64)
    np.testing.assert_array_equal((part * mult[None, :, None]).sum(axis=1), full)
    sv = O.shell_volume(0.05, 400)
    gr = full.sum(axis=0) / F / (n * (n / L ** 3) * sv)
    assert abs(gr[40:].mean() - 1.0) < 2e-3


# ------------------------------------------------------------------ spatial culling (cell-list variant)
def test_culled_path_equals_dense_and_oracle(B):
    """Morton-sorted tiles + bounding-box culling must give the same integers as the dense sweep."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(77)
    F, n = 3, 2600  # 11 tiles
    boxes = np.array([[40.0, 42.0, 44.0], [41.0, 41.0, 41.0], [39.5, 43.0, 40.0]])
    xyz = np.stack([rng.uniform(0, 1, (3, n)) * boxes[f][:, None] + 2.0 for f in range(F)])
    stray = rng.choice(n, 100, replace=False)
    xyz[:, :, stray] += (rng.integers(-2, 3, (F, 3, 100)) * boxes[:, :, None])  # atoms outside the box
    ty = rng.integers(1, 4, (F, n)).astype(np.int32)  # per-frame types
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3]])
    cuts = [3.0, 4.5, 6.0, 7.5]
    res = {}
    for cull in (0, 1):
        ctx = Context(0)
        ctx.set_option("rdf_cull", cull)
        for jsplit in (1, 3):
            ctx.set_option("rdf_jsplit", jsplit)
            ctx.set_option("rdf_batch", 2 if jsplit == 3 else 0)  # 3 frames in batches of 2 + 1
            res[(cull, jsplit)] = (B.rdf_loop(xyz, ty, boxes, rel, 7.5, 0.05, 150, ctx=ctx),
                                   B.rdf_loop(xyz, ty, boxes, rel, 7.5, 0.05, 150, per_frame=False, ctx=ctx),
                                   B.cn_loop(xyz, ty, boxes, rel, cuts, ctx=ctx))
        ctx.close()
    ref = res[(0, 1)]
    for f in range(F):
        cf, cp, cov = C.rdf_pairs(xyz[f], ty[f], rel, boxes[f], 7.5 * 7.5, 0.05, 150)
        np.testing.assert_array_equal(ref[0][0][f], cf)
        np.testing.assert_array_equal(ref[0][1][f], cp)
        np.testing.assert_array_equal(ref[2][f], C.cn_pairs(xyz[f], ty[f], rel, boxes[f], [c * c for c in cuts]))
    for key, val in res.items():
        np.testing.assert_array_equal(val[0][0], ref[0][0])
        np.testing.assert_array_equal(val[0][1], ref[0][1])
        assert val[0][2] == ref[0][2]
        np.testing.assert_array_equal(val[1][0], ref[0][0].sum(axis=0))
        np.testing.assert_array_equal(val[1][1], ref[0][1].sum(axis=0))
        np.testing.assert_array_equal(val[2], ref[2])


@pytest.mark.parametrize("case", ["centred", "offset", "half_box_lattice", "long_cut", "strays"])
def test_culled_path_wrap_variants(B, case):
    """The scalar-j kernel hoists wrap decisions out of the pair loop per (wave box, group box): cells with
    lo != 0, pairs at d == L/2 exactly (the reference wraps only for d > L/2), cutoffs near L/2 and atoms
    box lengths outside the cell must all give the C oracle's integers, for every kernel organisation."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng({"centred": 1, "offset": 2, "half_box_lattice": 3, "long_cut": 4, "strays": 5}[case])
    n = 4200  # 17 tiles
    L = np.array([32.0, 32.0, 32.0])
    r_cut, nb = 7.0, 140
    if case == "centred":
        xyz = rng.uniform(-0.5, 0.5, (3, n)) * L[:, None]
    elif case == "offset":
        L = np.array([30.0, 34.0, 38.0])
        xyz = rng.uniform(-0.3, 0.7, (3, n)) * L[:, None]
    elif case == "half_box_lattice":
        # multiples of L/64 = 0.5: differences hit +-L/2 = +-16 exactly, and many distances are bin edges
        xyz = rng.integers(0, 64, (3, n)).astype(np.float64) * 0.5
        r_cut, nb = 6.0, 120
    elif case == "long_cut":
        xyz = rng.uniform(0.0, 1.0, (3, n)) * L[:, None]
        r_cut, nb = 15.5, 310  # just below L/2: nearly every group pair needs a decision
    else:
        xyz = rng.uniform(0.0, 1.0, (3, n)) * L[:, None]
        idx = rng.choice(n, 300, replace=False)
        xyz[:, idx] += rng.integers(-3, 4, (3, 300)) * L[:, None]
    xyz = np.ascontiguousarray(xyz)[None]
    ty = rng.integers(1, 4, n).astype(np.int32)
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3], [1, 3]])
    cuts = [2.5, 3.0, 4.5, 6.0, 5.0]
    cf, cp, cov = C.rdf_pairs(xyz[0], ty, rel, L, r_cut * r_cut, 0.05, nb)
    ccn = C.cn_pairs(xyz[0], ty, rel, L, [c * c for c in cuts])
    assert cf.sum() > 0
    for cull, sj, sort, rows in ((0, 1, -1, -1), (1, 0, 0, -1), (1, 1, 0, 0), (1, 1, 1, 1), (1, 2, 1, 1), (1, 2, 0, 0)):
        ctx = Context(0)
        ctx.set_option("rdf_cull", cull)
        ctx.set_option("rdf_sj", sj)
        ctx.set_option("rdf_sort", sort)  # spatial sort with global counters / one block per frame
        ctx.set_option("rdf_rows", rows)  # ordered-pair rows without a row table / class rows
        for per_frame in (True, False):
            full, part, ov = B.rdf_loop(xyz, ty, L[None], rel, r_cut, 0.05, nb, per_frame=per_frame, ctx=ctx)
            full, part = (full[0], part[0]) if per_frame else (full, part)
            np.testing.assert_array_equal(full, cf, err_msg="%s cull=%d sj=%d sort=%d rows=%d pf=%d" % (case, cull, sj, sort, rows, per_frame))
            np.testing.assert_array_equal(part, cp)
            assert ov == cov
        np.testing.assert_array_equal(B.cn_loop(xyz, ty, L[None], rel, cuts, ctx=ctx)[0], ccn)
        ctx.close()


@pytest.mark.parametrize("n_types,nbins,bin_size", [(1, 60, 0.1), (2, 1500, 0.004), (5, 120, 0.05), (7, 400, 0.015)])
def test_ordered_rows_and_class_rows_agree(B, n_types, nbins, bin_size):
    """Scalar-j kernel: ordered-pair rows addressed through the bin guess (few types) and class rows with the
    row table (the fallback when T^2 rows do not fit LDS) against the C oracle; relations cover a == b, a != b,
    a repeated pair and a type pair nobody asks for; many narrow bins stress the guard band."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(100 + n_types)
    n, L = 2600, np.array([31.0, 29.0, 33.0])
    xyz = (rng.uniform(0, 1, (2, 3, n)) * L[None, :, None])
    ty = rng.integers(1, n_types + 1, n).astype(np.int32)
    rel = np.array([[1, 1], [1, n_types], [n_types, 1], [max(1, n_types - 1), n_types]])
    r_cut = nbins * bin_size
    box = np.tile(L, (2, 1))
    want = [C.rdf_pairs(xyz[f], ty, rel, L, r_cut * r_cut, bin_size, nbins) for f in range(2)]
    for rows in (0, 1):
        ctx = Context(0)
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_rows", rows)
        full, part, ov = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, ctx=ctx)
        for f in range(2):
            np.testing.assert_array_equal(full[f], want[f][0], err_msg="rows=%d" % rows)
            np.testing.assert_array_equal(part[f], want[f][1], err_msg="rows=%d" % rows)
        assert ov == want[0][2] + want[1][2]
        fs, ps, _ = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=False, ctx=ctx)
        np.testing.assert_array_equal(fs, full.sum(axis=0))
        np.testing.assert_array_equal(ps, part.sum(axis=0))
        ctx.close()


@pytest.mark.parametrize("case", ["plain", "centred_cell", "strays"])
def test_culled_atoms_x_sites_equals_dense_and_oracle(B, case):
    """Atoms x molecule sites (rdf_cn.py:122-162) through the spatially culled scalar-j kernel (two sorted sets,
    rectangular tile lists) against the dense sweep and the C oracle; a site may coincide with an atom."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng({"plain": 21, "centred_cell": 22, "strays": 23}[case])
    F, n, m = 2, 4100, 1300
    L = np.array([36.0, 38.0, 40.0])
    lo = -0.5 if case == "centred_cell" else 0.0
    xyz = (rng.uniform(lo, lo + 1, (F, 3, n)) * L[None, :, None])
    sites = (rng.uniform(lo, lo + 1, (F, 3, m)) * L[None, :, None])
    sites[:, :, :50] = xyz[:, :, :50]  # coincident points: rsq == 0 -> bin 0
    if case == "strays":
        xyz[:, :, rng.choice(n, 200, replace=False)] += (rng.integers(-2, 3, (F, 3, 200)) * L[None, :, None])
        sites[:, :, rng.choice(m, 80, replace=False)] -= (rng.integers(-2, 3, (F, 3, 80)) * L[None, :, None])
    ty = rng.integers(1, 5, n).astype(np.int32)
    st = rng.integers(1, 4, m).astype(np.int32)
    rel = np.array([[1, 1], [2, 3], [4, 2], [1, 1], [3, 3]])
    cuts = [3.0, 5.0, 6.5, 2.0, 4.0]
    box = np.tile(L, (F, 1))
    want = [C.rdf_rect(xyz[f], ty, sites[f], st, rel, L, 49.0, 0.05, 140) for f in range(F)]
    want_cn = [C.cn_rect(xyz[f], ty, sites[f], st, rel, L, [c * c for c in cuts]) for f in range(F)]
    for cull, sort in ((0, -1), (1, 0), (1, 1)):
        ctx = Context(0)
        ctx.set_option("rdf_cull", cull)
        ctx.set_option("rdf_sort", sort)
        part, ov = B.rdf_mol_loop(xyz, ty, sites, st, box, rel, 7.0, 0.05, 140, ctx=ctx)
        for f in range(F):
            np.testing.assert_array_equal(part[f], want[f][0], err_msg="%s cull=%d" % (case, cull))
        assert ov == sum(w[1] for w in want)
        ps, _ = B.rdf_mol_loop(xyz, ty, sites, st, box, rel, 7.0, 0.05, 140, per_frame=False, ctx=ctx)
        np.testing.assert_array_equal(ps, part.sum(axis=0))
        cn = B.cn_mol_loop(xyz, ty, sites, st, box, rel, cuts, ctx=ctx)
        for f in range(F):
            np.testing.assert_array_equal(cn[f], want_cn[f])
        if cull:
            assert "sj_kernel" in ctx.last_kernel_name()
        ctx.close()


def test_culled_path_randomised_against_dense(B):
    """40 random geometries (triclinic-free boxes of random aspect, cell origins anywhere, cutoffs from 5 % to
    49 % of the shortest edge, clustered and uniform atoms, a few strays): the culled scalar-j sweep with its
    hoisted wrap decisions must give the dense sweep's integers (the dense sweep itself is pinned to the oracle
    by the tests above)."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(20250328)
    dense, culled = Context(0), Context(0)
    dense.set_option("rdf_cull", 0)
    culled.set_option("rdf_cull", 1)
    for trial in range(40):
        n = int(rng.integers(2100, 5200))
        L = rng.uniform(18.0, 60.0, 3)
        lo = rng.uniform(-1.0, 1.0, 3) * L
        F = int(rng.integers(1, 4))
        if trial % 3 == 0:  # clustered: blobs around a few centres, wrapped into the cell
            centres = rng.uniform(0, 1, (6, 3))
            frac = (centres[rng.integers(0, 6, n)] + rng.normal(0, 0.08, (n, 3))) % 1.0
            xyz = np.stack([(frac.T * L[:, None] + lo[:, None])] * F) + rng.normal(0, 0.05, (F, 3, n))
        else:
            xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None] + lo[None, :, None]
        if trial % 4 == 1:
            idx = rng.choice(n, 30, replace=False)
            xyz[:, :, idx] += rng.integers(-2, 3, (F, 3, 30)) * L[None, :, None]
        r_cut = float(rng.uniform(0.05, 0.49) * L.min())
        nbins = int(r_cut / 0.05)
        n_types = int(rng.integers(1, 6))
        ty = rng.integers(1, n_types + 1, n).astype(np.int32)
        rel = np.array([[1, 1], [1, n_types], [n_types, n_types]])
        box = np.tile(L, (F, 1))
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, 0.05, nbins, ctx=dense)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, 0.05, nbins, ctx=culled)
        msg = "trial %d n=%d L=%s lo=%s r_cut=%.3f" % (trial, n, L, lo, r_cut)
        np.testing.assert_array_equal(a[0], b[0], err_msg=msg)
        np.testing.assert_array_equal(a[1], b[1], err_msg=msg)
        assert a[2] == b[2], msg
        cuts = [0.3 * r_cut, 0.6 * r_cut, r_cut]
        np.testing.assert_array_equal(B.cn_loop(xyz, ty, box, rel, cuts, ctx=dense),
                                      B.cn_loop(xyz, ty, box, rel, cuts, ctx=culled), err_msg=msg)
    dense.close()
    culled.close()


# ------------------------------------------------------------------ packed-f32 classification sweep (rdf_pk)
def _pk_case(rng, trial):
    """A geometry that stresses the packed-f32 sweep: the cutoff on a bin edge (its precondition), lattices whose
    distances sit exactly ON bin edges, cutoffs up to L/2, cell origins anywhere, strays box lengths outside."""
    n = int(rng.integers(2100, 6000))
    bin_size = float(rng.choice([0.05, 0.1, 0.025]))
    L = rng.uniform(20.0, 64.0, 3)
    if trial % 5 == 0:
        L[:] = L[0]  # cubic
    lo = rng.uniform(-1.0, 1.0, 3) * L
    F = int(rng.integers(1, 4))
    nbins = int(rng.uniform(0.06, 0.4999) * L.min() / bin_size)
    r_cut = nbins * bin_size  # on a bin edge, as in every RDF call that divides its cutoff into whole bins
    if trial % 6 == 5:        # ... or inside the last bin: the sweep then guards the cutoff's own error band too
        r_cut = min((nbins + float(rng.uniform(0.03, 0.97))) * bin_size, 0.4999 * float(L.min()))
        nbins = int(r_cut / bin_size)
    kind = trial % 4
    if kind == 0:  # a lattice with spacing = a whole number of bins: most distances are exactly on edges
        g = int(round(n ** (1 / 3))) + 1
        a = bin_size * max(1, int(L.min() / g / bin_size))
        idx = rng.choice(g ** 3, n, replace=False)
        cell = np.stack([idx % g, (idx // g) % g, idx // (g * g)]).astype(np.float64)
        xyz = np.stack([cell * a + lo[:, None]] * F)
    elif kind == 1:  # blobs
        centres = rng.uniform(0, 1, (6, 3))
        frac = (centres[rng.integers(0, 6, n)] + rng.normal(0, 0.07, (n, 3))) % 1.0
        xyz = np.stack([(frac.T * L[:, None] + lo[:, None])] * F) + rng.normal(0, 0.05, (F, 3, n))
    else:
        xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None] + lo[None, :, None]
    if trial % 3 == 1:
        idx = rng.choice(n, 25, replace=False)
        xyz[:, :, idx] += rng.integers(-2, 3, (F, 3, 25)) * L[None, :, None]
    box = np.tile(L, (F, 1))
    if trial % 7 == 3:
        box = box * (1.0 + 0.01 * np.arange(F))[:, None]  # NPT: another box every frame
    n_types = int(rng.integers(1, 5))
    rel = np.array([[1, 1], [1, n_types], [n_types, n_types]])
    if trial % 5 == 4:  # many types, all of them named: the T^2 ordered rows do not fit LDS -> class rows + row table
        n_types = int(rng.integers(6, 10))
        rel = np.array([[a, b] for a in range(1, n_types + 1) for b in range(a, n_types + 1)])[::3][:14]
        if trial % 10 == 9:  # (round 6) EVERY pair named: nothing for displaced rows to merge — the ordered rows in one
            # 16-wave block per CU where they fit the whole LDS, else the packed class rows in several passes
            rel = np.array([[a, b] for a in range(1, n_types + 1) for b in range(a, n_types + 1)])
    ty = rng.integers(1, n_types + 1, n).astype(np.int32)
    return xyz, ty, box, rel, r_cut, bin_size, nbins


def test_packed_f32_sweep_equals_f64_sweep(B):
    """rdf_pk: pairs are classified with packed f32 arithmetic and every pair inside the error band of a bin edge
    or of the cutoff is resolved by the exact f64 chain — the integers must be those of the all-f64 sweep (itself
    pinned to the oracle and the goldens above), per frame and frame-summed."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(77001)
    f64, pk = Context(0), Context(0)
    for ctx, v in ((f64, 0), (pk, 1)):
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_pk", v)
    engaged = 0
    for trial in range(36):
        xyz, ty, box, rel, r_cut, bin_size, nbins = _pk_case(rng, trial)
        per_frame = bool(trial % 2)
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=f64)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=pk)
        engaged += any(t in pk.last_kernel_name() for t in ("<3,", "<4,", "<5,", "<6,"))
        msg = "trial %d n=%d box=%s r_cut=%.4f bin=%.3f kernel=%s" % (trial, xyz.shape[2], box[0], r_cut, bin_size,
                                                                     pk.last_kernel_name())
        np.testing.assert_array_equal(a[0], b[0], err_msg=msg)
        np.testing.assert_array_equal(a[1], b[1], err_msg=msg)
        assert a[2] == b[2], msg
    assert engaged >= 24, "the packed-f32 kernel ran in only %d of 36 cases" % engaged
    f64.close()
    pk.close()


def test_rdf_and_cn_from_one_sweep(B):
    """mdhip_rdf_cn_atomic: RDF histograms and coordination counts from ONE sweep equal the two separate calls on
    every geometry of the packed-sweep stress set (lattices on bin edges, blobs, NPT boxes, strays box lengths
    outside the cell, ordered rows and class rows, cutoff on and inside a bin), with coordination cutoffs on bin
    edges, between them, equal for several relations, zero, and — for the fallback inside the call — beyond r_cut."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(4242)
    ctx = Context(0)
    ctx.set_option("rdf_cull", 1)
    ctx.set_option("cn_pk", 1)  # mdhip_cn_atomic through the packed sweep as well (the default since round 2)
    edge_table = Context(0)  # the f64 edge-table CN kernel (mdhip_cn_atomic itself now prefers the packed sweep too)
    edge_table.set_option("rdf_cull", 1)
    edge_table.set_option("cn_pk", 0)
    fused = 0
    for trial in range(24):
        xyz, ty, box, rel, r_cut, bin_size, nbins = _pk_case(rng, trial)
        R = len(rel)
        cuts = list(rng.uniform(0.1, 0.95, R) * r_cut)
        if trial % 3 == 0:
            cuts[0] = bin_size * int(0.5 * nbins)          # exactly a bin edge
        if trial % 4 == 1 and R > 1:
            cuts[1] = cuts[0]                               # two relations share a cutoff
        if trial % 5 == 2:
            cuts[-1] = 0.0                                  # a relation that counts nothing
        if trial % 8 == 7:
            cuts[0] = 1.2 * r_cut                           # beyond the RDF cutoff: two sweeps inside the call
        per_frame = bool(trial % 2)
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=ctx)
        cn = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=per_frame, ctx=edge_table)
        assert "<1," in edge_table.last_kernel_name() or "pair_hist_kernel" in edge_table.last_kernel_name() \
            or "fast" in edge_table.last_kernel_name(), edge_table.last_kernel_name()
        cn_pk = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=per_frame, ctx=ctx)  # coarse histogram + split bins
        np.testing.assert_array_equal(cn_pk, cn, err_msg="cn through the packed sweep, trial %d" % trial)
        f, p_, ov, cn2 = B.rdf_cn_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, cuts, per_frame=per_frame, ctx=ctx)
        name = ctx.last_kernel_name()
        fused += "true>" in name and name.count(",") == 2 and name.endswith(", true>")
        msg = "trial %d n=%d r_cut=%.4f bin=%.3f cuts=%s kernel=%s" % (trial, xyz.shape[2], r_cut, bin_size, cuts, name)
        np.testing.assert_array_equal(f, a[0], err_msg=msg)
        np.testing.assert_array_equal(p_, a[1], err_msg=msg)
        assert ov == a[2], msg
        np.testing.assert_array_equal(cn2, cn, err_msg=msg)
    assert fused >= 12, "the one-sweep kernel ran in only %d of 24 cases" % fused
    # and against the oracle directly, C2-like shape
    from mdproptools_amd import synth

    n, L = 4000, 36.8
    xyz = synth.rdf_frames(n, range(2), L, 9)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((2, 3), L)
    cuts = synth.cn_cutoffs(len(rel))
    for r_cut, bin_size, nbins in ((12.0, 0.05, 240), (12.02, 0.05, 240)):
        f, p_, ov, cn = B.rdf_cn_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, cuts, ctx=ctx)
        assert ctx.last_kernel_name().endswith(", true>"), ctx.last_kernel_name()
        for fr in range(2):
            cf, cp, _ = C.rdf_pairs(xyz[fr], ty, rel, box[fr], r_cut * r_cut, bin_size, nbins)
            np.testing.assert_array_equal(f[fr], cf)
            np.testing.assert_array_equal(p_[fr], cp)
            np.testing.assert_array_equal(cn[fr], C.cn_pairs(xyz[fr], ty, rel, box[fr], [c * c for c in cuts]))
    ctx.close()
    edge_table.close()


def test_packed_f32_sweep_against_oracle(B):
    """The default path of a C2-shaped call (packed-f32 sweep) against the C oracle: cutoff on a bin edge (<3, .>)
    and inside the last bin (<4, .>: the cutoff's own error band is guarded per pair)."""
    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context

    n, L = 4000, 36.8
    xyz = synth.rdf_frames(n, range(2), L, 9)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((2, 3), L)
    ctx = Context(0)
    ctx.set_option("rdf_cull", 1)
    for r_cut, bin_size, nbins in ((12.0, 0.05, 240), (12.02, 0.05, 240), (18.0, 0.1, 180), (18.3, 0.1, 183)):
        full, part, ov = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, ctx=ctx)
        on_edge = abs(r_cut / bin_size - round(r_cut / bin_size)) < 1e-6
        assert ("<3," if on_edge else "<4,") in ctx.last_kernel_name(), (r_cut, ctx.last_kernel_name())
        ovs = 0
        for f in range(2):
            cf, cp, cov = C.rdf_pairs(xyz[f], ty, rel, box[f], r_cut * r_cut, bin_size, nbins)
            np.testing.assert_array_equal(full[f], cf)
            np.testing.assert_array_equal(part[f], cp)
            ovs += cov
        assert ov == ovs
    ctx.close()


def test_packed_f32_sweep_more_types(B):
    """5 types at 400 bins (25 ordered rows, 50 KB of LDS per 8-wave block) and 7 types at 200 bins still take the
    packed-f32 sweep; 8 types at 400 bins do not fit and fall back. All against the all-f64 sweep and the C oracle."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(515)
    n, L, F = 4200, 37.0, 2
    xyz = rng.uniform(0, L, (F, 3, n))
    box = np.full((F, 3), L)
    f64, pk = Context(0), Context(0)
    for ctx, v in ((f64, 0), (pk, 1)):
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_pk", v)
    # (the last case: 9 types of which the relations name 3 — the other 6 share one index, so 4 types reach the kernel)
    for n_types, nbins, bin_size, expect_pk in ((5, 400, 0.04, True), (7, 200, 0.08, True), (8, 400, 0.04, False),
                                                (9, 400, 0.04, True)):
        ty = rng.integers(1, n_types + 1, n).astype(np.int32)
        rel = np.array([[a, b] for a in range(1, n_types + 1) for b in range(a, n_types + 1)][:12])
        if n_types == 9:
            rel = np.array([[2, 2], [2, 7], [7, 4], [4, 4]])
        r_cut = nbins * bin_size
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, ctx=f64)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, ctx=pk)
        assert ("<3," in pk.last_kernel_name()) == expect_pk, (n_types, nbins, pk.last_kernel_name())
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2]
        cf, cp, _ = C.rdf_pairs(xyz[0], ty, rel, box[0], r_cut * r_cut, bin_size, nbins)
        np.testing.assert_array_equal(b[0][0], cf)
        np.testing.assert_array_equal(b[1][0], cp)
    f64.close()
    pk.close()


def test_displaced_rows_reference_shape_and_random_relations(B):
    """Displaced ordered rows (row = A[ti] + B[tj]): the reference's own shape — nine atom types, the
    relations 9-1, 9-4, 9-6, 9-9, 1-3 of the example notebook: six type indices, 36 plain rows that do not fit — takes
    the packed table-free sweep `<3, .>` instead of the class-row kernel `<5, .>`, and random relation sets over 5-12
    types give the same integers with displacement off (class rows), forced on, the all-f64 sweep and the C oracle;
    per-frame and frame-summed; RDF + CN from one sweep; atoms x sites."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(20251004)
    n, L, F = 4300, 36.0, 3
    xyz = rng.uniform(0, L, (F, 3, n))
    box = np.full((F, 3), L)
    ctxs = {}
    # ("big": no displacement, but the plain 36+ rows in one 16-wave block per CU — the second new layout of round 6)
    for tag, opts in (("auto", {}), ("off", {"rdf_disp": 0, "rdf_big": 0}), ("force", {"rdf_disp": 2}), ("f64", {"rdf_pk": 0}),
                      ("big", {"rdf_disp": 0})):
        c = ctxs[tag] = Context(0)
        c.set_option("rdf_cull", 1)
        for k, v in opts.items():
            c.set_option(k, v)
    nbins, bin_size = 400, 0.04
    r_cut = nbins * bin_size
    cases = [(9, np.array([[9, 1], [9, 4], [9, 6], [9, 9], [1, 3]]))]
    for _ in range(9):
        T = int(rng.integers(5, 13))
        pairs = [(a, b) for a in range(1, T + 1) for b in range(a, T + 1)]
        pick = rng.permutation(len(pairs))[: int(rng.integers(2, 8))]
        cases.append((T, np.array([pairs[k] if rng.random() < 0.5 else pairs[k][::-1] for k in pick])))
    took_disp = 0
    for k, (T, rel) in enumerate(cases):
        ty = rng.integers(1, T + 1, n).astype(np.int32)
        per_frame = bool(k % 2)
        res = {}
        for tag, c in ctxs.items():
            res[tag] = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=c)
            res[tag + "_kernel"] = c.last_kernel_name()
        if k == 0:
            assert "<3," in res["auto_kernel"], res["auto_kernel"]
            assert "<5," in res["off_kernel"], res["off_kernel"]
        took_disp += ("<3," in res["auto_kernel"]) and ("<5," in res["off_kernel"])
        for tag in ("off", "force", "f64", "big"):
            msg = "case %d T=%d rel=%s %s vs auto (%s / %s)" % (k, T, rel.tolist(), tag, res[tag + "_kernel"], res["auto_kernel"])
            np.testing.assert_array_equal(res[tag][0], res["auto"][0], err_msg=msg)
            np.testing.assert_array_equal(res[tag][1], res["auto"][1], err_msg=msg)
            assert res[tag][2] == res["auto"][2], msg
        cf, cp, _ = C.rdf_pairs(xyz[0], ty, rel, box[0], r_cut * r_cut, bin_size, nbins)
        if per_frame:
            np.testing.assert_array_equal(res["auto"][0][0], cf)
            np.testing.assert_array_equal(res["auto"][1][0], cp)
        # RDF + CN from one sweep on the displaced rows
        cuts = list(rng.uniform(0.1, 0.9, len(rel)) * r_cut)
        cuts[0] = bin_size * 137
        f, p_, ov, cn = B.rdf_cn_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, cuts, per_frame=per_frame, ctx=ctxs["auto"])
        np.testing.assert_array_equal(f, res["auto"][0])
        np.testing.assert_array_equal(p_, res["auto"][1])
        cn_ref = B.cn_loop(xyz, ty, box, rel, cuts, per_frame=per_frame, ctx=ctxs["off"])
        np.testing.assert_array_equal(cn, cn_ref, err_msg="one-sweep CN, case %d (%s)" % (k, ctxs["auto"].last_kernel_name()))
        if per_frame:
            np.testing.assert_array_equal(cn[0], C.cn_pairs(xyz[0], ty, rel, box[0], [c * c for c in cuts]))
    assert took_disp >= 4, took_disp
    # atoms x sites: ordered (atom type, site type) classes, 9 x 5 types, four relations
    m = 2300
    sites = rng.uniform(0, L, (F, 3, m))
    ty = rng.integers(1, 10, n).astype(np.int32)
    st = rng.integers(1, 6, m).astype(np.int32)
    rel = np.array([[9, 1], [9, 4], [2, 4], [7, 5]])
    parts = {tag: B.rdf_mol_loop(xyz, ty, sites, st, box, rel, r_cut, bin_size, nbins, ctx=c) for tag, c in ctxs.items()}
    for tag in ("off", "force", "f64", "big"):
        np.testing.assert_array_equal(parts[tag][0], parts["auto"][0], err_msg=tag)
    cp, _ = C.rdf_rect(xyz[0], ty, sites[0], st, rel, box[0], r_cut * r_cut, bin_size, nbins)
    np.testing.assert_array_equal(parts["auto"][0][0], cp)
    for c in ctxs.values():
        c.close()


def test_packed_f32_sweep_far_from_origin(B):
    """A cell 1e5 A away from the origin: the f32 boxes and tile centres lose ~0.01 A there, the tile-relative f32
    coordinates of the pair chain do not. Against the all-f64 sweep and the C oracle."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(99)
    n, F = 3600, 2
    L = np.array([38.0, 41.0, 36.5])
    lo = np.array([3.0e4, -7.0e4, 1.0e5])
    xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None] + lo[None, :, None]
    ty = rng.integers(1, 4, n).astype(np.int32)
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3]])
    box = np.tile(L, (F, 1))
    f64, pk = Context(0), Context(0)
    for ctx, v in ((f64, 0), (pk, 1)):
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_pk", v)
    for r_cut, bin_size, nbins in ((12.0, 0.05, 240), (18.0, 0.05, 360)):
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, ctx=f64)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, ctx=pk)
        assert "<3," in pk.last_kernel_name()
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2]
        cf, cp, _ = C.rdf_pairs(xyz[0], ty, rel, box[0], r_cut * r_cut, bin_size, nbins)
        np.testing.assert_array_equal(b[0][0], cf)
        np.testing.assert_array_equal(b[1][0], cp)
    f64.close()
    pk.close()


def test_packed_f32_sweep_atoms_x_sites(B):
    """Atoms x sites (R5) through the ordered-row layout and the packed-f32 sweep: against the all-f64 sweep, the
    class-row layout and the C oracle; sites coincide with atoms, strays on both sides, a cutoff close to L/2."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(808)
    F, n, m = 2, 5200, 2300
    L = np.array([33.0, 35.0, 34.0])
    xyz = rng.uniform(0, 1, (F, 3, n)) * L[None, :, None]
    sites = rng.uniform(0, 1, (F, 3, m)) * L[None, :, None]
    sites[:, :, :40] = xyz[:, :, :40]
    xyz[:, :, rng.choice(n, 60, replace=False)] += rng.integers(-2, 3, (F, 3, 60)) * L[None, :, None]
    sites[:, :, rng.choice(m, 30, replace=False)] -= rng.integers(-1, 2, (F, 3, 30)) * L[None, :, None]
    ty = rng.integers(1, 6, n).astype(np.int32)
    st = rng.integers(1, 4, m).astype(np.int32)
    rel = np.array([[1, 1], [2, 3], [4, 2], [3, 3]])
    box = np.tile(L, (F, 1))
    res = {}
    for r_cut, bin_size, nbins in ((7.0, 0.05, 140), (16.4, 0.1, 164)):
        for tag, opts in (("pk", {"rdf_pk": 1}), ("f64", {"rdf_pk": 0}), ("rows", {"rdf_pk": 0, "rdf_rows": 0})):
            ctx = Context(0)
            ctx.set_option("rdf_cull", 1)
            for k, v in opts.items():
                ctx.set_option(k, v)
            res[tag] = B.rdf_mol_loop(xyz, ty, sites, st, box, rel, r_cut, bin_size, nbins, ctx=ctx)
            name = ctx.last_kernel_name()
            assert ("<3," in name) == (tag == "pk") and ("<2," in name) == (tag == "f64"), (tag, name)
            ctx.close()
        for tag in ("f64", "rows"):
            np.testing.assert_array_equal(res["pk"][0], res[tag][0], err_msg=tag)
            assert res["pk"][1] == res[tag][1]
        want = C.rdf_rect(xyz[0], ty, sites[0], st, rel, L, r_cut * r_cut, bin_size, nbins)
        np.testing.assert_array_equal(res["pk"][0][0], want[0])


def test_fewer_bins_than_the_cutoff_spans(B):
    """A caller of the C-ABI may pass fewer bins than int(r_cut / bin_size): every in-cutoff pair beyond the last
    bin is then overflow (as in the oracle), and the table-free fast kernels — whose rows have nbins + 1 words —
    must not be used."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(31)
    n, L = 3000, 32.0
    xyz = rng.uniform(0, L, (1, 3, n))
    ty = rng.integers(1, 4, n).astype(np.int32)
    rel = np.array([[1, 1], [1, 2], [3, 3]])
    box = np.full((1, 3), L)
    for cull in (0, 1):
        ctx = Context(0)
        ctx.set_option("rdf_cull", cull)
        full, part, ov = B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 150, ctx=ctx)  # int(12 / 0.05) = 240 bins
        cf, cp, cov = C.rdf_pairs(xyz[0], ty, rel, box[0], 144.0, 0.05, 150)
        np.testing.assert_array_equal(full[0], cf)
        np.testing.assert_array_equal(part[0], cp)
        assert ov == cov and cov > 0
        ctx.close()


def test_packed_f32_sweep_class_rows(B):
    """9 atom types, all named by relations: the 81 ordered rows do not fit LDS, so the packed-f32 sweep runs on the
    class-row layout with its LDS row table (<5, .>; <6, .> with the cutoff inside a bin). Against the all-f64 sweep
    (class rows, MODE 0) and the C oracle."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(606)
    n, L, F = 4300, 37.5, 2
    xyz = rng.uniform(0, L, (F, 3, n))
    xyz[:, :, rng.choice(n, 40, replace=False)] += rng.integers(-2, 3, (F, 3, 40)) * L
    box = np.full((F, 3), L)
    ty = rng.integers(1, 10, n).astype(np.int32)
    rel = np.array([[a, b] for a in range(1, 10) for b in range(a, 10)])[::4]  # 12 relations over all 9 types
    f64, pk = Context(0), Context(0)
    for ctx, v in ((f64, 0), (pk, 1)):
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_pk", v)
        ctx.set_option("rdf_disp", 0)  # (round 6: displaced rows would fit these 12 relations into the table-free sweep)
    for r_cut, bin_size, nbins, tag in ((16.0, 0.04, 400, "<5,"), (16.03, 0.04, 400, "<6,"), (18.7, 0.1, 187, "<5,")):
        for per_frame in (True, False):
            a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=f64)
            b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=pk)
            assert tag in pk.last_kernel_name() and "<0," in f64.last_kernel_name(), (pk.last_kernel_name(), f64.last_kernel_name())
            np.testing.assert_array_equal(a[0], b[0])
            np.testing.assert_array_equal(a[1], b[1])
            assert a[2] == b[2]
            if per_frame:
                cf, cp, _ = C.rdf_pairs(xyz[0], ty, rel, box[0], r_cut * r_cut, bin_size, nbins)
                np.testing.assert_array_equal(b[0][0], cf)
                np.testing.assert_array_equal(b[1][0], cp)
    f64.close()
    pk.close()


def test_packed_class_rows_in_several_passes(B):
    """Every unordered pair of nine types named (45 relations, the "all partial RDFs" call on the reference's example):
    the class rows do not fit LDS at once — 45 x 401 words — and there is nothing for displaced rows to merge. Round 6
    runs the packed class-row sweep in several passes (two here) instead of handing the call to the all-f64 kernel;
    12 types / 78 relations take four. Against the all-f64 sweep, the single-pass-only setting and the C oracle,
    per-frame and frame-summed; the overflow count (pairs beyond the last bin edge) is not multiplied by the passes."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(909)
    n, L, F = 4300, 37.5, 2
    xyz = rng.uniform(0, L, (F, 3, n))
    box = np.full((F, 3), L)
    ctxs = {}
    # "big": the default since the same round — the 81 ordered rows of nine types in ONE 16-wave block per CU (the whole LDS
    # for one histogram, `pair_hist_sj_kernel<3, ., false, true>`); twelve types (144 rows) do not fit that either
    for tag, opts in (("pk", {"rdf_big": 0}), ("big", {}), ("one", {"rdf_pk_passes": 0, "rdf_big": 0}), ("f64", {"rdf_pk": 0})):
        c = ctxs[tag] = Context(0)
        c.set_option("rdf_cull", 1)
        for k, v in opts.items():
            c.set_option(k, v)
    for T, r_cut, bin_size, nbins in ((9, 16.0, 0.04, 400), (12, 16.03, 0.04, 400), (9, 15.99, 0.04, 399)):
        ty = rng.integers(1, T + 1, n).astype(np.int32)
        rel = np.array([[a, b] for a in range(1, T + 1) for b in range(a, T + 1)])
        for per_frame in (True, False):
            res = {tag: B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=c)
                   for tag, c in ctxs.items()}
            names = {tag: c.last_kernel_name() for tag, c in ctxs.items()}
            assert ("<5," in names["pk"] or "<6," in names["pk"]) and "<0," in names["one"] and "<0," in names["f64"], names
            assert ctxs["pk"].last_kernel_ms()[1] >= 2, ctxs["pk"].last_kernel_ms()  # launches of the pair kernel
            if T == 9:
                assert names["big"].endswith(", false, true>") and ("<3," in names["big"] or "<4," in names["big"]), names
                assert ctxs["big"].last_kernel_ms()[1] == 1
            else:
                assert names["big"] == names["pk"], names
            for tag in ("one", "f64", "big"):
                np.testing.assert_array_equal(res[tag][0], res["pk"][0], err_msg=tag)
                np.testing.assert_array_equal(res[tag][1], res["pk"][1], err_msg=tag)
                assert res[tag][2] == res["pk"][2], (tag, res[tag][2], res["pk"][2])
            if per_frame:
                cf, cp, cov = C.rdf_pairs(xyz[0], ty, rel, box[0], r_cut * r_cut, bin_size, nbins)
                np.testing.assert_array_equal(res["pk"][0][0], cf)
                np.testing.assert_array_equal(res["pk"][1][0], cp)
    for c in ctxs.values():
        c.close()


def test_culled_path_large_box_auto(B):
    """BASELINE C3 geometry at reduced N (same density: L = 48.3 A for 10k atoms, r_cut 6.8): the
    automatic choice takes the culled path; result against the C oracle."""
    from mdproptools_amd import synth

    n, L = 10_000, 48.27
    xyz = synth.rdf_frames(n, range(2), L, 3)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((2, 3), L)
    full, part, ov = B.rdf_loop(xyz, ty, box, rel, 6.8, 0.05, 136)
    cn = B.cn_loop(xyz, ty, box, rel, synth.cn_cutoffs(10))
    for f in range(2):
        cf, cp, _ = C.rdf_pairs(xyz[f], ty, rel, box[f], 6.8 * 6.8, 0.05, 136)
        np.testing.assert_array_equal(full[f], cf)
        np.testing.assert_array_equal(part[f], cp)
        np.testing.assert_array_equal(cn[f], C.cn_pairs(xyz[f], ty, rel, box[f], [c * c for c in synth.cn_cutoffs(10)]))


# ------------------------------------------------------------------ full-size properties (BASELINE C3, C4, C5)
def test_c3_geometry_properties(B):
    """100 000 atoms, L = 104 A (BASELINE C3), 2 of its frames: the culled sweep equals the dense sweep,
    partial histograms add up to the full one, CN equals the histogram summed below a bin-edge cutoff."""
    import torch
    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context

    cfg = synth.rdf_config("C3")
    n, L = cfg["n_atoms"], cfg["box_len"]
    xyz = torch.from_numpy(synth.rdf_frames(n, range(2), L, cfg["seed_offset"])).cuda()
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((2, 3), L)
    out = {}
    for cull in (0, 1):
        ctx = Context(0)
        ctx.set_option("rdf_cull", cull)
        out[cull] = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, ctx=ctx)
        ctx.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    full, part, ov = out[1]
    mult = np.array([1 if a == b else 2 for a, b in rel], dtype=np.uint64)
    np.testing.assert_array_equal((part * mult[None, :, None]).sum(axis=1), full)
    # 8.0 A is an exact bin edge for bin_size 0.05 only if edges[160] == 64.0; use the histogram's own edge
    e = B.bin_edges(0.05, 400)
    cut = float(np.sqrt(e[160]))
    if cut * cut == e[160]:
        cn = B.cn_loop(xyz, ty, box, rel, [cut] * len(rel))
        np.testing.assert_array_equal(cn, part[:, :, :160].sum(axis=2))
    # ideal gas: in-cutoff fraction of all pairs
    frac = float(full[0].sum()) / 2 / (n * (n - 1) / 2)
    assert abs(frac - 4 / 3 * np.pi * 20.0 ** 3 / L ** 3) < 2e-4


def test_c4_msd_properties(B):
    """50 000 entities (BASELINE C4), 64 frames: scaling law, origin symmetry, group additivity,
    full-lag average of a ballistic trajectory."""
    import torch
    from mdproptools_amd import synth

    E, F = 50_000, 64
    r = synth.random_walk(E, F)
    d = torch.from_numpy(r).cuda()
    pairs = [(0, t) for t in range(F)]
    s1 = B.msd_pairs(d, pairs, [0, E], scale=1.0)
    s3 = B.msd_pairs(d, pairs, [0, E], scale=3.0)
    np.testing.assert_allclose(s3, 9.0 * s1, rtol=1e-12)
    back = B.msd_pairs(d, [(t, 0) for t in range(F)], [0, E], scale=1.0)
    np.testing.assert_allclose(back, s1, rtol=1e-13)  # (a-b)^2 == (b-a)^2
    split = B.msd_pairs(d, pairs, [0, 12_345, 30_000, E], scale=1.0)
    np.testing.assert_allclose(split.sum(axis=1, keepdims=True), s1, rtol=1e-12)
    # a random walk with sigma = 0.1 per axis per frame: msd(t) ~ 3 * 0.01 * t
    msd_last = s1[-1, 0, 3] / E
    assert abs(msd_last / (3 * 0.01 * (F - 1)) - 1.0) < 0.02
    win = B.msd_windows(d, 4, scale=1.0)
    kept = r[::4]
    np.testing.assert_allclose(win[:, 3], ((kept[1:] - kept[:-1]) ** 2).sum(axis=(0, 1)), rtol=1e-12)
    # ballistic motion r = r0 + v t: msd[lag] = |v|^2 lag^2 exactly in expectation per entity
    rng = np.random.default_rng(2)
    v = rng.normal(0, 1, (3, 2000))
    traj = rng.uniform(0, 50, (1, 3, 2000)) + v[None] * np.arange(300)[:, None, None]
    out = B.lag_msd(traj, 299, [0, 2000])
    expect = (v ** 2).sum(axis=0).mean() * np.arange(300) ** 2
    np.testing.assert_allclose(out[:, 0, 3], expect, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 16, 17, 255, 256, 257, 4096, 4097, 40_000, 70_000, 1_100_000])
def test_xcorr_fft_all_pass_plans(B, n):
    """The library's own power-of-two transforms (csrc/fft_pow2.hip) behind MDHIP_XCORR_FFT: every shape of the pass
    plan (no pass at H = 1, one short pass, tiles narrower than 16 columns, two, three passes, radices 2..256),
    cross- and auto-correlation, batches, against numpy's FFT of the same zero-padded series. The sums
    c[k] (n - k) are compared with 1e-13 |a| |b| (their Cauchy-Schwarz bound), i.e. rounding level at every lag."""
    rng = np.random.default_rng(n)
    nb = 3 if n <= 70_000 else 1
    a = rng.standard_normal((nb, n)) * np.array([1.0, 1e3, 1e-4])[:nb, None]
    b = rng.standard_normal((nb, n)) + 0.5
    L = 2
    while L < 2 * n:
        L *= 2
    w = n - np.arange(n)

    def ref(x, y):
        return np.fft.irfft(np.fft.rfft(x, L) * np.conj(np.fft.rfft(y, L)), L)[..., :n]

    got = B.xcorr(a, b, method=B.XCORR_FFT)
    scale = np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1)
    assert (np.abs(got * w - ref(a, b)).max(axis=1) <= 1e-13 * scale).all()
    auto = B.xcorr(a, method=B.XCORR_FFT)
    assert (np.abs(auto * w - ref(a, a)).max(axis=1) <= 1e-13 * np.linalg.norm(a, axis=1) ** 2).all()
    np.testing.assert_allclose(auto[:, 0], (a * a).mean(axis=1), rtol=1e-12)
    few = B.xcorr(a, b, method=B.XCORR_FFT, n_lags=max(1, n // 3))
    np.testing.assert_array_equal(few, got[:, :max(1, n // 3)])


def test_lag_msd_batched_transforms_with_a_large_radix_pass(B):
    """Full-lag MSD of a 70 000-frame trajectory: padded length 2^18, the batched global transforms
    (mdhip_fft_r2c / mdhip_fft_c2r, lag_variant 4) plan 2^17 points as a radix-2^9 pass (radix-8 network) + a radix-2^8
    pass (radix-4 network) over a batch of 60 series; against the difference kernel within the reported bound."""
    ctx = B.default_context()
    rng = np.random.default_rng(77)
    try:
        # 70 000 frames: 2^17 points = radix 2^9 (radix-8 network) + 2^8; 140 000 frames: 2^18 points = 2^9 + 2^9, so the
        # inverse's LAST pass (conjugating output) runs the radix-8 network too
        for F, E, lags in ((70_000, 20, 3000), (140_000, 86, 800)):
            r = np.cumsum(rng.normal(0, 0.1, (F, 3, E)), axis=0) + rng.uniform(-50, 50, (1, 3, E))
            ctx.set_option("lag_variant", 1)
            exact = B.lag_msd(r, lags, [0, 12, E])
            ctx.set_option("lag_variant", 4)
            fft = B.lag_msd(r, lags, [0, 12, E])
            bound = ctx.last_rel_bound()
            assert ctx.last_kernel_name() == "lag_msd_fft" and 0.0 < bound < 1e-8
            nz = exact > 0
            assert (np.abs(fft[nz] - exact[nz]) / exact[nz]).max() <= bound, (F, bound)
    finally:
        ctx.set_option("lag_variant", 1)


@pytest.mark.parametrize("n", [140_000, 300_000, 1_000_000])
def test_xcorr_fft_large_radix_passes(B, n):
    """Two-pass plans (radix 2^9 and 2^10 through the radix-8 network, fft_pass8_kernel / fft_mid_acf_kernel: last rounds of 8 and of 16
    points, 4- and 8-column tiles, the XCD-paired tile order) against numpy's FFT at rounding level, and against the
    three-pass plan of the radix-4 network (fft_logr 8) and the radix-4 network at the large radices (fft_net8 0), with
    the spectrum step inside the inverse's first pass (default) and as its own kernel (fft_specfuse 0), cross- and
    autocorrelation: every plan within 1e-13 |a| |b| of the reference at every lag."""
    ctx = B.default_context()
    rng = np.random.default_rng(n)
    a = rng.standard_normal((3, n)) * np.array([1.0, 1e3, 1e-4])[:, None]
    b = rng.standard_normal((3, n)) + 0.5
    L = 2
    while L < 2 * n:
        L *= 2
    w = n - np.arange(n)
    ref_ab = np.fft.irfft(np.fft.rfft(a, L) * np.conj(np.fft.rfft(b, L)), L)[..., :n]
    ref_aa = np.fft.irfft(np.abs(np.fft.rfft(a, L)) ** 2, L)[..., :n]
    na, nb_ = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)
    try:
        # ({}: the autocorrelation runs in THREE launches, round 5 — fft_mid_acf_kernel: the forward transform's second
        # pass, the spectrum step and the inverse's first pass on one tile, the inverse through the transposed network;
        # fft_mid 0 the four launches it replaces, 2 its narrower tiles)
        for opts in ({}, {"fft_mid": 0}, {"fft_mid": 2}, {"fft_specfuse": 0, "fft_mid": 0}, {"fft_logr": 8}, {"fft_net8": 0},
                     {"fft_net8": 2}):
            ctx.set_option("fft_mid", opts.get("fft_mid", 1))
            ctx.set_option("fft_logr", opts.get("fft_logr", 10))
            ctx.set_option("fft_net8", opts.get("fft_net8", 1))
            ctx.set_option("fft_specfuse", opts.get("fft_specfuse", 1))
            got = B.xcorr(a, b, method=B.XCORR_FFT)
            assert (np.abs(got * w - ref_ab).max(axis=1) <= 1e-13 * na * nb_).all(), opts
            auto = B.xcorr(a, method=B.XCORR_FFT, n_lags=n // 2)
            assert (np.abs(auto * w[: n // 2] - ref_aa[:, : n // 2]).max(axis=1) <= 1e-13 * na * na).all(), opts
    finally:
        ctx.set_option("fft_logr", 10)
        ctx.set_option("fft_net8", 1)
        ctx.set_option("fft_specfuse", 1)
        ctx.set_option("fft_mid", 1)


def test_c5_acf_properties(B):
    """n = 1e5 AR(1) series (BASELINE C5 at a tenth of its length): FFT and direct estimators agree,
    acf[0] is the mean square, the cumulative trapezoid of a constant is a ramp."""
    from mdproptools_amd import synth

    p = synth.ar1_series(100_000)
    fft = B.xcorr(p, method=B.XCORR_FFT)
    direct = B.xcorr(p, method=B.XCORR_DIRECT)
    for k in range(3):
        np.testing.assert_allclose(fft[k], direct[k], rtol=0, atol=1e-10 * direct[k][0])
        np.testing.assert_allclose(direct[k][0], np.mean(p[k] ** 2), rtol=1e-12)
        # the last lag is the product of the two end samples
        np.testing.assert_allclose(direct[k][-1], p[k][-1] * p[k][0], rtol=1e-9, atol=1e-9 * direct[k][0])
    ramp = B.cumtrapz(np.full((2, 100_000), 2.5), 0.5, leading_zero=True)
    np.testing.assert_allclose(ramp[0], 1.25 * np.arange(100_000), rtol=1e-12)
    lin = B.cumtrapz(np.arange(100_001.0), 1.0)
    np.testing.assert_allclose(lin, 0.5 * np.arange(1, 100_001) ** 2, rtol=1e-12)


# ------------------------------------------------------------------ residence autocorrelation
def test_shell_residence_vs_oracle(B):
    """Exact autocovariance numerators of the shell indicator: random walkers in a periodic box, shells with
    and without a lower bound, same-type relation (diagonal cleared), runs longer than 64 frames, an empty
    shell; integers must equal the oracle's."""
    rng = np.random.default_rng(41)
    # (the last two, round 6: DENSE shells — half of all pairs inside, every pair of the table occupied, long probe chains —
    # and more frames than one mask word, with the larger set on either side)
    for F, ni, nj, lo, hi, same in [(30, 12, 40, 0.0, 3.0, False), (150, 7, 25, 1.5, 4.0, False),
                                     (70, 33, 33, 0.0, 3.5, True), (10, 5, 9, 0.0, 0.01, False),
                                     (200, 40, 300, 0.0, 6.0, False), (130, 310, 30, 1.0, 5.5, False),
                                     (90, 260, 260, 0.0, 5.0, True)]:
        L = np.array([11.0, 12.0, 13.0])
        n = ni if same else ni + nj
        r = rng.uniform(0, 1, (1, 3, n)) * L[None, :, None] + np.cumsum(rng.normal(0, 0.15, (F, 3, n)), axis=0)
        xi = np.ascontiguousarray(r[:, :, :ni])
        xj = xi if same else np.ascontiguousarray(r[:, :, ni:])
        box = np.tile(L, (F, 1))
        counts, nrec = B.shell_residence(xi, xj, box, lo * lo, hi * hi, exclude_diagonal=same)
        h = np.array([O.shell_indicator(xi[f].T, xj[f].T, L, lo * lo, hi * hi, same) for f in range(F)])
        want = O.residence_counts(h)
        np.testing.assert_array_equal(counts.astype(np.int64), want)
        assert nrec == int(h.sum())
        # the pairs' presence masks live in a hash table sized from an estimate and filled in ONE sweep (round 6: no record
        # list, no sort); a table that fills up is swept again with one sized from the hit count: forced here with 8 slots
        ctx = B.default_context()
        ctx.set_option("residence_cap", 5)
        try:
            c2, n2 = B.shell_residence(xi, xj, box, lo * lo, hi * hi, exclude_diagonal=same)
        finally:
            ctx.set_option("residence_cap", 0)
        np.testing.assert_array_equal(c2, counts)
        assert n2 == nrec


def test_residence_time_dropin_golden(B, tmp_path):
    """ResidenceTime.calc_auto_correlation / fit_auto_correlation against the real reference's output."""
    from conftest import load_golden
    from mdproptools_amd import io as mio
    from mdproptools_amd.dynamical.residence_time import ResidenceTime

    g = load_golden("residence.npz")
    cols = [str(c) for c in g["columns"]]
    for s, b, t in zip(g["steps"], g["bounds"], g["frames"]):
        mio.write_dump(str(tmp_path / ("dump.nvt.%d.dump" % s)), int(s), b, cols, t)
    rt = ResidenceTime(g["r_cut"].tolist(), g["rel"].tolist(), str(tmp_path / "dump.nvt.*.dump"), dt=2,
                       num_mols=g["num_mols"].tolist(), num_atoms_per_mol=g["num_atoms_per_mol"].tolist(),
                       working_dir=str(tmp_path))
    rt.calc_auto_correlation()
    assert list(rt.corr_df.columns) == [str(c) for c in g["corr_cols"]]
    got, ref = rt.corr_df.to_numpy(), g["corr"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_allclose(got[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-12, atol=1e-15)
    assert (tmp_path / "auto_correlation.csv").exists()
    # default ids (the reference stops with a ValueError there): selection by LAMMPS type works
    rt9 = ResidenceTime([[0, 14.0]], [[9], [9]], str(tmp_path / "dump.nvt.*.dump"), working_dir=str(tmp_path))
    rt9.calc_auto_correlation()
    np.testing.assert_allclose(rt9.corr_df["9-9"].to_numpy(), ref[:, 3], rtol=1e-12)  # Mg-Mg = pseudo-type 32-32
