"""CPU-only checks of the boundary: the library builds, loads, exports every symbol the header
declares, its host-side edge table is exact, and the product fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import REPO
from mdproptools_amd import _lib


def test_header_symbols_are_exported():
    text = open(os.path.join(REPO, "include", "mdhip.h")).read()
    declared = set(re.findall(r"\b(mdhip_[a-z_0-9]+)\s*\(", text))
    declared.discard("mdhip_ctx")
    assert declared, "no declarations found"
    lib = _lib.load()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.mdhip_version() == 600


def _ref_bin(rsq, ddr):
    return (np.sqrt(rsq) / ddr).astype(np.int64)


@pytest.mark.parametrize("r_cut,ddr", [(20.0, 0.05), (13.0, 0.05), (10.0, 0.1), (12.0, 0.02), (20.0, 0.01),
                                       (7.3, 0.173)])
def test_bin_edges_are_exact(r_cut, ddr):
    nb = int(r_cut / ddr)
    e = _lib.bin_edges(ddr, nb)
    assert e[0] == 0.0 and np.all(np.diff(e) > 0)
    k = np.arange(1, nb + 1)
    # e[k] is in bin k and the double just below it is in bin k-1
    np.testing.assert_array_equal(_ref_bin(e[1:], ddr), k)
    np.testing.assert_array_equal(_ref_bin(np.nextafter(e[1:], 0.0), ddr), k - 1)
    # the reference rule agrees with table binning on random rsq, including values next to edges
    rng = np.random.default_rng(7)
    rsq = np.concatenate([rng.uniform(0, r_cut ** 2, 20000), e, np.nextafter(e[1:], 0), np.nextafter(e, np.inf)])
    rsq = rsq[rsq < e[-1]]
    np.testing.assert_array_equal(np.searchsorted(e, rsq, side="right") - 1, _ref_bin(rsq, ddr))


def test_edges_are_not_the_naive_squares():
    """(k*ddr)**2 is the wrong edge for most bins, and bin 400 is reachable below 20**2."""
    e = _lib.bin_edges(0.05, 400)
    naive = (np.arange(401) * 0.05) ** 2
    assert int((e != naive).sum()) == 242
    assert e[400] < 400.0  # overflow bin reachable for (20, 0.05)
    e13 = _lib.bin_edges(0.05, 260)
    assert e13[260] == 169.0  # not reachable for (13, 0.05)


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, not compute on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.MdhipError):
        _lib.Context(0)
    from mdproptools_amd import backend

    with pytest.raises(_lib.MdhipError):
        backend.cumtrapz(np.arange(8.0), 1.0)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "mdproptools_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                hits = re.findall(r"^\s*(?:from|import)\s+oracle\b|__import__\([\"']oracle|import_module\([\"']oracle",
                                  src, flags=re.M)
                assert not hits, (os.path.join(root, f), hits)
    # tools/ holds product-side measurement scripts: no oracle there either (oracle-based ones live in tests/bench/)
    for f in os.listdir(os.path.join(REPO, "tools")):
        if f.endswith(".py"):
            src = open(os.path.join(REPO, "tools", f)).read()
            assert not re.findall(r"^\s*(?:from|import)\s+oracle\b", src, flags=re.M), f
    # bench.py: only inside the cpu_baseline_* (the timed CPU sample) and cpu_check_* (the checker of a leg's result)
    # functions — never in the code that produces a measured value
    src = open(os.path.join(REPO, "bench.py")).read()
    for m in re.finditer(r"^\s*(?:from|import)\s+oracle\b", src, flags=re.M):
        enclosing = re.findall(r"^def (\w+)\(", src[: m.start()], flags=re.M)[-1]
        assert enclosing.startswith("cpu_baseline") or enclosing.startswith("cpu_check"), enclosing


def test_header_is_plain_c(tmp_path):
    """include/mdhip.h is the drop-in boundary: it must compile as C99 (no C++-isms, no torch / HIP types) and
    link against libmdhip.so from a C program."""
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    src = tmp_path / "use_mdhip.c"
    src.write_text(
        '#include "mdhip.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  double e[5];\n"
        "  if (mdhip_version() <= 0) return 1;\n"
        "  if (mdhip_bin_edges(0.05, 4, e) != MDHIP_OK || e[0] != 0.0) return 2;\n"
        '  printf("%d %.17g\\n", mdhip_version(), e[4]);\n'
        "  return 0;\n}\n")
    exe = tmp_path / "use_mdhip"
    lib_dir = os.path.join(REPO, "mdproptools_amd")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(REPO, "include"),
                        str(src), "-L", lib_dir, "-l:libmdhip.so", "-Wl,-rpath," + lib_dir, "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0, (run.stdout, run.stderr)
    assert float(run.stdout.split()[1]) > 0.039  # edges[4] ~ (4 * 0.05)^2


def _f32_guess_emulation(rng, n, L, r_cut, bin_size, s_cap, wrap):
    """Emulates, pair by pair in IEEE float32, what pair_hist_sj_kernel<3> computes for the bin guess (DESIGN.md
    4.1b): tile-relative coordinates rounded to f32, the packed difference / product / two fused multiply-adds,
    an (exactly rounded) square root, 1/ddr rounded to f32 and the final fma — next to the reference's f64 value."""
    f32 = np.float32
    Lv = np.array(L, dtype=np.float64)
    c = rng.uniform(0, 1, 3) * Lv                               # tile centre
    xj = c + rng.uniform(-1, 1, (n, 3)) * 0.25 * (s_cap - r_cut)  # j atoms around the centre
    # i atoms anywhere within reach of the j atoms, possibly across the periodic boundary
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    xi = xj + u * rng.uniform(0, 1.02 * r_cut, (n, 1)) + rng.integers(-1, 2, (n, 3)) * Lv
    # the reference (rdf_cn.py:35-69): single wrap, f64
    d = xi - xj
    d = np.where(np.abs(d) > Lv / 2, d - np.sign(d) * Lv, d)
    rsq_ref = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    g_ref = np.sqrt(rsq_ref) / bin_size
    # the device chain; it only covers lanes with |x_i - c| + h < 1.49 L (the reference's single wrap is the nearest
    # image there) — anything else is swept by the f64 chain
    q = xi - c
    h = np.abs(xj - c).max(axis=0)
    cov = np.all(np.abs(q) + h < 1.49 * Lv, axis=1)
    xi, xj, q, g_ref, rsq_ref = xi[cov], xj[cov], q[cov], g_ref[cov], rsq_ref[cov]
    xr_i = (q - Lv * np.rint(q / Lv)).astype(f32)               # f64 per lane, then one rounding to f32
    xr_j = (xj - c).astype(f32)
    dd = xr_i - xr_j                                            # f32 subtraction
    if wrap:
        L32, iL32 = Lv.astype(f32), (1.0 / Lv).astype(f32)
        nn = np.rint(dd * iL32)                                 # f32 product, round to even
        dd = (dd.astype(np.float64) - nn.astype(np.float64) * L32.astype(np.float64)).astype(f32)  # one fma rounding
    else:
        # where the kernel takes the plain difference: |d'| <= L - r_cut - margin on every axis (axis_plain)
        margin = 1.0e-3 * Lv + 1.0e-3
        keep = np.all(np.abs(dd.astype(np.float64)) <= Lv - r_cut - margin, axis=1)
        dd, g_ref, rsq_ref = dd[keep], g_ref[keep], rsq_ref[keep]
        # d' is the nearest image unless some |d'| > L/2 — and then BOTH d' and the nearest image are beyond the
        # cutoff on that axis alone, so the pair is out of the cutoff for the reference and for the f32 chain alike
        far = np.any(np.abs(dd.astype(np.float64)) > Lv / 2, axis=1)
        assert np.all(rsq_ref[far] > r_cut * r_cut * (1 + 1e-6))
        assert np.all(np.max(np.abs(dd[far].astype(np.float64)), axis=1) > r_cut * (1 + 1e-6))
        dd, g_ref, rsq_ref = dd[~far], g_ref[~far], rsq_ref[~far]
    r = dd[:, 0] * dd[:, 0]                                     # f32 product
    r = (dd[:, 1].astype(np.float64) ** 2 + r.astype(np.float64)).astype(f32)  # fma: exact in f64, one rounding
    r = (dd[:, 2].astype(np.float64) ** 2 + r.astype(np.float64)).astype(f32)
    s = np.sqrt(r.astype(np.float64)).astype(f32)               # <= 0.5 ulp; the bound allows 1 ulp (v_sqrt_f32)
    gs = f32(1.0 / bin_size)
    g32 = (s.astype(np.float64) * np.float64(gs)).astype(f32)   # fma with a zero addend
    inside = rsq_ref < (1.02 * r_cut) ** 2
    return np.abs(g32.astype(np.float64) - g_ref)[inside]


@pytest.mark.parametrize("L,r_cut,bin_size", [((50.0, 50.0, 50.0), 20.0, 0.05), ((104.0, 104.0, 104.0), 20.0, 0.05),
                                              ((31.0, 44.0, 37.5), 15.4, 0.1), ((26.0, 26.0, 26.0), 12.5, 0.025)])
def test_packed_f32_error_bound_covers_emulation(L, r_cut, bin_size):
    """The exactness of the packed-f32 sweep rests on mdhip_pk_error_bound: every pair whose f32 guess is farther
    than that from an integer is binned without the f64 chain. Emulate the device's f32 operations on 400k random
    pairs per case (with and without the per-pair wrap) and compare the worst observed deviation with the bound."""
    lib = _lib.load()
    nbins = int(round(r_cut / bin_size))
    edge = (256 * L[0] * L[1] * L[2] / 10000.0) ** (1 / 3)
    s_cap = r_cut + 3.5 * edge
    bound = lib.mdhip_pk_error_bound(r_cut, bin_size, nbins, 1, s_cap, max(L))
    assert 0 < bound < 0.01
    rng = np.random.default_rng(4242)
    for wrap in (False, True):
        err = _f32_guess_emulation(rng, 400_000, L, r_cut, bin_size, s_cap, wrap)
        assert len(err) > 100_000
        # the emulated chain has an exactly rounded sqrt (the device: 1 ulp), so it must sit well inside
        assert err.max() < 0.8 * bound, (wrap, err.max(), bound)
        assert err.max() > 0.02 * bound, "the bound is vacuous"


def _displace(cls):
    import ctypes as C

    lib = _lib.load()
    cls = np.ascontiguousarray(cls, dtype=np.int32)
    n_ti, n_tj = cls.shape
    a = np.full(n_ti, -1, np.int32)
    b = np.full(n_tj, -1, np.int32)
    rc = np.full(n_ti * n_tj + 8, -7, np.int32)
    rows = C.c_int(-1)
    ip = lambda v: v.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    assert lib.mdhip_row_displacement(n_ti, n_tj, ip(cls), ip(a), ip(b), ip(rc), C.byref(rows)) == 0
    return rows.value, a, b, rc


def _tri_classes(n, rels):
    cls = np.full((n, n), len(rels), np.int32)
    for k, (x, y) in enumerate(rels):
        cls[x, y] = cls[y, x] = k
    return cls


def test_row_displacement_never_mixes_classes():
    """mdhip_row_displacement: row(ti, tj) = a[ti] + b[tj] may merge type pairs only within one class, every
    row's class is what row_cls says, fewer rows than n_ti * n_tj — for the reference's own example (nine atom types, the
    five relations 9-1, 9-4, 9-6, 9-9, 1-3 -> six type indices, 36 plain rows), star-shaped and random relation sets,
    and rectangular (atoms x sites) tables."""
    rng = np.random.default_rng(5)
    cases = [_tri_classes(6, [(4, 0), (4, 2), (4, 3), (4, 4), (0, 1)]),  # C1: indices of 1, 3, 4, 6, 9, other
             _tri_classes(3, [(1, 0), (1, 1)]),                            # altered ids: 32-17, 32-32
             _tri_classes(10, [(9, j) for j in range(1, 10)]),
             _tri_classes(9, [(i, j) for i in range(9) for j in range(i, 9)])]  # every pair named: nothing to gain
    for _ in range(30):
        n = int(rng.integers(3, 13))
        pairs = [(i, j) for i in range(n) for j in range(i, n)]
        pick = rng.permutation(len(pairs))[: int(rng.integers(1, min(len(pairs), 12) + 1))]
        cases.append(_tri_classes(n, [pairs[k] for k in pick]))
    for _ in range(10):  # rectangular: ordered (atom type, site type) classes
        n_ti, n_tj = int(rng.integers(2, 9)), int(rng.integers(2, 6))
        cls = np.full((n_ti, n_tj), 0, np.int32)
        k = int(rng.integers(1, 6))
        cls[:] = k
        for c in range(k):
            cls[rng.integers(0, n_ti), rng.integers(0, n_tj)] = c
        cases.append(cls)
    gained = 0
    for cls in cases:
        rows, a, b, rc = _displace(cls)
        n_ti, n_tj = cls.shape
        if rows == 0:
            continue
        gained += 1
        assert 0 < rows < n_ti * n_tj
        assert a.min() >= 0 and b.min() >= 0 and a.max() + b.max() + 1 == rows
        seen = {}
        for i in range(n_ti):
            for j in range(n_tj):
                r = int(a[i] + b[j])
                assert rc[r] == cls[i, j], (cls, a, b, i, j)
                seen[r] = True
        assert all(rc[r] == -1 for r in range(rows) if r not in seen)
        again = _displace(cls)
        assert again[0] == rows and np.array_equal(again[1], a) and np.array_equal(again[2], b)  # deterministic
    rows_c1 = _displace(cases[0])[0]
    assert 0 < rows_c1 <= 18, rows_c1  # 36 plain rows; <= 33 needed to fit a third of LDS at 400 bins
    assert _displace(cases[3])[0] == 0
    assert gained >= 20
