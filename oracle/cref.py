"""
oracle/cref.py — ctypes loader for oracle/liboracle.so (the C checker).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from mdproptools_amd/.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "cpu_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _prep(xyz_soa, types, rel):
    xyz = np.ascontiguousarray(xyz_soa, dtype=np.float64)
    ty = np.ascontiguousarray(types, dtype=np.int32)
    rl = np.ascontiguousarray(rel, dtype=np.int32).reshape(-1, 2)
    return xyz, ty, rl


def rdf_pairs(xyz_soa, types, rel, lengths, rc2, ddr, nbins, rows=None):
    """xyz_soa [3,N]; returns (full u64[nb], part u64[R,nb], overflow). rows=(i0, i1): head rows i0 <= i < i1 only."""
    xyz, ty, rl = _prep(xyz_soa, types, rel)
    L = np.ascontiguousarray(lengths, dtype=np.float64)
    full = np.zeros(nbins, dtype=np.uint64)
    part = np.zeros((len(rl), nbins), dtype=np.uint64)
    ov = C.c_uint64(0)
    i0, i1 = (0, xyz.shape[1]) if rows is None else rows
    lib().oracle_rdf_pairs_rows(
        C.c_int64(xyz.shape[1]), C.c_int64(i0), C.c_int64(i1), _p(xyz, C.c_double), _p(ty, C.c_int32), C.c_int(len(rl)),
        _p(rl, C.c_int32), _p(L, C.c_double), C.c_double(rc2), C.c_double(ddr), C.c_int(nbins),
        _p(full, C.c_uint64), _p(part, C.c_uint64), C.byref(ov))
    return full, part, ov.value


def cn_pairs(xyz_soa, types, rel, lengths, rc2_list, rows=None):
    xyz, ty, rl = _prep(xyz_soa, types, rel)
    L = np.ascontiguousarray(lengths, dtype=np.float64)
    rc2 = np.ascontiguousarray(rc2_list, dtype=np.float64)
    cn = np.zeros(len(rl), dtype=np.uint64)
    i0, i1 = (0, xyz.shape[1]) if rows is None else rows
    lib().oracle_cn_pairs_rows(
        C.c_int64(xyz.shape[1]), C.c_int64(i0), C.c_int64(i1), _p(xyz, C.c_double), _p(ty, C.c_int32), C.c_int(len(rl)),
        _p(rl, C.c_int32), _p(L, C.c_double), _p(rc2, C.c_double), _p(cn, C.c_uint64))
    return cn


def rdf_rect(xyz_soa, types, sxyz_soa, stypes, rel, lengths, rc2, ddr, nbins):
    xyz, ty, rl = _prep(xyz_soa, types, rel)
    sx = np.ascontiguousarray(sxyz_soa, dtype=np.float64)
    st = np.ascontiguousarray(stypes, dtype=np.int32)
    L = np.ascontiguousarray(lengths, dtype=np.float64)
    part = np.zeros((len(rl), nbins), dtype=np.uint64)
    ov = C.c_uint64(0)
    lib().oracle_rdf_rect(
        C.c_int64(xyz.shape[1]), _p(xyz, C.c_double), _p(ty, C.c_int32), C.c_int64(sx.shape[1]),
        _p(sx, C.c_double), _p(st, C.c_int32), C.c_int(len(rl)), _p(rl, C.c_int32),
        _p(L, C.c_double), C.c_double(rc2), C.c_double(ddr), C.c_int(nbins),
        _p(part, C.c_uint64), C.byref(ov))
    return part, ov.value


def cn_rect(xyz_soa, types, sxyz_soa, stypes, rel, lengths, rc2_list):
    xyz, ty, rl = _prep(xyz_soa, types, rel)
    sx = np.ascontiguousarray(sxyz_soa, dtype=np.float64)
    st = np.ascontiguousarray(stypes, dtype=np.int32)
    L = np.ascontiguousarray(lengths, dtype=np.float64)
    rc2 = np.ascontiguousarray(rc2_list, dtype=np.float64)
    cn = np.zeros(len(rl), dtype=np.uint64)
    lib().oracle_cn_rect(
        C.c_int64(xyz.shape[1]), _p(xyz, C.c_double), _p(ty, C.c_int32), C.c_int64(sx.shape[1]),
        _p(sx, C.c_double), _p(st, C.c_int32), C.c_int(len(rl)), _p(rl, C.c_int32),
        _p(L, C.c_double), _p(rc2, C.c_double), _p(cn, C.c_uint64))
    return cn


def msd_pairs(r_soa, pairs, group_off):
    """r_soa [F,3,E]; pairs [P,2]; group_off [G+1] -> sums [P,G,4]."""
    r = np.ascontiguousarray(r_soa, dtype=np.float64)
    pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    go = np.ascontiguousarray(group_off, dtype=np.int64)
    out = np.zeros((len(pr), len(go) - 1, 4))
    lib().oracle_msd_pairs(
        C.c_int64(r.shape[2]), _p(r, C.c_double), C.c_int(len(pr)), _p(pr, C.c_int32),
        C.c_int(len(go) - 1), _p(go, C.c_int64), _p(out, C.c_double))
    return out


def xcorr_direct(a, b, n_lags=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n_lags = len(a) if n_lags is None else n_lags
    out = np.zeros(n_lags)
    lib().oracle_xcorr_direct(C.c_int64(len(a)), _p(a, C.c_double), _p(b, C.c_double),
                              C.c_int64(n_lags), _p(out, C.c_double))
    return out


def lag_msd(r_soa, lags, group_off):
    """r_soa [F,3,E]; lags: the lags to evaluate -> means [len(lags), G, 4] over origins and entities."""
    r = np.ascontiguousarray(r_soa, dtype=np.float64)
    lg = np.ascontiguousarray(lags, dtype=np.int32)
    go = np.ascontiguousarray(group_off, dtype=np.int64)
    out = np.zeros((len(lg), len(go) - 1, 4))
    lib().oracle_lag_msd(C.c_int64(r.shape[0]), C.c_int64(r.shape[2]), _p(r, C.c_double), C.c_int(len(lg)),
                         _p(lg, C.c_int32), C.c_int(len(go) - 1), _p(go, C.c_int64), _p(out, C.c_double))
    return out


def rdf_pairs_threaded(xyz_soa, types, rel, lengths, rc2, ddr, nbins, n_threads):
    """One frame split over host threads by head rows (row i costs N-1-i pairs: rows are dealt so that every thread
    gets about the same number of pairs); the ctypes calls release the GIL. Integer sums: same result as rdf_pairs."""
    from concurrent.futures import ThreadPoolExecutor

    n = np.asarray(xyz_soa).shape[1]
    # boundaries b_k with equal pair counts: pairs above row b = (n-b)(n-b-1)/2
    total = n * (n - 1) / 2.0
    bounds = [0]
    for k in range(1, n_threads):
        rem = total * (1.0 - k / n_threads)
        bounds.append(int(n - (1.0 + np.sqrt(1.0 + 8.0 * rem)) / 2.0))
    bounds.append(n)
    with ThreadPoolExecutor(max_workers=n_threads) as pool:
        parts = list(pool.map(lambda k: rdf_pairs(xyz_soa, types, rel, lengths, rc2, ddr, nbins,
                                                  rows=(bounds[k], bounds[k + 1])), range(n_threads)))
    return (sum(p[0] for p in parts), sum(p[1] for p in parts), sum(p[2] for p in parts))
