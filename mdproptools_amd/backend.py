"""
numpy-facing wrappers over the C-ABI, one per reference kernel (the drop-in seam).

Argument meaning follows the reference's private functions
(/root/reference/mdproptools/structural/rdf_cn.py:72-162,
dynamical/diffusion.py:207-238, dynamical/conductivity.py:97-114,216-232,
dynamical/viscosity.py:86-153); layouts are the SoA planes of include/mdhip.h.
Coordinates may be host ndarrays, CUDA/HIP torch tensors (float64, contiguous)
or `_lib.DevPtr` — device-resident inputs are used in place.

Everything here runs on the GPU through libmdhip.so; there is no CPU path.

`async_=True` (where offered) issues the call through its *_async entry point: the device work is queued on the
context's stream and a `_lib.Pending` handle comes back at once; `handle.wait()` completes the call (and every call
issued before it) and returns what the synchronous form returns. Calls issued back to back run back to back on the
GPU, with no host round trip in between.
"""

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Pending, as_input, default_context, ptr, result_array

XCORR_FFT = 0
XCORR_DIRECT = 1


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _dev_out(out, shape, dtype="torch.float64", ctx=None):
    """Checks a device result buffer (`out=`): contiguous CUDA tensor of `shape` and `dtype` on the context's device;
    whatever torch still has queued on that tensor's stream (its allocation's fill, say) is waited for, since the
    context writes it from a stream of its own. Returns its address."""
    if not (getattr(out, "is_cuda", False) and out.is_contiguous() and str(out.dtype) == dtype
            and tuple(out.shape) == tuple(shape)):
        raise ValueError("out must be a contiguous %s CUDA tensor of shape %s" % (dtype, tuple(shape)))
    if ctx is not None:
        if out.device.index is not None and out.device.index != ctx.device:
            raise ValueError("out lives on cuda:%d but the mdhip context is bound to device %d"
                             % (out.device.index, ctx.device))
        import torch

        cur = torch.cuda.current_stream(out.device)
        if getattr(ctx, "_stream", None) != cur.cuda_stream:
            cur.synchronize()
    return C.c_void_p(out.data_ptr())


def _shape3(x, name):
    shp = tuple(x.shape)
    if len(shp) != 3 or shp[1] != 3:
        raise ValueError("%s must have shape [n_frames, 3, n]" % name)
    return shp


def cutoff_sq(r_cut):
    """r_cut**2 as the jitted reference evaluates it: one multiply (numba lowers `x ** 2` with a
    literal exponent to x*x; rdf_cn.py:66). CPython's float pow differs in ~0.1% of inputs."""
    r = float(r_cut)
    return r * r


def rdf_loop(xyz, types, box, relation_matrix, r_cut, ddr, nbins, per_frame=True, ctx=None, edges=None, async_=False):
    """
    `_rdf_loop` (rdf_cn.py:72-97) for every frame of xyz [F,3,N].

    Returns (rdf_full uint64 [F,nbins], rdf_part uint64 [F,R,nbins], overflow) — or the frame sums
    [nbins] / [R,nbins] when per_frame is False.
    """
    ctx = ctx or default_context()
    F, _, N = _shape3(xyz, "xyz")
    xp, on_dev, keep = as_input(xyz, ctx)
    ty = _i32(types)
    stride = 0 if ty.ndim == 1 else N
    if ty.size != (N if stride == 0 else F * N):
        raise ValueError("types must have shape [N] or [F, N]")
    bx = _f64(box).reshape(F, 3)
    rel = _i32(relation_matrix).reshape(-1, 2)
    R = len(rel)
    lead = (F,) if per_frame else ()
    full = np.zeros(lead + (nbins,), dtype=np.uint64)
    part = np.zeros(lead + (R, nbins), dtype=np.uint64)
    ov = C.c_uint64(0)
    ed = None if edges is None else _f64(edges)
    fn = ctx.lib.mdhip_rdf_atomic_async if async_ else ctx.lib.mdhip_rdf_atomic
    ctx.check(fn(
        ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), R, ptr(rel, C.c_int32),
        cutoff_sq(r_cut), float(ddr), int(nbins), None if ed is None else ptr(ed), int(bool(per_frame)),
        ptr(full, C.c_uint64), ptr(part, C.c_uint64), C.byref(ov)))
    if async_:
        return Pending(ctx, (full, part, ov), keep=(keep, ty, bx, rel, ed), finish=lambda r: (r[0], r[1], int(r[2].value)))
    return full, part, int(ov.value)


def rdf_loop_dev(xyz, types, box, relation_matrix, r_cut, ddr, nbins, out, ctx=None, async_=False):
    """
    Frame-summed `_rdf_loop` with the sums left on the device: `out` = contiguous int64 CUDA tensor of
    (1 + R) * nbins + 1 words (rdf_full | rdf_part | overflow; the bit patterns are the uint64 counts), overwritten.
    Used by the multi-GPU layer so that the all-reduce reads the buffer the kernels wrote.
    """
    ctx = ctx or default_context()
    F, _, N = _shape3(xyz, "xyz")
    xp, on_dev, keep = as_input(xyz, ctx)
    ty = _i32(types)
    stride = 0 if ty.ndim == 1 else N
    if ty.size != (N if stride == 0 else F * N):
        raise ValueError("types must have shape [N] or [F, N]")
    bx = _f64(box).reshape(F, 3)
    rel = _i32(relation_matrix).reshape(-1, 2)
    words = (1 + len(rel)) * int(nbins) + 1
    if not (getattr(out, "is_cuda", False) and out.is_contiguous() and str(out.dtype) == "torch.int64"
            and out.numel() == words):
        raise ValueError("out must be a contiguous int64 CUDA tensor of %d words" % words)
    op = _dev_out(out, out.shape, "torch.int64", ctx)
    fn = ctx.lib.mdhip_rdf_atomic_dev_async if async_ else ctx.lib.mdhip_rdf_atomic_dev
    ctx.check(fn(
        ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), len(rel), ptr(rel, C.c_int32),
        cutoff_sq(r_cut), float(ddr), int(nbins), None, op))
    if async_:
        return Pending(ctx, out, keep=(keep, ty, bx, rel))
    return out


def rdf_cn_loop(xyz, types, box, relation_matrix, r_cut, ddr, nbins, cn_cut_list, per_frame=True, ctx=None,
                async_=False):
    """
    `_rdf_loop` and `_cn_loop` (rdf_cn.py:72-119) from ONE sweep over the pairs: returns
    (rdf_full, rdf_part, overflow, cn) — the integers of rdf_loop(...) and cn_loop(..., cn_cut_list).
    """
    ctx = ctx or default_context()
    F, _, N = _shape3(xyz, "xyz")
    xp, on_dev, keep = as_input(xyz, ctx)
    ty = _i32(types)
    stride = 0 if ty.ndim == 1 else N
    if ty.size != (N if stride == 0 else F * N):
        raise ValueError("types must have shape [N] or [F, N]")
    bx = _f64(box).reshape(F, 3)
    rel = _i32(relation_matrix).reshape(-1, 2)
    R = len(rel)
    rc2 = _f64([cutoff_sq(r) for r in cn_cut_list])
    if len(rc2) != R:
        raise ValueError("one coordination cutoff per relation is required")
    lead = (F,) if per_frame else ()
    full = np.zeros(lead + (nbins,), dtype=np.uint64)
    part = np.zeros(lead + (R, nbins), dtype=np.uint64)
    cn = np.zeros(lead + (R,), dtype=np.uint64)
    ov = C.c_uint64(0)
    fn = ctx.lib.mdhip_rdf_cn_atomic_async if async_ else ctx.lib.mdhip_rdf_cn_atomic
    ctx.check(fn(
        ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), R, ptr(rel, C.c_int32),
        cutoff_sq(r_cut), float(ddr), int(nbins), None, ptr(rc2), int(bool(per_frame)),
        ptr(full, C.c_uint64), ptr(part, C.c_uint64), C.byref(ov), ptr(cn, C.c_uint64)))
    if async_:
        return Pending(ctx, (full, part, ov, cn), keep=(keep, ty, bx, rel, rc2),
                       finish=lambda r: (r[0], r[1], int(r[2].value), r[3]))
    return full, part, int(ov.value), cn


def cn_loop(xyz, types, box, relation_matrix, r_cut_list, per_frame=True, ctx=None, out=None, async_=False):
    """`_cn_loop` (rdf_cn.py:100-119): raw counts uint64 [F,R] (or [R]). `out` (frame-summed only): an int64 CUDA
    tensor [R] that receives the counts on the device (their bit patterns; the multi-GPU layer all-reduces it)."""
    ctx = ctx or default_context()
    F, _, N = _shape3(xyz, "xyz")
    xp, on_dev, keep = as_input(xyz, ctx)
    ty = _i32(types)
    stride = 0 if ty.ndim == 1 else N
    bx = _f64(box).reshape(F, 3)
    rel = _i32(relation_matrix).reshape(-1, 2)
    rc2 = _f64([cutoff_sq(r) for r in r_cut_list])
    if len(rc2) != len(rel):
        raise ValueError("one cutoff per relation is required")
    if out is not None:
        if per_frame:
            raise ValueError("a device result buffer holds the frame-summed counts: pass per_frame=False")
        op = _dev_out(out, (len(rel),), "torch.int64", ctx)
        if async_:
            ctx.check(ctx.lib.mdhip_cn_atomic_async(
                ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), len(rel), ptr(rel, C.c_int32),
                ptr(rc2), 0, op, 1))
            return Pending(ctx, out, keep=(keep, ty, bx, rel, rc2))
        ctx.check(ctx.lib.mdhip_cn_atomic_dev(
            ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), len(rel), ptr(rel, C.c_int32),
            ptr(rc2), op))
        return out
    cn = np.zeros(((F,) if per_frame else ()) + (len(rel),), dtype=np.uint64)
    if async_:
        ctx.check(ctx.lib.mdhip_cn_atomic_async(
            ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), len(rel), ptr(rel, C.c_int32),
            ptr(rc2), int(bool(per_frame)), C.c_void_p(cn.ctypes.data), 0))
        return Pending(ctx, cn, keep=(keep, ty, bx, rel, rc2))
    ctx.check(ctx.lib.mdhip_cn_atomic(
        ctx.h, F, N, xp, on_dev, ptr(ty, C.c_int32), stride, ptr(bx), len(rel), ptr(rel, C.c_int32),
        ptr(rc2), int(bool(per_frame)), ptr(cn, C.c_uint64)))
    return cn


def rdf_mol_loop(xyz, types, sites, site_types, box, relation_matrix, r_cut, ddr, nbins, per_frame=True,
                 ctx=None):
    """`_rdf_mol_loop` (rdf_cn.py:122-141): atoms [F,3,N] x sites [F,3,M] -> (rdf_part, overflow)."""
    ctx = ctx or default_context()
    F, _, N = _shape3(xyz, "xyz")
    F2, _, M = _shape3(sites, "sites")
    if F2 != F:
        raise ValueError("xyz and sites must have the same number of frames")
    xp, x_dev, k1 = as_input(xyz, ctx)
    sp, s_dev, k2 = as_input(sites, ctx)
    ty, st = _i32(types), _i32(site_types)
    bx = _f64(box).reshape(F, 3)
    rel = _i32(relation_matrix).reshape(-1, 2)
    part = np.zeros(((F,) if per_frame else ()) + (len(rel), nbins), dtype=np.uint64)
    ov = C.c_uint64(0)
    ctx.check(ctx.lib.mdhip_rdf_sites(
        ctx.h, F, N, xp, x_dev, ptr(ty, C.c_int32), M, sp, s_dev, ptr(st, C.c_int32), ptr(bx), len(rel),
        ptr(rel, C.c_int32), cutoff_sq(r_cut), float(ddr), int(nbins), None, int(bool(per_frame)),
        ptr(part, C.c_uint64), C.byref(ov)))
    return part, int(ov.value)


def cn_mol_loop(xyz, types, sites, site_types, box, relation_matrix, r_cut_list, per_frame=True, ctx=None):
    """`_cn_mol_loop` (rdf_cn.py:144-162)."""
    ctx = ctx or default_context()
    F, _, N = _shape3(xyz, "xyz")
    _, _, M = _shape3(sites, "sites")
    xp, x_dev, k1 = as_input(xyz, ctx)
    sp, s_dev, k2 = as_input(sites, ctx)
    ty, st = _i32(types), _i32(site_types)
    bx = _f64(box).reshape(F, 3)
    rel = _i32(relation_matrix).reshape(-1, 2)
    rc2 = _f64([cutoff_sq(r) for r in r_cut_list])
    cn = np.zeros(((F,) if per_frame else ()) + (len(rel),), dtype=np.uint64)
    ctx.check(ctx.lib.mdhip_cn_sites(
        ctx.h, F, N, xp, x_dev, ptr(ty, C.c_int32), M, sp, s_dev, ptr(st, C.c_int32), ptr(bx), len(rel),
        ptr(rel, C.c_int32), ptr(rc2), int(bool(per_frame)), ptr(cn, C.c_uint64)))
    return cn


def segment_com(attr, atom_mass, seg_off, atom_q=None, out=None, ctx=None, async_=False):
    """
    `calc_com` / `_define_mol_cols` arithmetic (com_mols.py:58-60, rdf_cn.py:233-238):
    attr [F,K,N] -> com [F,K,M], plus seg_mass [M] and seg_q [M] (None without atom_q).
    `out` may be a device tensor [F,K,M] to keep the result on the GPU.
    """
    ctx = ctx or default_context()
    shp = tuple(attr.shape)
    if len(shp) != 3:
        raise ValueError("attr must have shape [n_frames, n_attr, n_atoms]")
    F, K, N = shp
    ap, a_dev, keep = as_input(attr, ctx)
    m = _f64(atom_mass)
    off = _i64(seg_off)
    M = len(off) - 1
    q = None if atom_q is None else _f64(atom_q)
    seg_mass = np.zeros(M)
    seg_q = None if q is None else np.zeros(M)
    if out is None:
        res = result_array((F, K, M), device=ctx.device)
        op, o_dev = C.c_void_p(res.ctypes.data), 0
    else:
        res = out
        op, o_dev, _k = as_input(out, ctx)
    fn = ctx.lib.mdhip_segment_com_async if async_ else ctx.lib.mdhip_segment_com
    ctx.check(fn(
        ctx.h, F, N, K, ap, a_dev, ptr(m), None if q is None else ptr(q), M, ptr(off, C.c_int64), op,
        o_dev, ptr(seg_mass), None if seg_q is None else ptr(seg_q)))
    if async_:
        return Pending(ctx, (res, seg_mass, seg_q), keep=(keep, m, off, q))
    return res, seg_mass, seg_q


def msd_pairs(r, pairs, group_off, scale=1.0, per_entity=False, ctx=None, out=None):
    """
    Frame-pair displacement sums (diffusion.py:212-218): r [F,3,E], pairs [P,2] ->
    sums [P,G,4] (+ per-entity rows [P,E,4] when requested). `out`: a float64 CUDA tensor [P,G,4] that receives the
    sums on the device (no per-entity rows then).
    """
    ctx = ctx or default_context()
    F, _, E = _shape3(r, "r")
    rp, on_dev, keep = as_input(r, ctx)
    pr = _i32(pairs).reshape(-1, 2)
    go = _i64(group_off)
    G = len(go) - 1
    if out is not None:
        if per_entity:
            raise ValueError("per-entity rows are not available with a device result buffer")
        ctx.check(ctx.lib.mdhip_msd_pairs_dev(
            ctx.h, F, E, rp, on_dev, float(scale), len(pr), ptr(pr, C.c_int32), G, ptr(go, C.c_int64),
            _dev_out(out, (len(pr), G, 4))))
        return out
    sums = np.zeros((len(pr), G, 4))
    pe = np.zeros((len(pr), E, 4)) if per_entity else None
    ctx.check(ctx.lib.mdhip_msd_pairs(
        ctx.h, F, E, rp, on_dev, float(scale), len(pr), ptr(pr, C.c_int32), G, ptr(go, C.c_int64),
        ptr(sums), None if pe is None else C.c_void_p(pe.ctypes.data), 0))
    return (sums, pe) if per_entity else sums


def msd_origin(r, origin, group_off, scale=1.0, cols=None, out=None, ctx=None, async_=False):
    """
    Single-origin MSD of a FRAME SHARD (diffusion.py:212-218 with the frames dealt to one process per GPU): every
    frame of r [F,3,E] against `origin` [3,E] (host array or CUDA tensor — the time-0 frame, broadcast by its owner).
    Returns sums [F,G,4]: a host array, or `out` (float64 CUDA tensor [F,G,4]) when given. `cols`: the per-entity
    columns dx2, dy2, dz2, msd, each [F*E] — a host float64 array [4, F*E] with contiguous rows or a CUDA tensor.
    """
    ctx = ctx or default_context()
    F, _, E = _shape3(r, "r")
    rp, on_dev, keep = as_input(r, ctx)
    if tuple(origin.shape) != (3, E):
        raise ValueError("origin must have shape [3, n_ent]")
    op, o_dev, keep2 = as_input(origin, ctx)
    go = _i64(group_off)
    G = len(go) - 1
    if out is None:
        sums = np.zeros((F, G, 4))
        sp, s_dev = C.c_void_p(sums.ctypes.data), 0
    else:
        sums, sp, s_dev = out, _dev_out(out, (F, G, 4), ctx=ctx), 1
    cp, c_dev, stride = None, 0, 0
    if cols is not None:
        if getattr(cols, "is_cuda", False):
            cp, c_dev, stride = _dev_out(cols, (4, F * E), ctx=ctx), 1, F * E
        else:
            if not (isinstance(cols, np.ndarray) and cols.dtype == np.float64 and cols.shape == (4, F * E)
                    and (F * E == 0 or (cols.strides[1] == 8 and cols.strides[0] % 8 == 0
                                        and cols.strides[0] >= 8 * F * E))):
                raise ValueError("cols must be a float64 array [4, n_frames * n_ent] with contiguous rows")
            cp, stride = C.c_void_p(cols.ctypes.data), (cols.strides[0] // 8 if F * E else 0)
    fn = ctx.lib.mdhip_msd_origin_async if async_ else ctx.lib.mdhip_msd_origin
    ctx.check(fn(ctx.h, F, E, rp, on_dev, op, o_dev, float(scale), G, ptr(go, C.c_int64), sp, s_dev, cp, stride, c_dev))
    if async_:
        return Pending(ctx, sums, keep=(keep, keep2, go, cols))
    return sums


def msd_pairs_cols(r, pairs, group_off, cols, scale=1.0, ctx=None):
    """
    `msd_pairs` with the per-entity values written as the four COLUMNS dx2, dy2, dz2, msd of `cols` — a host float64
    array [4, P*E] whose rows are contiguous (they may be rows of a larger C-ordered block: the block a DataFrame
    wraps without copying). Returns sums [P,G,4].
    """
    ctx = ctx or default_context()
    F, _, E = _shape3(r, "r")
    rp, on_dev, keep = as_input(r, ctx)
    pr = _i32(pairs).reshape(-1, 2)
    go = _i64(group_off)
    G = len(go) - 1
    if not (isinstance(cols, np.ndarray) and cols.dtype == np.float64 and cols.shape == (4, len(pr) * E)
            and cols.strides[1] == 8 and cols.strides[0] % 8 == 0 and cols.strides[0] >= 8 * len(pr) * E):
        raise ValueError("cols must be a float64 array [4, n_pairs * n_ent] with contiguous rows")
    sums = np.zeros((len(pr), G, 4))
    ctx.check(ctx.lib.mdhip_msd_pairs_cols(
        ctx.h, F, E, rp, on_dev, float(scale), len(pr), ptr(pr, C.c_int32), G, ptr(go, C.c_int64),
        ptr(sums), C.c_void_p(cols.ctypes.data), cols.strides[0] // 8, 0))
    return sums


def msd_windows(r, tao, scale=1.0, ctx=None, out=None, async_=False):
    """Fixed-lag window sums per entity (diffusion.py:225-237): r [F,3,E] -> [E,4] (`out`: float64 CUDA tensor [E,4])."""
    ctx = ctx or default_context()
    F, _, E = _shape3(r, "r")
    rp, on_dev, keep = as_input(r, ctx)
    if out is not None:
        op = _dev_out(out, (E, 4), ctx=ctx)
        if async_:
            ctx.check(ctx.lib.mdhip_msd_windows_async(ctx.h, F, E, rp, on_dev, float(scale), int(tao), op, 1))
            return Pending(ctx, out, keep=keep)
        ctx.check(ctx.lib.mdhip_msd_windows_dev(ctx.h, F, E, rp, on_dev, float(scale), int(tao), op))
        return out
    out = np.zeros((E, 4))
    if async_:
        ctx.check(ctx.lib.mdhip_msd_windows_async(ctx.h, F, E, rp, on_dev, float(scale), int(tao),
                                                  C.c_void_p(out.ctypes.data), 0))
        return Pending(ctx, out, keep=keep)
    ctx.check(ctx.lib.mdhip_msd_windows(ctx.h, F, E, rp, on_dev, float(scale), int(tao), ptr(out)))
    return out


def lag_msd(r, max_lag, group_off, scale=1.0, ctx=None, out=None, async_=False, status_out=None):
    """Full lag average (superset): r [F,3,E] -> [max_lag+1, G, 4] (`out`: float64 CUDA tensor of that shape).
    `status_out` (asynchronous call with a device result only): a float64 CUDA tensor whose first element receives the
    call's status on the device, behind its kernels (mdhip_lag_msd_status_dev: the spectral path's error bound, +inf
    when its result will be rewritten at completion, 0 for the exact path)."""
    ctx = ctx or default_context()
    F, _, E = _shape3(r, "r")
    rp, on_dev, keep = as_input(r, ctx)
    go = _i64(group_off)
    G = len(go) - 1
    if out is not None:
        op = _dev_out(out, (int(max_lag) + 1, G, 4), ctx=ctx)
        if async_:
            ctx.check(ctx.lib.mdhip_lag_msd_async(ctx.h, F, E, rp, on_dev, float(scale), int(max_lag), G,
                                                  ptr(go, C.c_int64), op, 1))
            pend = Pending(ctx, out, keep=(keep, go, status_out))
            if status_out is not None:
                ctx.check(ctx.lib.mdhip_lag_msd_status_dev(ctx.h, _dev_out(status_out, tuple(status_out.shape), ctx=ctx)))
            return pend
        ctx.check(ctx.lib.mdhip_lag_msd_dev(ctx.h, F, E, rp, on_dev, float(scale), int(max_lag), G,
                                            ptr(go, C.c_int64), op))
        ctx.note_fallbacks()
        return out
    out = np.zeros((int(max_lag) + 1, G, 4))
    if async_:
        ctx.check(ctx.lib.mdhip_lag_msd_async(ctx.h, F, E, rp, on_dev, float(scale), int(max_lag), G,
                                              ptr(go, C.c_int64), C.c_void_p(out.ctypes.data), 0))
        return Pending(ctx, out, keep=(keep, go))
    ctx.check(ctx.lib.mdhip_lag_msd(ctx.h, F, E, rp, on_dev, float(scale), int(max_lag), G,
                                    ptr(go, C.c_int64), ptr(out)))
    ctx.note_fallbacks()
    return out


def charge_flux(vel, atom_mass, atom_q, seg_off, seg_type, n_types, vel_conv, charge_conv, ctx=None, out=None,
                async_=False):
    """`conductivity_loop` for every frame (_conductivity.py:11-35): vel [F,3,N] -> j [3,T,F] (`out`: float64 CUDA
    tensor of that shape)."""
    ctx = ctx or default_context()
    F, _, N = _shape3(vel, "vel")
    vp_, on_dev, keep = as_input(vel, ctx)
    m, q = _f64(atom_mass), _f64(atom_q)
    off = _i64(seg_off)
    st = _i32(seg_type)
    dev = out is not None
    if dev:
        op = _dev_out(out, (3, int(n_types), F), ctx=ctx)
    else:
        out = np.zeros((3, int(n_types), F))
        op = C.c_void_p(out.ctypes.data)
    if async_:
        ctx.check(ctx.lib.mdhip_charge_flux_async(
            ctx.h, F, N, vp_, on_dev, ptr(m), ptr(q), len(off) - 1, ptr(off, C.c_int64), ptr(st, C.c_int32),
            int(n_types), float(vel_conv), float(charge_conv), op, int(dev)))
        return Pending(ctx, out, keep=(keep, m, q, off, st))
    if dev:
        ctx.check(ctx.lib.mdhip_charge_flux_dev(
            ctx.h, F, N, vp_, on_dev, ptr(m), ptr(q), len(off) - 1, ptr(off, C.c_int64), ptr(st, C.c_int32),
            int(n_types), float(vel_conv), float(charge_conv), op))
        return out
    ctx.check(ctx.lib.mdhip_charge_flux(
        ctx.h, F, N, vp_, on_dev, ptr(m), ptr(q), len(off) - 1, ptr(off, C.c_int64), ptr(st, C.c_int32),
        int(n_types), float(vel_conv), float(charge_conv), ptr(out)))
    return out


def xcorr(a, b=None, method=XCORR_FFT, n_lags=None, ctx=None, lag_begin=0, out=None, async_=False):
    """
    c[p][k] = sum_t a_p[t+k] b_p[t] / (n-k) (conductivity.py:109-114, viscosity.py:103-115).
    a, b: [n] or [P,n]; b=None gives the autocorrelation. lag_begin > 0 (direct method): the lags
    lag_begin .. lag_begin + n_lags - 1 only. `out`: float64 CUDA tensor [P, n_lags] that receives the lags on the
    device (returned as it is).
    """
    ctx = ctx or default_context()
    single = len(a.shape) == 1
    ap, a_dev, k1 = as_input(a, ctx)
    k2 = None  # (the converted copies as_input hands the library must live until the call has completed)
    shp = tuple(a.shape)
    P, n = (1, shp[0]) if single else shp
    if b is None:
        bp, b_dev = ap, a_dev
    else:
        bp, b_dev, k2 = as_input(b, ctx)
        if b_dev != a_dev:
            raise ValueError("a and b must both be host arrays or both device tensors")
    n_lags = n - int(lag_begin) if n_lags is None else int(n_lags)
    if out is not None:
        fn = ctx.lib.mdhip_xcorr_lags_dev_async if async_ else ctx.lib.mdhip_xcorr_lags_dev
        ctx.check(fn(ctx.h, n, P, ap, bp, a_dev, int(method), int(lag_begin), n_lags, _dev_out(out, (P, n_lags), ctx=ctx)))
        return Pending(ctx, out, keep=(a, b, k1, k2)) if async_ else out
    out = result_array((P, n_lags), device=ctx.device)
    if async_:
        if lag_begin:
            raise ValueError("a lag range is asynchronous only with a device result buffer")
        ctx.check(ctx.lib.mdhip_xcorr_async(ctx.h, n, P, ap, bp, a_dev, int(method), n_lags, ptr(out)))
        return Pending(ctx, out[0] if single else out, keep=(a, b, k1, k2))
    ctx.check(ctx.lib.mdhip_xcorr_lags(ctx.h, n, P, ap, bp, a_dev, int(method), int(lag_begin), n_lags, ptr(out)))
    return out[0] if single else out


def cumtrapz(y, dx, leading_zero=False, ctx=None, out=None, async_=False):
    """Cumulative trapezoid (viscosity.py:151, conductivity.py:231): y [n] or [S,n] (`out`: float64 CUDA tensor
    [S, n-1 (+1)] that receives the integrals on the device)."""
    ctx = ctx or default_context()
    single = len(y.shape) == 1
    yp, on_dev, keep = as_input(y, ctx)
    shp = tuple(y.shape)
    S, n = (1, shp[0]) if single else shp
    m = n - 1 + (1 if leading_zero else 0)
    if out is not None:
        fn = ctx.lib.mdhip_cumtrapz_dev_async if async_ else ctx.lib.mdhip_cumtrapz_dev
        ctx.check(fn(ctx.h, n, S, yp, on_dev, float(dx), int(bool(leading_zero)), _dev_out(out, (S, max(m, 0)), ctx=ctx)))
        return Pending(ctx, out, keep=keep) if async_ else out
    out = result_array((S, max(m, 0)), device=ctx.device)
    fn = ctx.lib.mdhip_cumtrapz_async if async_ else ctx.lib.mdhip_cumtrapz
    ctx.check(fn(ctx.h, n, S, yp, on_dev, float(dx), int(bool(leading_zero)), ptr(out)))
    res = out[0] if single else out
    return Pending(ctx, res, keep=keep) if async_ else res


def green_kubo(a, b=None, method=XCORR_FFT, acf_scale=1.0, dx=1.0, integral_scale=1.0, leading_zero=False,
               want_acf=True, want_mean=False, ctx=None, async_=False):
    """
    The Green-Kubo chain in ONE call, nothing but the results crossing the bus (mdhip_green_kubo):
      acf      = xcorr(a, b) * acf_scale                      viscosity.py:178-184, conductivity.py:109-114
      integral = integral_scale * cumtrapz(acf, dx)           viscosity.py:151-152, conductivity.py:229-231
      mean     = integral.mean(axis=0)                        viscosity.py:189
    a, b: [S, n] host arrays or CUDA tensors (b=None: autocorrelation). Returns (acf [S,n] or None, integral
    [S, n-1 (+1)], mean [n-1 (+1)] or None); large results are page-locked arrays (written by DMA).
    """
    ctx = ctx or default_context()
    ap, a_dev, k1 = as_input(a, ctx)
    S, n = tuple(a.shape)
    if b is None:
        bp, k2 = ap, None
    else:
        bp, b_dev, k2 = as_input(b, ctx)
        if b_dev != a_dev:
            raise ValueError("a and b must both be host arrays or both device tensors")
    m = n - 1 + (1 if leading_zero else 0)
    acf = result_array((S, n), device=ctx.device) if want_acf else None
    integral = result_array((S, max(m, 0)), device=ctx.device)
    mean = result_array((max(m, 0),), device=ctx.device) if want_mean else None
    fn = ctx.lib.mdhip_green_kubo_async if async_ else ctx.lib.mdhip_green_kubo
    ctx.check(fn(ctx.h, n, S, ap, bp, a_dev, int(method), float(acf_scale), float(dx), float(integral_scale),
                 int(bool(leading_zero)), None if acf is None else ptr(acf), ptr(integral),
                 None if mean is None else ptr(mean)))
    res = (acf, integral, mean)
    return Pending(ctx, res, keep=(k1, k2)) if async_ else res


bin_edges = _lib.bin_edges


def shell_residence(xyz_i, xyz_j, box, r_lo_sq, r_hi_sq, exclude_diagonal=False, ctx=None):
    """
    Numerators of the neighbour-shell autocovariance (residence_time.py:96-131): central atoms xyz_i [F,3,Ni],
    shell atoms xyz_j [F,3,Nj] -> (counts uint64 [F], number of in-shell records).
    counts[k] = sum_{i,j,t} h_ij(t) h_ij(t+k), h = (rsq > r_lo_sq) & (rsq <= r_hi_sq).
    """
    ctx = ctx or default_context()
    F, _, Ni = _shape3(xyz_i, "xyz_i")
    F2, _, Nj = _shape3(xyz_j, "xyz_j")
    if F2 != F:
        raise ValueError("xyz_i and xyz_j must have the same number of frames")
    ip, i_dev, k1 = as_input(xyz_i, ctx)
    jp, j_dev, k2 = as_input(xyz_j, ctx)
    bx = _f64(box).reshape(F, 3)
    counts = np.zeros(F, dtype=np.uint64)
    nrec = C.c_uint64(0)
    ctx.check(ctx.lib.mdhip_shell_residence(
        ctx.h, F, Ni, ip, i_dev, Nj, jp, j_dev, ptr(bx), float(r_lo_sq), float(r_hi_sq),
        int(bool(exclude_diagonal)), ptr(counts, C.c_uint64), C.byref(nrec)))
    return counts, int(nrec.value)
