"""
ctypes binding of libmdhip.so (C-ABI: include/mdhip.h).

This is the only place the package touches native code. There is no CPU
fallback: if the library is missing it is built (hipcc must be present), and
if no gfx950 device is usable `Context()` raises `MdhipError`.
"""

import ctypes as C
import os
import threading

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdhip.so")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)
c_up = C.POINTER(C.c_uint64)
vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/mdhip.h declares
PROTOTYPES = {
    "mdhip_version": (C.c_int, []),
    "mdhip_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "mdhip_destroy": (None, [vp]),
    "mdhip_last_error": (C.c_char_p, [vp]),
    "mdhip_set_stream": (C.c_int, [vp, vp]),
    "mdhip_sync": (C.c_int, [vp]),
    "mdhip_wait": (C.c_int, [vp, C.c_int]),
    "mdhip_pending": (C.c_int, [vp]),
    "mdhip_call_stats": (C.c_int, [vp, C.c_int, c_dp, c_dp, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "mdhip_last_ticket": (C.c_longlong, [vp]),
    "mdhip_ticket_stats": (C.c_int, [vp, C.c_longlong, c_dp, c_dp, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "mdhip_ticket_status": (C.c_int, [vp, C.c_longlong, C.POINTER(C.c_int)]),
    "mdhip_fallbacks": (C.c_longlong, [vp]),
    "mdhip_last_kernel_ms": (C.c_double, [vp, C.POINTER(C.c_int)]),
    "mdhip_last_aux_ms": (C.c_double, [vp]),
    "mdhip_last_kernel_name": (C.c_char_p, [vp]),
    "mdhip_last_rel_bound": (C.c_double, [vp]),
    "mdhip_build_id": (C.c_char_p, []),
    "mdhip_device_name": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "mdhip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp)]),
    "mdhip_host_alloc_on": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
    "mdhip_host_free": (None, [vp]),
    "mdhip_set_option": (C.c_int, [vp, C.c_char_p, C.c_int]),
    "mdhip_bin_edges": (C.c_int, [C.c_double, C.c_int, c_dp]),
    "mdhip_pk_error_bound": (C.c_double, [C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double]),
    "mdhip_row_displacement": (C.c_int, [C.c_int, C.c_int, c_ip, c_ip, c_ip, c_ip, C.POINTER(C.c_int)]),
    "mdhip_rdf_atomic": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                   c_ip, C.c_double, C.c_double, C.c_int, c_dp, C.c_int, c_up, c_up, c_up]),
    "mdhip_rdf_atomic_dev": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                       c_ip, C.c_double, C.c_double, C.c_int, c_dp, vp]),
    "mdhip_rdf_cn_atomic": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                      c_ip, C.c_double, C.c_double, C.c_int, c_dp, c_dp, C.c_int, c_up, c_up, c_up,
                                      c_up]),
    "mdhip_cn_atomic": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                  c_ip, c_dp, C.c_int, c_up]),
    "mdhip_cn_atomic_dev": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                      c_ip, c_dp, vp]),
    "mdhip_rdf_sites": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, vp, C.c_int, c_ip,
                                  c_dp, C.c_int, c_ip, C.c_double, C.c_double, C.c_int, c_dp, C.c_int, c_up,
                                  c_up]),
    "mdhip_cn_sites": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, vp, C.c_int, c_ip,
                                 c_dp, C.c_int, c_ip, c_dp, C.c_int, c_up]),
    "mdhip_segment_com": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int, c_dp, c_dp, C.c_int64,
                                    c_lp, vp, C.c_int, c_dp, c_dp]),
    "mdhip_msd_pairs": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, c_ip, C.c_int,
                                  c_lp, c_dp, vp, C.c_int]),
    "mdhip_msd_pairs_cols": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, c_ip, C.c_int,
                                       c_lp, c_dp, vp, C.c_int64, C.c_int]),
    "mdhip_msd_origin": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, vp, C.c_int, C.c_double, C.c_int, c_lp,
                                   vp, C.c_int, vp, C.c_int64, C.c_int]),
    "mdhip_msd_pairs_dev": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, c_ip, C.c_int,
                                      c_lp, vp]),
    "mdhip_msd_windows": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, c_dp]),
    "mdhip_msd_windows_dev": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, vp]),
    "mdhip_lag_msd_dev": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, C.c_int, c_lp,
                                    vp]),
    "mdhip_charge_flux_dev": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_dp, c_dp, C.c_int64, c_lp, c_ip,
                                        C.c_int, C.c_double, C.c_double, vp]),
    "mdhip_xcorr_lags_dev": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64, vp]),
    "mdhip_lag_msd": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, C.c_int, c_lp,
                                c_dp]),
    "mdhip_charge_flux": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_dp, c_dp, C.c_int64, c_lp, c_ip,
                                    C.c_int, C.c_double, C.c_double, c_dp]),
    "mdhip_xcorr": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int64, c_dp]),
    "mdhip_xcorr_lags": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64, c_dp]),
    "mdhip_cumtrapz": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int, C.c_double, C.c_int, c_dp]),
    "mdhip_cumtrapz_dev": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int, C.c_double, C.c_int, vp]),
    "mdhip_cumtrapz_async": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int, C.c_double, C.c_int, c_dp]),
    "mdhip_cumtrapz_dev_async": (C.c_int, [vp, C.c_int64, C.c_int, vp, C.c_int, C.c_double, C.c_int, vp]),
    "mdhip_green_kubo": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_double, C.c_int, c_dp, c_dp, c_dp]),
    "mdhip_green_kubo_async": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.c_double, C.c_int, c_dp, c_dp, c_dp]),
    "mdhip_rdf_atomic_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                         c_ip, C.c_double, C.c_double, C.c_int, c_dp, C.c_int, c_up, c_up, c_up]),
    "mdhip_rdf_atomic_dev_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                             c_ip, C.c_double, C.c_double, C.c_int, c_dp, vp]),
    "mdhip_rdf_cn_atomic_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                            c_ip, C.c_double, C.c_double, C.c_int, c_dp, c_dp, C.c_int, c_up, c_up, c_up,
                                            c_up]),
    "mdhip_cn_atomic_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_ip, C.c_int64, c_dp, C.c_int,
                                        c_ip, c_dp, C.c_int, vp, C.c_int]),
    "mdhip_segment_com_async": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int, c_dp, c_dp, C.c_int64,
                                          c_lp, vp, C.c_int, c_dp, c_dp]),
    "mdhip_msd_origin_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, vp, C.c_int, C.c_double, C.c_int, c_lp,
                                         vp, C.c_int, vp, C.c_int64, C.c_int]),
    "mdhip_msd_pairs_dev_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, c_ip, C.c_int,
                                            c_lp, vp]),
    "mdhip_msd_windows_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, vp, C.c_int]),
    "mdhip_lag_msd_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_double, C.c_int, C.c_int, c_lp,
                                      vp, C.c_int]),
    "mdhip_lag_msd_status_dev": (C.c_int, [vp, vp]),
    "mdhip_charge_flux_async": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, c_dp, c_dp, C.c_int64, c_lp, c_ip,
                                          C.c_int, C.c_double, C.c_double, vp, C.c_int]),
    "mdhip_xcorr_async": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int64, c_dp]),
    "mdhip_xcorr_lags_dev_async": (C.c_int, [vp, C.c_int64, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                             vp]),
    "mdhip_shell_residence": (C.c_int, [vp, C.c_int64, C.c_int64, vp, C.c_int, C.c_int64, vp, C.c_int, c_dp,
                                        C.c_double, C.c_double, C.c_int, c_up, c_up]),
    "mdhip_dump_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "mdhip_dump_close": (None, [vp]),
    "mdhip_dump_error": (C.c_char_p, [vp]),
    "mdhip_dump_n_frames": (C.c_int64, [vp]),
    "mdhip_dump_frame_info": (C.c_int, [vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_dp, c_dp,
                                        C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "mdhip_dump_read": (C.c_int, [vp, C.c_int64, C.c_int, c_ip, C.c_int, c_dp, C.c_int]),
    "mdhip_dump_read_cols": (C.c_int, [vp, C.c_int64, C.c_int, c_ip, C.c_int, C.POINTER(c_dp), C.c_int]),
    "mdhip_dump_read_files": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.c_char_p,
                                        C.c_int64, C.POINTER(c_dp), c_lp, c_lp, c_dp, c_dp, c_ip, C.c_int, C.c_char_p,
                                        C.c_int, C.c_int, c_dp, c_ip]),
    "mdhip_log_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "mdhip_log_close": (None, [vp]),
    "mdhip_log_error": (C.c_char_p, [vp]),
    "mdhip_log_n_runs": (C.c_int64, [vp]),
    "mdhip_log_run_info": (C.c_int, [vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.c_char_p, C.c_int]),
    "mdhip_log_read": (C.c_int, [vp, C.c_int64, c_dp, c_ip, C.c_int]),
}


class MdhipError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libmdhip error %d: %s" % (code, text))
        self.code = code


_lib = None
_lock = threading.Lock()
STRICT = True  # every symbol include/mdhip.h declares must be there (A/B tools that load older builds clear this)


def _preload_torch_runtime():
    """
    PyTorch-ROCm wheels bundle their own libamdhip64.so.7. Two HIP runtimes in one
    process cannot both own the device ("No HIP GPUs are available" from whichever initialises
    second), so when torch is installed its runtime is loaded FIRST and libmdhip.so then binds to
    the same copy by SONAME. Without torch the system ROCm in /opt/rocm/lib is used (RUNPATH).
    """
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def load(build_if_missing=True):
    """Load libmdhip.so (building it in-tree first if it is missing) and bind every prototype."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        _preload_torch_runtime()
        path = os.environ.get("MDHIP_LIB") or LIB_PATH  # (MDHIP_LIB: another BUILD of this library — build.build_variant)
        if not os.path.exists(path):
            if not build_if_missing or path != LIB_PATH:
                raise MdhipError(-4, "libmdhip.so not found at %s" % path)
            _build.build()
        lib = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            if not STRICT and not hasattr(lib, name):
                continue  # (tools/ab_libs*.py comparing an OLDER build of the library, which lacks newer entry points)
            fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a, ctype=C.c_double):
    return a.ctypes.data_as(C.POINTER(ctype))


class DevPtr:
    """A device pointer the caller owns (e.g. torch tensor .data_ptr()); passed with on_device=1."""

    def __init__(self, address, keepalive=None):
        self.address = int(address)
        self.keepalive = keepalive


def as_input(a, ctx=None):
    """-> (void pointer, on_device flag, keepalive) for a host ndarray, a torch tensor or a DevPtr.
    With `ctx`, a CUDA tensor that lives on another device than the context's is refused (its pointer would be
    dereferenced on the wrong GPU)."""
    if isinstance(a, DevPtr):
        return vp(a.address), 1, a
    if hasattr(a, "data_ptr") and hasattr(a, "is_cuda"):  # torch tensor without importing torch here
        if not a.is_contiguous() or str(a.dtype) != "torch.float64":
            raise ValueError("device tensors must be contiguous float64")
        if a.is_cuda:
            if ctx is not None and a.device.index is not None and a.device.index != ctx.device:
                raise ValueError("tensor lives on cuda:%d but the mdhip context is bound to device %d"
                                 % (a.device.index, ctx.device))
            # the context launches on its own non-blocking stream: whatever torch still has queued to produce
            # this tensor must have finished first
            import torch

            cur = torch.cuda.current_stream(a.device)
            if ctx is None or getattr(ctx, "_stream", None) != cur.cuda_stream:
                cur.synchronize()  # (a context that launches on that very stream is ordered behind it anyway)
            return vp(a.data_ptr()), 1, a
        a = a.numpy()
    arr = _f64(a)
    return vp(arr.ctypes.data), 0, arr


class PinnedPool:
    """
    Page-locked host arrays for LARGE results (include/mdhip.h: a destination in page-locked memory is written by DMA at
    the PCIe rate; a pageable one goes through the runtime's bounce buffers at a third of it and blocks the host). The
    memory comes from mdhip_host_alloc_on and goes back to a free list when the last array that views it dies, so a
    caller who asks for the same result shape again and again (replicates of a Green-Kubo run) pays the page-locking
    once. At most `max_keep` bytes are kept for reuse.

    Page-locking is not free (tools/pin_cost.py, MI355X host: hipHostMalloc of 24 MB 1.6 ms and as much again to free it,
    against 1.3 ms for the runtime's staged copy into fresh pageable pages and 0.46 ms for the DMA into locked ones), so it
    pays only for memory that comes BACK: at most `max_live` blocks per size are handed out at a time (eight: two calls'
    worth of a three-array result and a couple of inputs kept around) — a caller that keeps every result (replicate lists) gets ordinary arrays after that,
    a caller that drops them (a loop over trajectories, the bench) keeps getting the same locked blocks.
    """

    GRAIN = 1 << 20

    def __init__(self, max_keep=1 << 30, max_live=8):
        self.free = {}
        self.live = {}
        self.kept = 0
        self.max_keep = max_keep
        self.max_live = max_live
        self.lock = threading.Lock()

    def empty(self, shape, dtype=np.float64, device=-1):
        import weakref

        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        cap = max(self.GRAIN, -(-n // self.GRAIN) * self.GRAIN)
        with self.lock:
            lst = self.free.get(cap)
            ptr = lst.pop() if lst else None
            if ptr is not None:
                self.kept -= cap
            elif self.live.get(cap, 0) >= self.max_live:
                return np.empty(shape, dtype=dtype)
            self.live[cap] = self.live.get(cap, 0) + 1
        if ptr is None:
            out = vp()
            rc = load().mdhip_host_alloc_on(int(device), cap, C.byref(out))
            if rc != 0 or not out.value:
                with self.lock:
                    self.live[cap] -= 1
                return np.empty(shape, dtype=dtype)  # (no page-locked memory to be had: an ordinary array works too)
            ptr = out.value
        buf = (C.c_char * cap).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, cap)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def _release(self, ptr, cap):
        with self.lock:
            self.live[cap] = max(0, self.live.get(cap, 0) - 1)
            if self.kept + cap <= self.max_keep:
                self.free.setdefault(cap, []).append(ptr)
                self.kept += cap
                return
        try:
            load().mdhip_host_free(vp(ptr))
        except Exception:
            pass


PINNED = PinnedPool()
PIN_RESULTS_FROM = 4 << 20  # results of at least this many bytes are given page-locked arrays (= MD_STAGE_MAX of ctx.h)


def result_array(shape, dtype=np.float64, device=-1, zero=False):
    """A host array for a library result: page-locked when it is large (see PinnedPool), numpy's own otherwise."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n >= PIN_RESULTS_FROM:
        a = PINNED.empty(shape, dtype, device)
        if zero:
            a[...] = 0
        return a
    return np.zeros(shape, dtype=dtype) if zero else np.empty(shape, dtype=dtype)


class Pending:
    """
    Handle of an asynchronous library call (the *_async entry points of include/mdhip.h): the call's device work is
    queued, `wait()` completes it — and every call of the context issued before it — and returns what the synchronous
    wrapper would have returned, or raises what the completion of THIS call returned (mdhip_ticket_status), whoever
    completed it. The call's arrays (results and converted inputs) belong to the CONTEXT until the call has completed:
    a handle that is dropped un-waited leaves nothing dangling — the library still writes that call's results when a
    later wait / sync / close completes it, into memory the context keeps alive.
    """

    def __init__(self, ctx, result, keep=None, finish=None):
        self.ctx, self._result, self._finish = ctx, result, finish
        ctx._issued += 1
        self._seq = ctx._issued
        self._done = False
        self.ticket = ctx.last_ticket()
        if len(ctx._live) >= 32:  # handles dropped un-waited and completed by synchronous calls since
            ctx._completed_upto(ctx._issued - 1 - ctx.pending())
        ctx._live[self._seq] = (self.ticket, result, keep)

    def wait(self):
        if not self._done:
            ctx = self.ctx
            keep = ctx._issued - self._seq
            rc = 0
            if ctx.pending() > keep:
                rc = ctx.lib.mdhip_wait(ctx.h, int(keep))
            self._done = True
            ctx._completed_upto(self._seq)
            nfb = C.c_int(0)
            st = ctx.lib.mdhip_ticket_status(ctx.h, int(self.ticket), C.byref(nfb))
            if st == -6:  # (cannot happen: the wait above completed it)
                raise MdhipError(st, "call %d still in flight after its wait" % self.ticket)
            if st != 0 and st != -7:
                raise MdhipError(st, (ctx.lib.mdhip_last_error(ctx.h) or b"").decode())
            if st == -7 and rc != 0:  # MDHIP_EUNKNOWN, more than 64 calls ago: all that is known is what the wait returned
                ctx.check(rc)
            if nfb.value:
                ctx._warn_fallback(self.ticket, nfb.value)
                ctx._fallbacks_seen = max(ctx._fallbacks_seen, ctx.fallbacks())
            # rc != 0 with this call fine: the error belongs to an earlier call — its own handle raises it (its status
            # is remembered by ticket); a handle that was dropped cannot, so say it here once
            if rc != 0 and st == 0:
                ctx._orphan_error(rc)
            if self._finish is not None:
                self._result = self._finish(self._result)
                self._finish = None
        return self._result

    def stats(self):
        """(kernel ms, preparation ms, launches, dominant kernel) of this call, once it has completed."""
        return self.ctx.ticket_stats(self.ticket)


class Ready:
    """The same protocol for a result that is already there."""

    def __init__(self, result):
        self._result = result

    def wait(self):
        return self._result


class Context:
    """One mdhip context (device + stream + workspace). Not thread-safe; use one per thread."""

    def __init__(self, device=0):
        self.lib = load()
        h = vp()
        rc = self.lib.mdhip_create(C.byref(h), int(device))
        if rc != 0:
            raise MdhipError(rc, (self.lib.mdhip_last_error(None) or b"").decode())
        self.h = h
        self.device = int(device)
        self._issued = 0       # asynchronous calls issued so far (Pending handles count from here)
        self._stream = None    # None: the context's own stream; else the hipStream_t handle it launches on
        self._live = {}        # seq -> (ticket, results, keep-alive inputs) of the calls not known to have completed
        self._fallbacks_seen = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.mdhip_destroy(self.h)  # completes what is in flight: the arrays in _live are still there
            self.h = None
        self._live = {}

    def _completed_upto(self, seq):
        """Calls complete in issue order: everything up to `seq` has, its arrays go back to their handles' owners."""
        for k in [k for k in self._live if k <= seq]:
            del self._live[k]

    def _orphan_error(self, rc):
        """A wait completed a FAILED call on behalf of a later one. If that call's handle is still alive it raises from
        its own wait(); otherwise nobody would ever hear of it."""
        import warnings

        text = (self.lib.mdhip_last_error(self.h) or b"").decode()
        warnings.warn("an earlier asynchronous mdhip call failed at completion (%d: %s); its handle raises this from "
                      "wait() — if it was dropped, this warning is the only report" % (rc, text), RuntimeWarning,
                      stacklevel=3)

    def _warn_fallback(self, ticket, n):
        import warnings

        warnings.warn("mdhip call %d took %d slow-path repeat(s) (the staged full-lag MSD kernel's grid was not resident "
                      "as a whole — another process on this GPU?): results are the same, the call took "
                      "seconds instead of milliseconds" % (ticket, n), RuntimeWarning, stacklevel=3)

    def note_fallbacks(self):
        """After a synchronous call: warn when it took a slow-path repeat (see mdhip_ticket_status)."""
        n = self.fallbacks()
        if n > self._fallbacks_seen:
            self._warn_fallback(self.last_ticket(), n - self._fallbacks_seen)
        self._fallbacks_seen = n

    def fallbacks(self):
        """Slow-path repeats since the context was created (mdhip_fallbacks)."""
        if not STRICT and not hasattr(self.lib, "mdhip_fallbacks"):
            return 0  # (an older build loaded by an A/B tool)
        return int(self.lib.mdhip_fallbacks(self.h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise MdhipError(rc, (self.lib.mdhip_last_error(self.h) or b"").decode())

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        self.check(self.lib.mdhip_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_option(self, key, value):
        self.check(self.lib.mdhip_set_option(self.h, key.encode(), int(value)))
        self.__dict__.setdefault("_options", {})[key] = int(value)

    def get_option(self, key, default=-1):
        """What set_option last set for `key` through THIS object (`default`: never set — the library's own default)."""
        return self.__dict__.get("_options", {}).get(key, default)

    def set_stream(self, stream_handle):
        """Launch on a caller-owned hipStream_t (e.g. torch.cuda.Stream.cuda_stream); None / 0 restores the own stream.
        Completes what is in flight first."""
        self.check(self.lib.mdhip_set_stream(self.h, vp(stream_handle) if stream_handle else None))
        self._stream = int(stream_handle) if stream_handle else None

    def sync(self):
        """Completes every asynchronous call in flight (their results are then in place) and waits for the stream;
        raises the first error among them."""
        rc = self.lib.mdhip_sync(self.h)
        self._completed_upto(self._issued)
        self.check(rc)

    def wait(self, keep_in_flight=0):
        """Completes the asynchronous calls in flight, oldest first, until at most `keep_in_flight` remain."""
        rc = self.lib.mdhip_wait(self.h, int(keep_in_flight))
        self._completed_upto(self._issued - self.pending())
        self.check(rc)

    def pending(self):
        return int(self.lib.mdhip_pending(self.h))

    def call_stats(self, back=0):
        """(kernel ms, preparation ms, launches, dominant kernel) of the call completed `back` calls ago."""
        ms, aux, n, name = C.c_double(0), C.c_double(0), C.c_int(0), C.c_char_p()
        self.check(self.lib.mdhip_call_stats(self.h, int(back), C.byref(ms), C.byref(aux), C.byref(n), C.byref(name)))
        return float(ms.value), float(aux.value), int(n.value), (name.value or b"").decode()

    def last_ticket(self):
        return int(self.lib.mdhip_last_ticket(self.h))

    def ticket_stats(self, ticket):
        ms, aux, n, name = C.c_double(0), C.c_double(0), C.c_int(0), C.c_char_p()
        self.check(self.lib.mdhip_ticket_stats(self.h, int(ticket), C.byref(ms), C.byref(aux), C.byref(n), C.byref(name)))
        return float(ms.value), float(aux.value), int(n.value), (name.value or b"").decode()

    def last_kernel_ms(self):
        n = C.c_int(0)
        ms = self.lib.mdhip_last_kernel_ms(self.h, C.byref(n))
        return float(ms), int(n.value)

    def last_aux_ms(self):
        return float(self.lib.mdhip_last_aux_ms(self.h))

    def last_kernel_name(self):
        return (self.lib.mdhip_last_kernel_name(self.h) or b"").decode()

    def last_rel_bound(self):
        return float(self.lib.mdhip_last_rel_bound(self.h))


_default = {}


def default_context(device=None):
    """Process-wide context for `device` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        try:  # more ranks than GPUs (several ranks sharing a card): wrap around instead of "device out of range"
            import torch

            n_dev = torch.cuda.device_count()
            if n_dev > 0:
                device %= n_dev
        except Exception:
            pass
    ctx = _default.get(device)
    if ctx is None or ctx.h is None:
        ctx = _default[device] = Context(device)
    return ctx


def bin_edges(bin_size, nbins):
    """Exact edges of bin = trunc(sqrt(rsq)/bin_size) (host-only; no GPU needed)."""
    lib = load()
    out = np.empty(nbins + 1, dtype=np.float64)
    rc = lib.mdhip_bin_edges(float(bin_size), int(nbins), ptr(out))
    if rc != 0:
        raise MdhipError(rc, "mdhip_bin_edges(%r, %r)" % (bin_size, nbins))
    return out
