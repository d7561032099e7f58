"""
Multi-GPU layer: one process per GPU, frames sharded across ranks, `torch.distributed` collectives
("nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed code at all (its only concurrency is a process pool over frames,
/root/reference/mdproptools/dynamical/conductivity.py:190-194). Frames are independent for every
kernel on the path (rdf_cn.py:459; diffusion.py:174; _conductivity.py:7), so:

  RDF / CN     contiguous frame blocks per rank, no data-path exchange; the frame-summed uint64
               histograms ((1+R) x nbins words, tens of KB) are all-reduced once — exact integers, so
               the result does not depend on the number of ranks. With a varying box the per-frame
               integer histograms are all-gathered instead and normalised in frame order.
               On RCCL the sums never leave the GPU before the collective: the kernels' row sums are turned
               into rdf_full | rdf_part | overflow on the device (mdhip_rdf_atomic_dev) and that buffer is
               what all_reduce reads.
  MSD          contiguous frame blocks per rank; the origin frame (24*E bytes) is broadcast from its
               owner, every rank reduces its own frame pairs, the [F_local][G][4] sums are all-gathered.
  full-lag MSD the compute-bound lag x origin average shards by ENTITIES (no exchange in): every rank
               reduces its entity slice over all lags, the per-lag sums are all-reduced (double).
  direct ACF   the n^2/2 estimator shards by LAG RANGE (every rank holds the whole series, 8 MB at
               n = 1e6): ranges of equal work (lag k costs n - k products), slices all-gathered.
  charge flux  as RDF: per-frame [3][T] vectors all-gathered.
  FFT ACF / running integrals: a single transform does not shard — replicas only.

Device residency. Every sharded call below takes its shard as a host array or as a CUDA tensor. With a CUDA tensor the
kernels write their (small) results into device buffers (the *_dev entry points of include/mdhip.h) and the
collective reads those buffers: on RCCL nothing touches the host between the kernel and the collective, and the caller
gets ONE device-to-host copy of the reduced / gathered result (or the device tensor itself, `return_device=True`).
With gloo (tests: several ranks sharing one GPU, or CPU only) the same code stages the collective's operands through
host memory, since gloo gathers only host tensors.

`compute` arguments exist so that the sharding/collective logic can be exercised on CPU with the
oracle as the stand-in (tests/test_dist_gloo.py); the product default is the GPU backend.
"""

import os

import numpy as np


def _dist():
    import torch.distributed as dist

    return dist


def is_distributed():
    try:
        d = _dist()
        return d.is_available() and d.is_initialized()
    except Exception:
        return False


def rank_world():
    if is_distributed():
        d = _dist()
        return d.get_rank(), d.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them)."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world, **kwargs)


def frame_shard(n_frames, rank=None, world=None):
    """Contiguous block [lo, hi) of frames owned by `rank`; sizes differ by at most one."""
    if rank is None or world is None:
        rank, world = rank_world()
    base, extra = divmod(int(n_frames), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _device_for_collectives():
    import torch

    d = _dist()
    if d.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allreduce_u64(arrays):
    """Sum uint64 arrays over all ranks (packed into one int64 message). Exact while totals < 2^63."""
    if not is_distributed():
        return [np.array(a, dtype=np.uint64) for a in arrays]
    import torch

    d = _dist()
    flat = np.concatenate([np.asarray(a, dtype=np.uint64).reshape(-1) for a in arrays]).view(np.int64)
    t = torch.from_numpy(flat.copy()).to(_device_for_collectives())
    d.all_reduce(t, op=d.ReduceOp.SUM)
    out = t.cpu().numpy().view(np.uint64)
    res, pos = [], 0
    for a in arrays:
        n = int(np.prod(np.shape(a)))
        res.append(out[pos:pos + n].reshape(np.shape(a)).copy())
        pos += n
    return res


class _PendingSum:
    """Handle of an all-reduce in flight (allreduce_u64_async): wait() returns the summed arrays."""

    def __init__(self, work, tensor, shapes):
        self._work, self._t, self._shapes = work, tensor, shapes
        self._out = None

    def wait(self):
        if self._out is None:
            if self._work is not None:
                self._work.wait()
            flat = self._t.cpu().numpy().view(np.uint64)
            self._out, pos = [], 0
            for shp in self._shapes:
                n = int(np.prod(shp))
                self._out.append(flat[pos:pos + n].reshape(shp).copy())
                pos += n
        return self._out


def allreduce_u64_async(arrays):
    """allreduce_u64 without waiting: the collective runs while the caller goes on (e.g. with the next batch of
    frames); `.wait()` on the returned handle gives the sums."""
    shapes = [np.shape(a) for a in arrays]
    flat = np.concatenate([np.asarray(a, dtype=np.uint64).reshape(-1) for a in arrays]).view(np.int64)
    import torch

    if not is_distributed():
        return _PendingSum(None, torch.from_numpy(flat.copy()), shapes)
    d = _dist()
    t = torch.from_numpy(flat.copy()).to(_device_for_collectives())
    return _PendingSum(d.all_reduce(t, op=d.ReduceOp.SUM, async_op=True), t, shapes)


def _is_tensor(x):
    return hasattr(x, "is_cuda") and hasattr(x, "data_ptr")


def _coll_tensor(x):
    """(tensor on the device the backend's collectives take, came_as_tensor, home device, uint64 flag)."""
    import torch

    if _is_tensor(x):
        return x.contiguous().to(_device_for_collectives()), True, x.device, False
    x = np.ascontiguousarray(x)
    as_u64 = x.dtype == np.uint64
    return torch.from_numpy(x.view(np.int64) if as_u64 else x).to(_device_for_collectives()), False, None, as_u64


def _coll_result(t, came_as_tensor, home, as_u64):
    if came_as_tensor:
        return t.to(home)
    a = t.cpu().numpy()
    return a.view(np.uint64) if as_u64 else a


def _gather_blocks(local, counts):
    """Per-rank blocks of rows (counts[r] rows from rank r) concatenated in rank order on every rank. `local` is a host
    array (float64 / int64 / uint64) or a tensor; the result has the same kind (a CUDA tensor stays on the GPU: on
    RCCL the gather reads and writes device memory only)."""
    import torch

    d = _dist()
    _, world = rank_world()
    t, was_t, home, as_u64 = _coll_tensor(local)
    row_shape = tuple(t.shape[1:])
    longest = max(max(counts), 1)
    if all(c == longest for c in counts):
        send = t
    else:
        send = torch.zeros((longest,) + row_shape, dtype=t.dtype, device=t.device)
        send[: t.shape[0]] = t
    recv = [torch.empty_like(send) for _ in range(world)]
    d.all_gather(recv, send)
    out = torch.cat([recv[r][: counts[r]] for r in range(world)]) if world > 1 else recv[0][: counts[0]]
    return _coll_result(out, was_t, home, as_u64)


def allgather_rows(local, n_total_rows):
    """
    Concatenate per-rank blocks of rows (frame shards, in rank order) into the full array on every rank.
    `local` [rows_local, ...] — host array or tensor (see _gather_blocks); blocks may differ in length by one
    (see frame_shard).
    """
    if not is_distributed():
        return local if _is_tensor(local) else np.ascontiguousarray(local)
    _, world = rank_world()
    counts = [frame_shard(n_total_rows, r, world)[1] - frame_shard(n_total_rows, r, world)[0] for r in range(world)]
    return _gather_blocks(local, counts)


def allgather_counts(n_local):
    """The number of rows every rank holds, in rank order (one small all-gather)."""
    if not is_distributed():
        return [int(n_local)]
    import torch

    d = _dist()
    _, world = rank_world()
    cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=_device_for_collectives())
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    d.all_gather(counts, cnt)
    return [int(c.item()) for c in counts]


def allgather_var(local, counts=None):
    """
    Concatenate per-rank blocks of rows of ANY length (rank order) into the full array on every rank: the block lengths
    are gathered first (unless the caller has them), blocks are padded to the longest one. Host array or tensor.
    """
    if not is_distributed():
        return local if _is_tensor(local) else np.ascontiguousarray(local)
    if counts is None:
        counts = allgather_counts(local.shape[0])
    return _gather_blocks(local, counts)


def require_all_nonempty(n_local, what="frame"):
    """Collective check that every rank holds at least one item: the counts are all-gathered FIRST, so that all ranks
    raise together — a rank that raised alone would leave the others waiting in the next collective until the
    process-group timeout."""
    if not is_distributed():
        return
    import torch

    d = _dist()
    _, world = rank_world()
    dev = _device_for_collectives()
    cnt = torch.tensor([int(n_local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    d.all_gather(counts, cnt)
    empty = [r for r, c in enumerate(counts) if int(c.item()) == 0]
    if empty:
        raise ValueError("rank(s) %s hold no %s: use at most as many ranks as there are %ss" % (empty, what, what))


def raise_together(err, what="its share of the work"):
    """
    Collective agreement on rank-local failures: every rank passes the exception its local work raised (None: it went
    through). If any rank failed, ALL ranks raise here — the failing ones their own exception, the others a RuntimeError
    that names them — instead of one rank raising alone and the rest waiting in the next collective until the process
    group times out. A no-op for a single process (the exception is re-raised as it is).
    """
    if not is_distributed():
        if err is not None:
            raise err
        return
    flags = allgather_counts(0 if err is None else 1)
    if not any(flags):
        return
    if err is not None:
        raise err
    failed = [r for r, f in enumerate(flags) if f]
    raise RuntimeError("rank(s) %s failed on %s; this rank stops with them" % (failed, what))


def shard_items(items):
    """This rank's contiguous block of a list (e.g. the dump files of a trajectory, in frame order)."""
    rank, world = rank_world()
    lo, hi = frame_shard(len(items), rank, world)
    return items[lo:hi]


def my_files(file_pattern):
    """This rank's contiguous share of the files matching `file_pattern` (numeric order), or None when the
    trajectory should not be split by files (single process, or fewer files than ranks)."""
    if not is_distributed():
        return None
    from . import io as mio

    matches = mio._sorted_matches(str(file_pattern))
    if len(matches) < rank_world()[1]:
        return None
    return shard_items(matches)


def is_writer():
    """Rank 0 (or a single process): the one that writes result files."""
    return rank_world()[0] == 0


def broadcast_array(arr, src, shape, dtype=np.float64, like=None):
    """Broadcast a float64 array from rank `src` (others pass arr=None). `like`: a CUDA tensor of this rank — the
    result is then a tensor on its device, and on RCCL the broadcast goes device to device."""
    import torch

    if like is not None and _is_tensor(like):
        if not is_distributed():
            return arr
        d = _dist()
        rank, _ = rank_world()
        dev = _device_for_collectives()
        t = arr.contiguous().to(dev) if rank == src else torch.empty(tuple(shape), dtype=like.dtype, device=dev)
        d.broadcast(t, src=src)
        return t.to(like.device)
    if not is_distributed():
        return np.asarray(arr, dtype=dtype)
    d = _dist()
    rank, _ = rank_world()
    buf = np.ascontiguousarray(arr, dtype=dtype) if rank == src else np.zeros(shape, dtype=dtype)
    t = torch.from_numpy(buf).to(_device_for_collectives())
    d.broadcast(t, src=src)
    return t.cpu().numpy()


def allreduce_tensor(t):
    """Sum a tensor over all ranks; returns a tensor on t's device (on RCCL: t itself, reduced in place)."""
    if not is_distributed():
        return t
    d = _dist()
    c = t.contiguous().to(_device_for_collectives())
    d.all_reduce(c, op=d.ReduceOp.SUM)
    return c.to(t.device)


# ------------------------------------------------------------------------------------------------
# sharded hot-path calls
# ------------------------------------------------------------------------------------------------


def _on_rccl(x):
    """True when the collective can read device memory directly: RCCL backend and a CUDA tensor input."""
    return is_distributed() and _dist().get_backend() == "nccl" and bool(getattr(x, "is_cuda", False))


class _PendingRdf:
    """
    One sharded `_rdf_loop` step in flight, in two stages:
      reduce()  completes the LOCAL library call (mdhip_wait: its kernels have run, its flags are checked) and issues
                the all-reduce of the uint64 words, asynchronously — on RCCL straight from the device buffer the kernels
                wrote;
      wait()    (reduce() first if it has not happened) waits for the collective and returns
                [rdf_full [nbins], rdf_part [R, nbins], [overflow]] of the whole trajectory.
    A pipeline calls reduce() of step k right after it has ISSUED step k + 1 and wait() of step k one step later: the
    GPU always has the next step's kernels queued while the host waits, and the collective of step k — which gets no
    compute unit before the persistent pair kernel of step k + 1 lets go of some — has a whole step to finish.
    """

    def __init__(self, local, n_rel, nbins, on_device):
        self._local, self._R, self._nb, self._dev = local, n_rel, nbins, on_device
        self._work, self._t, self._out = None, None, None
        self.stats = None

    def reduce(self):
        if self._t is not None or self._out is not None:
            return self
        import torch

        res = self._local.wait()
        self.stats = self._local.stats() if hasattr(self._local, "stats") else None
        if self._dev:
            self._t = res  # int64 CUDA tensor: full | part | overflow
        else:
            full, part, ov = res
            flat = np.concatenate([full.reshape(-1), part.reshape(-1), np.array([ov], dtype=np.uint64)]).view(np.int64)
            self._t = torch.from_numpy(flat)
        if is_distributed():
            d = _dist()
            if not self._dev:
                self._t = self._t.to(_device_for_collectives())
            self._work = d.all_reduce(self._t, op=d.ReduceOp.SUM, async_op=True)
        return self

    def wait(self):
        if self._out is None:
            self.reduce()
            if self._work is not None:
                self._work.wait()
            flat = self._t.cpu().numpy().view(np.uint64)
            nb, R = self._nb, self._R
            self._out = [flat[:nb].copy(), flat[nb:(1 + R) * nb].reshape(R, nb).copy(), flat[(1 + R) * nb:].copy()]
            self._t = self._local = None
        return self._out


def rdf_sharded_async(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins, compute=None, ctx=None):
    """
    `_rdf_loop` over a frame-sharded trajectory, constant box: every rank passes ITS frames [F_local,3,N]; returns a
    handle (see _PendingRdf: reduce(), wait()) whose wait() gives [rdf_full [nbins], rdf_part [R,nbins], [overflow]] of
    the whole trajectory on every rank — one all-reduce of (1+R)*nbins+1 uint64 words. The local sweep is issued through
    the library's *_async entry points: this call returns with the kernels queued, so that a pipeline can issue the next
    batch of frames before it waits for this one. On RCCL with device-resident frames the words go from the kernels to
    the collective without touching the host.
    """
    R = len(np.asarray(relation_matrix).reshape(-1, 2))
    if compute is None and _on_rccl(xyz_local):
        import torch

        from . import backend

        out = torch.empty((1 + R) * int(nbins) + 1, dtype=torch.int64, device=xyz_local.device)
        local = backend.rdf_loop_dev(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins, out, ctx=ctx,
                                     async_=True)
        return _PendingRdf(local, R, int(nbins), True)
    if compute is None:
        from . import backend

        local = backend.rdf_loop(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins, per_frame=False,
                                 ctx=ctx, async_=True)
        return _PendingRdf(local, R, int(nbins), False)
    from ._lib import Ready

    return _PendingRdf(Ready(compute(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins)), R, int(nbins), False)


def rdf_sharded(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins, compute=None, ctx=None):
    """rdf_sharded_async, waited for: (rdf_full [nbins], rdf_part [R,nbins], overflow) on every rank."""
    full, part, ovv = rdf_sharded_async(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins,
                                        compute=compute, ctx=ctx).wait()
    return full, part, int(ovv[0])


def rdf_sharded_per_frame(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins, n_frames_total,
                          compute=None, ctx=None):
    """
    Varying box (NPT): per-frame integer histograms are all-gathered in frame order so that the host can
    normalise every frame with its own volume exactly as the reference does (rdf_cn.py:502-521).
    Returns (rdf_full [F,nbins], rdf_part [F,R,nbins], overflow).

    ONE collective: a rank's rows travel as [frames_local, (1 + R) * nbins + 1] words — totals | partials | the rank's
    overflow count in the last column of its first row (zeros below) — so the gathered last column sums to the
    job's overflow. With device-resident frames the packed rows are gathered as a device tensor (RCCL reads and writes
    device memory; gloo takes the same tensor through the host), and `ctx` is the context the sweep runs in.
    """
    if compute is None:
        from . import backend

        def compute(x, t, b, rel, rc, dd, nb):
            return backend.rdf_loop(x, t, b, rel, rc, dd, nb, per_frame=True, ctx=ctx)

    full, part, ov = compute(xyz_local, types, box_local, relation_matrix, r_cut, ddr, nbins)
    full = np.asarray(full, dtype=np.uint64)
    part = np.asarray(part, dtype=np.uint64)
    fl, R = full.shape[0], part.shape[1]
    packed = np.zeros((fl, (1 + R) * int(nbins) + 1), dtype=np.uint64)
    packed[:, :nbins] = full
    packed[:, nbins:-1] = part.reshape(fl, R * int(nbins))
    if fl:
        packed[0, -1] = np.uint64(ov)
    if _is_tensor(xyz_local) and xyz_local.is_cuda and is_distributed():
        import torch

        dev = torch.from_numpy(packed.view(np.int64)).to(xyz_local.device, non_blocking=True)
        rows = allgather_rows(dev, n_frames_total).cpu().numpy().view(np.uint64)
    else:
        rows = allgather_rows(packed, n_frames_total)
    if not is_distributed() and fl == 0:
        return full, part, int(ov)
    return (np.ascontiguousarray(rows[:, :nbins]),
            np.ascontiguousarray(rows[:, nbins:-1]).reshape(rows.shape[0], R, int(nbins)),
            int(rows[:, -1].sum()))


def cn_sharded(xyz_local, types, box_local, relation_matrix, r_cut_list, compute=None, ctx=None):
    """`_cn_loop` counts summed over all frames of all ranks (one all-reduce of R words). With device-resident frames
    the counts go from the library's device buffer to the collective (mdhip_cn_atomic_dev)."""
    if compute is None and _is_tensor(xyz_local) and xyz_local.is_cuda:
        import torch

        from . import backend

        R = len(np.asarray(relation_matrix).reshape(-1, 2))
        out = torch.empty(R, dtype=torch.int64, device=xyz_local.device)
        backend.cn_loop(xyz_local, types, box_local, relation_matrix, r_cut_list, per_frame=False, ctx=ctx, out=out)
        return allreduce_tensor(out).cpu().numpy().view(np.uint64)
    if compute is None:
        from . import backend

        def compute(x, t, b, rel, cuts):
            return backend.cn_loop(x, t, b, rel, cuts, per_frame=False, ctx=ctx)

    (cn,) = allreduce_u64([compute(xyz_local, types, box_local, relation_matrix, r_cut_list)])
    return cn


def frame_blocks(n_frames_total, counts=None, world=None):
    """[(lo, hi)] per rank of a trajectory dealt to the ranks in contiguous blocks: frame_shard's even split, or the
    given per-rank frame counts (ranks that parsed whole files hold what their files held)."""
    if world is None:
        world = rank_world()[1]
    if counts is None:
        return [frame_shard(n_frames_total, r, world) for r in range(world)]
    if len(counts) != world or sum(counts) != int(n_frames_total):
        raise ValueError("counts must give one frame count per rank, adding up to n_frames_total")
    edges = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    return [(int(edges[r]), int(edges[r + 1])) for r in range(world)]


def frame_owner(n_frames_total, frame, world=None, counts=None):
    """The rank whose contiguous frame block holds `frame`."""
    blocks = frame_blocks(n_frames_total, counts, world)
    return next(r for r, (lo, hi) in enumerate(blocks) if lo <= frame < hi)


def msd_single_origin_sharded(r_local, n_frames_total, group_off, scale=1.0, origin_frame=0, compute=None, ctx=None,
                              return_device=False, counts=None, cols=None):
    """
    Single-origin MSD sums (diffusion.py:212-218) over a frame-sharded trajectory r_local [F_local,3,E]:
    the origin frame is broadcast from its owner (24 E bytes), every rank reduces its own (origin, t) pairs, the
    [F_local,G,4] sums are all-gathered into [F,G,4] (frame order).
    r_local as a CUDA tensor: the broadcast, the reduction (mdhip_msd_origin: the origin frame is a separate device
    buffer, the shard is not copied) and the gather all work on device memory; the result comes back as a host array,
    or as the device tensor with return_device. `cols`: receives the per-entity columns of THIS rank's frames
    (backend.msd_origin); `counts`: per-rank frame counts when the split is not frame_shard's.
    """
    rank, world = rank_world()
    blocks = frame_blocks(n_frames_total, counts, world)
    lo, hi = blocks[rank]
    owner = frame_owner(n_frames_total, origin_frame, world, counts)
    cnts = [b - a for a, b in blocks]
    E = r_local.shape[2]
    G = len(group_off) - 1
    if int(r_local.shape[0]) != hi - lo:
        raise ValueError("this rank holds %d frames, its block is [%d, %d)" % (int(r_local.shape[0]), lo, hi))
    if compute is None and _is_tensor(r_local) and r_local.is_cuda:
        import torch

        from . import backend

        r0 = broadcast_array(r_local[origin_frame - lo] if rank == owner else None, owner, (3, E), like=r_local)
        sums = torch.empty((hi - lo, G, 4), dtype=torch.float64, device=r_local.device)
        if hi > lo:
            backend.msd_origin(r_local, r0, group_off, scale=scale, out=sums, cols=cols, ctx=ctx)
        full = _gather_blocks(sums, cnts) if is_distributed() else sums
        return full if return_device else full.cpu().numpy()
    r_np = np.asarray(r_local.cpu() if _is_tensor(r_local) else r_local)
    r0 = broadcast_array(r_np[origin_frame - lo] if rank == owner else None, owner, (3, E))
    if compute is None:
        from . import backend

        sums = backend.msd_origin(r_np, r0, group_off, scale=scale, cols=cols, ctx=ctx) if hi > lo \
            else np.zeros((0, G, 4))
    else:
        stacked = np.concatenate([r0[None], r_np]) if hi > lo else r0[None]
        pairs = np.column_stack([np.zeros(hi - lo, dtype=np.int32), 1 + np.arange(hi - lo, dtype=np.int32)])
        sums = compute(stacked, pairs, group_off, scale) if hi > lo else np.zeros((0, G, 4))
    return _gather_blocks(sums, cnts) if is_distributed() else sums


def msd_windows_sharded(r_local, n_frames_total, tao, scale=1.0, compute=None, ctx=None, return_device=False,
                        counts=None):
    """
    Fixed-lag per-entity window sums (diffusion.py:225-237; mdhip_msd_windows) over a frame-sharded trajectory:
    frames kept = 0, tao, 2 tao, ... of the GLOBAL order; a window (k-1, k) belongs to the rank that holds kept frame
    k. Every rank therefore needs ONE frame from before its block — the last kept frame of the ranks below it — and
    gets it from an all-gather of "my last kept frame" (24 E bytes per rank). The per-entity sums [E,4] of the ranks
    are added by an all-reduce (double: the windows are summed rank by rank instead of in one sequence, so the last
    bits may differ from a single-GPU run; inside the rtol 1e-10 bar by orders of magnitude).
    """
    rank, world = rank_world()
    blocks = frame_blocks(n_frames_total, counts, world)
    lo, hi = blocks[rank]
    E = int(r_local.shape[2])
    tao = int(tao)

    def kept_of(a, b):  # global indices of the kept frames inside [a, b)
        return np.arange(-(-a // tao) * tao, b, tao)

    kept_local = kept_of(lo, hi) - lo
    dev = _is_tensor(r_local) and r_local.is_cuda and compute is None
    if compute is None:
        from . import backend

        def compute(r, sc):
            return backend.msd_windows(r, 1, scale=sc, ctx=ctx)

    if dev:
        import torch

        zero = torch.zeros((1, 3, E), dtype=torch.float64, device=r_local.device)
        mine = r_local[int(kept_local[-1])][None] if len(kept_local) else zero
        lasts = allgather_var(mine, counts=[1] * world) if world > 1 else mine
    else:
        r_np = np.asarray(r_local.cpu() if _is_tensor(r_local) else r_local)
        mine = r_np[int(kept_local[-1])][None] if len(kept_local) else np.zeros((1, 3, E))
        lasts = allgather_var(mine, counts=[1] * world) if world > 1 else mine
        kept = r_np[kept_local]
    # the halo: the last kept frame of the nearest rank below that has one
    halo = None
    for q in range(rank - 1, -1, -1):
        if len(kept_of(*blocks[q])):
            halo = lasts[q]
            break
    if dev:
        from . import backend

        sums = torch.zeros((E, 4), dtype=torch.float64, device=r_local.device)
        if len(kept_local):
            # my kept frames lie tao apart from the first one on: the kernel strides over them where they are (no
            # gathered copy of every tao-th frame) ...
            k0 = int(kept_local[0])
            backend.msd_windows(r_local[k0:], tao, scale=scale, ctx=ctx, out=sums)
            if halo is not None:
                # ... and the one window that reaches back to the rank below is three planes of arithmetic
                # (the kernel's operations: scale, subtract, square, (dx2 + dy2) + dz2)
                d2 = (r_local[k0] * scale - halo * scale) ** 2
                sums += torch.stack([d2[0], d2[1], d2[2], (d2[0] + d2[1]) + d2[2]], dim=1)
        sums = allreduce_tensor(sums)
        return sums if return_device else sums.cpu().numpy()
    block = kept if halo is None else np.concatenate([halo[None], kept])
    sums = np.asarray(compute(np.ascontiguousarray(block), scale)) if block.shape[0] > 1 else np.zeros((E, 4))
    if is_distributed():
        import torch

        sums = allreduce_tensor(torch.from_numpy(np.ascontiguousarray(sums))).numpy()
    return sums


def charge_flux_sharded(vel_local, n_frames_total, atom_mass, atom_q, seg_off, seg_type, n_types, vel_conv,
                        charge_conv, compute=None, ctx=None):
    """Per-frame charge flux of a frame-sharded trajectory, gathered to j [3, T, F] on every rank (the reference's
    only frame-parallel gather, conductivity.py:190-194). Device-resident velocities: the [3,T,F_local] block stays
    on the GPU (mdhip_charge_flux_dev) until it has been gathered."""
    if compute is None and _is_tensor(vel_local) and vel_local.is_cuda:
        import torch

        from . import backend

        j_local = torch.empty((3, int(n_types), int(vel_local.shape[0])), dtype=torch.float64, device=vel_local.device)
        backend.charge_flux(vel_local, atom_mass, atom_q, seg_off, seg_type, n_types, vel_conv, charge_conv, ctx=ctx,
                            out=j_local)
        rows = j_local.permute(2, 0, 1).contiguous()  # [F_local, 3, T]
        return allgather_rows(rows, n_frames_total).permute(1, 2, 0).contiguous().cpu().numpy()
    if compute is None:
        from . import backend

        def compute(*a):
            return backend.charge_flux(*a, ctx=ctx)
    j_local = compute(vel_local, atom_mass, atom_q, seg_off, seg_type, n_types, vel_conv, charge_conv)
    rows = np.ascontiguousarray(np.moveaxis(j_local, 2, 0))  # [F_local, 3, T]
    return np.moveaxis(allgather_rows(rows, n_frames_total), 0, 2)


_LAG_CONST = {}  # lag_msd_sharded: device-resident weights per (device, shapes)


def lag_msd_sharded(r_local, entity_range, max_lag, group_off, scale=1.0, compute=None, ctx=None):
    """
    Full lag x origin MSD (mdhip_lag_msd; the compute-bound superset of diffusion.py:225-238) with the ENTITIES
    sharded: this rank holds r_local [F,3,E_local], the entities entity_range = (e_lo, e_hi) of the global order;
    group_off [G+1] are the global contiguous groups. Every rank reduces its slice over all lags (no exchange in),
    the per-(lag, group) SUMS are all-reduced (double: the order of the ranks' partial sums differs from the single-
    GPU order in the last bits, inside the rtol 1e-10 bar) and divided by the global counts.
    Returns msd [max_lag+1, G, 4] on every rank. entity_shard(E) gives the contiguous split.
    r_local as a CUDA tensor: means -> sums -> all-reduce -> means all on the device (mdhip_lag_msd_dev), one copy to
    the host at the end.
    """
    dev = compute is None and _is_tensor(r_local) and r_local.is_cuda
    if compute is None:
        from . import backend

        def compute(r, ml, goff, sc):
            return backend.lag_msd(r, ml, goff, scale=sc, ctx=ctx)

    e_lo, e_hi = int(entity_range[0]), int(entity_range[1])
    goff = np.asarray(group_off, dtype=np.int64)
    G = len(goff) - 1
    F = int(r_local.shape[0])
    n_lags = int(max_lag) + 1
    # this rank's part of every group, as local offsets; groups it holds nothing of are left out of the call
    lo = np.clip(goff[:-1], e_lo, e_hi) - e_lo
    hi = np.clip(goff[1:], e_lo, e_hi) - e_lo
    held = [g for g in range(G) if hi[g] > lo[g]]
    counts = (F - np.arange(n_lags)).astype(np.float64)[:, None] * (goff[1:] - goff[:-1]).astype(np.float64)[None, :]
    origins = (F - np.arange(n_lags)).astype(np.float64)[:, None]
    if dev:
        import torch

        from . import backend

        sums = torch.zeros((n_lags, G, 4), dtype=torch.float64, device=r_local.device)
        if held and F > 0:
            loc_off = np.array([lo[held[0]]] + [hi[g] for g in held], dtype=np.int64)
            x = r_local[:, :, int(loc_off[0]):int(loc_off[-1])].contiguous()
            means = torch.empty((n_lags, len(held), 4), dtype=torch.float64, device=r_local.device)
            backend.lag_msd(x, int(max_lag), loc_off - loc_off[0], scale=scale, ctx=ctx, out=means)
            # (the weights and the index of the held groups depend on the shapes only: kept on the device between calls)
            key = (str(r_local.device), F, n_lags, tuple(int(v) for v in (hi[held] - lo[held])), tuple(held))
            cached = _LAG_CONST.get(key)
            if cached is None:
                w = torch.from_numpy(origins * (hi[held] - lo[held]).astype(np.float64)[None, :]).to(r_local.device)
                cached = (w[:, :, None].contiguous(), torch.as_tensor(held, device=r_local.device))
                if len(_LAG_CONST) > 8:
                    _LAG_CONST.clear()
                _LAG_CONST[key] = cached
            if len(held) == G:
                sums = means * cached[0]
            else:
                sums[:, cached[1], :] = means * cached[0]
        sums = allreduce_tensor(sums).cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(counts[:, :, None] > 0, sums / np.maximum(counts, 1.0)[:, :, None], 0.0)
    sums = np.zeros((n_lags, G, 4))
    if held and F > 0:
        # the held groups are contiguous in the slice (groups are contiguous globally)
        loc_off = np.array([lo[held[0]]] + [hi[g] for g in held], dtype=np.int64)
        x = r_local[:, :, int(loc_off[0]):int(loc_off[-1])]
        if hasattr(x, "contiguous"):
            x = x.contiguous()
        means = np.asarray(compute(x, int(max_lag), loc_off - loc_off[0], scale))
        for k, g in enumerate(held):
            sums[:, g, :] = means[:, k, :] * (origins * float(hi[g] - lo[g]))
    if is_distributed():
        import torch

        sums = allreduce_tensor(torch.from_numpy(sums)).numpy()
    with np.errstate(invalid="ignore", divide="ignore"):
        out = np.where(counts[:, :, None] > 0, sums / counts[:, :, None], 0.0)
    return out


import weakref

# msd_step_sharded: context -> {(device index, post): torch stream}. Keyed on the Context OBJECT (weakly): an id() can be
# reused by a new context once the old one has been collected, and would hand it a stream it never bound.
_STEP_STREAM = weakref.WeakKeyDictionary()


def _step_stream(device, ctx, post=False):
    """The stream a fused step is ISSUED on: the context's kernels and torch's own work before them (the pre-exchange,
    the zero fill of the result buffer) are queued on ONE stream, so that their order needs no host wait in between.
    `post`: the second stream of the pair, for what follows the kernels of a step that has been waited for.
    NOTE: the context STAYS bound to that stream after the step (steps are pipelined: step k + 1 is issued before step k
    is waited for, so there is no point at which a step could hand the old stream back); `release_step_stream(ctx)`
    restores the context's own stream when the caller is done with sharded steps."""
    import torch

    per_ctx = _STEP_STREAM.setdefault(ctx, {})
    key = (device.index, bool(post))
    s = per_ctx.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)
        per_ctx[key] = s
    if not post and getattr(ctx, "_stream", None) != s.cuda_stream:
        ctx.set_stream(s.cuda_stream)
    return s


def release_step_stream(ctx):
    """Undo what msd_step_sharded_async did to `ctx`: complete what is in flight and launch on the context's own stream
    again."""
    if _STEP_STREAM.pop(ctx, None) is not None:
        ctx.set_stream(None)


_LAGW = {}


def _lag_weights(origins, g_hi, g_lo, held, dev):
    """[n_lags, len(held)] origins x entities of this rank's part of every held group, on the device (kept from step to
    step: a fresh tensor would be a synchronous copy from pageable memory in every step)."""
    key = (dev.index, len(origins), tuple(int(g_hi[g] - g_lo[g]) for g in held))
    w = _LAGW.get(key)
    if w is None:
        if len(_LAGW) > 16:
            _LAGW.clear()
        w = _LAGW[key] = torch_from(origins * (g_hi[held] - g_lo[held]).astype(np.float64)[None, :], dev)
    return w


def torch_from(a, dev):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _pinned_like(t):
    """A page-locked host tensor of t's size (torch's caching host allocator: recycled once the step's results — numpy
    views that keep the tensor alive — are dropped) and its numpy view. Page-locked as TORCH knows it: a non_blocking
    copy into memory torch does not know to be page-locked is a synchronous one."""
    import torch

    ht = torch.empty(int(t.numel()), dtype=t.dtype, pin_memory=True)
    return ht.numpy(), ht


_POST_GROUP = {}


def _post_group():
    """A second communicator for the all-reduce that ENDS a fused step. Collectives of one communicator run in the order
    they were issued; a step's all-reduce is issued (queued behind the step's kernels) before the next step's all-gather,
    so on ONE communicator the next step's pre-exchange — and with it the next step's kernels — would wait for this
    step's kernels, post-processing and collective: no pipelining (measured on the shard of 8: 1.25 ms per step against
    0.95, a one-rank RCCL group). Created collectively by the first fused step of every rank. OPT-IN
    (MDHIP_STEP_POST_GROUP=1): see the call site."""
    d = _dist()
    key = id(d.group.WORLD)
    g = _POST_GROUP.get(key)
    if g is None:
        g = _POST_GROUP[key] = d.new_group(backend=d.get_backend())
    return g


def _allreduce_inplace(t, group=None):
    """Sum `t` (a CUDA tensor) over the ranks, in place: RCCL on the tensor itself, gloo through a host copy."""
    d = _dist()
    if d.get_backend() == "nccl":
        d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
        return
    c = t.cpu()
    d.all_reduce(c, op=d.ReduceOp.SUM, group=group)
    t.copy_(c)


def _allgather_equal(t):
    """[world, ...] of every rank's `t` (same shape everywhere; CUDA tensor in, CUDA tensor out)."""
    import torch

    d = _dist()
    _, world = rank_world()
    if d.get_backend() == "nccl":
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        d.all_gather_into_tensor(out, t.contiguous())
        return out
    c = t.cpu()
    recv = [torch.empty_like(c) for _ in range(world)]
    d.all_gather(recv, c)
    return torch.stack(recv).to(t.device)


def msd_step_sharded_async(r_f, r_e, n_frames_total, entity_range, group_off, tao, scale=1.0, lag_scale=1.0, origin_frame=0,
                     ctx=None, counts=None, compute=None):
    """
    The three MSD reductions of one trajectory as ONE step (BASELINE configs[3]; what `bench.py --workload c4` times):
      single origin  (diffusion.py:212-218)  frames dealt to the ranks, r_f [F_local,3,E]     -> sums  [F,G,4]
      fixed lag tao  (diffusion.py:225-237)  the same frames                                   -> sums  [E,4]
      full lag x origin average (superset)   ENTITIES dealt to the ranks, r_e [F,3,E_local]   -> means [F,G,4]
    Returns a handle: wait() -> (single, windows, lag) as host arrays, the same on every rank, plus the per-call kernel
    times. The library calls are ISSUED when this function returns (their kernels queued behind the all-gather); what
    has to wait for them — the all-reduce and the copy to the host — is queued behind them and waited for in
    wait(). A pipeline issues step k + 1 before it waits for step k, so that the GPU never idles while the host works.

    What crosses between the ranks is coalesced into ONE collective before the kernels and ONE after them:
      before  an all-gather of two frames per rank — the origin frame (zeros from every rank but its owner) and the rank's
              last kept frame (the one-frame halo of the fixed-lag windows);
      after   an all-reduce of ONE buffer: the single-origin rows at their global offsets in a zero array (a row is
              non-zero on exactly one rank: the sum IS the gather, exactly), the window sums, the lag sums.
    The three library calls are issued through their *_async entry points on the stream the collectives are ordered
    with; the host waits ONCE per step (round 5: the spectral lag path finishes on the device, every length since round 6;
    the round-4 order — two waits — is MDHIP_STEP_ONE_WAIT=0), plus, with more than one rank, one summed word on which the
    ranks agree about errors that only showed at completion. With one process it is three queued calls and one wait.
    `compute` (tests: the exchange logic on CPU over gloo, with the oracle as the stand-in): a dict of
    "origin"(r [f,3,E], r0 [3,E], goff, scale) -> [f,G,4], "windows"(r [f,3,E], tao, scale) -> [E,4] (windows between the
    frames 0, tao, 2 tao ... of r) and "lag"(x [F,3,e], max_lag, loc_off, scale) -> means [F,g,4] on numpy arrays.
    """
    import contextlib

    import torch

    rank, world = rank_world()
    on_gpu = compute is None
    if on_gpu:
        from . import backend
        from ._lib import default_context

        ctx = ctx or default_context()
    else:
        r_f = torch.as_tensor(np.ascontiguousarray(r_f))
        r_e = torch.as_tensor(np.ascontiguousarray(r_e))
    F = int(n_frames_total)
    blocks = frame_blocks(F, counts, world)
    lo, hi = blocks[rank]
    E = int(r_f.shape[2])
    goff = np.asarray(group_off, dtype=np.int64)
    G = len(goff) - 1
    tao = int(tao)
    max_lag = F - 1
    n_lags = F
    e_lo, e_hi = int(entity_range[0]), int(entity_range[1])
    if int(r_f.shape[0]) != hi - lo or int(r_e.shape[0]) != F or int(r_e.shape[2]) != e_hi - e_lo:
        raise ValueError("shard shapes do not match the frame block / entity range of this rank")
    # this rank's part of every group of the entity shard (as lag_msd_sharded)
    g_lo = np.clip(goff[:-1], e_lo, e_hi) - e_lo
    g_hi = np.clip(goff[1:], e_lo, e_hi) - e_lo
    held = [g for g in range(G) if g_hi[g] > g_lo[g]]
    lag_counts = (F - np.arange(n_lags)).astype(np.float64)[:, None] * (goff[1:] - goff[:-1]).astype(np.float64)[None, :]
    origins = (F - np.arange(n_lags)).astype(np.float64)[:, None]

    def kept_of(a, b):  # global indices of the kept frames inside [a, b)
        return np.arange(-(-a // tao) * tao, b, tao)

    kept_local = kept_of(lo, hi) - lo
    stats = {}
    if not is_distributed() and on_gpu:
        h1 = backend.msd_origin(r_f, r_f[origin_frame], goff, scale=scale, ctx=ctx, async_=True)
        h2 = backend.msd_windows(r_f, tao, scale=scale, ctx=ctx, async_=True)
        h3 = backend.lag_msd(r_e, max_lag, goff, scale=lag_scale, ctx=ctx, async_=True)

        def finish_local():
            out = (h1.wait(), h2.wait(), h3.wait())  # (h1 first: the waits complete the calls in issue order)
            for key, h in (("single", h1), ("fixed", h2), ("lag", h3)):
                stats[key] = h.stats()
            return out + (stats,)

        return _Deferred(finish_local)
    dev = r_f.device
    owner = frame_owner(F, origin_frame, world, counts)
    nS, nW, nL = F * G * 4, E * 4, n_lags * G * 4
    with (torch.cuda.stream(_step_stream(dev, ctx)) if on_gpu else contextlib.nullcontext()):
        zero = None if (rank == owner and len(kept_local)) else torch.zeros((3, E), dtype=torch.float64, device=dev)
        mine = torch.stack([r_f[origin_frame - lo] if rank == owner else zero,
                            r_f[int(kept_local[-1])] if len(kept_local) else zero])
        allf = _allgather_equal(mine) if world > 1 else mine[None]  # [world, 2, 3, E]
        r0 = allf[owner, 0].contiguous()
        halo = None
        for q in range(rank - 1, -1, -1):
            if len(kept_of(*blocks[q])):
                halo = allf[q, 1]
                break
        # (+ 2: a failure flag — a rank whose local calls raised still takes part in the all-reduce and every rank raises
        # after it, instead of the others waiting in the collective until it times out — and the STATUS of the lag call
        # (mdhip_lag_msd_status_dev: its error bound, +inf when its result will be rewritten at completion), which
        # every rank reads from the reduced buffer: whether the lag sums must be redone is agreed on without a host
        # wait between the kernels and the collective)
        # (only what no kernel of this rank writes is zeroed: the other ranks' single-origin rows, the lag sums of groups
        # held elsewhere, the two flag words — the window sums, most of the buffer, are written whole by their kernel; a
        # fill of all 1.9 MB cost the shard-of-8 step 37 us of its ~1 ms)
        on_dev_fill = on_gpu
        res = (torch.empty if on_dev_fill else torch.zeros)(nS + nW + nL + 2, dtype=torch.float64, device=dev)
        single = res[:nS].view(F, G, 4)
        win = res[nS:nS + nW].view(E, 4)
        lagsum = res[nS + nW:nS + nW + nL].view(n_lags, G, 4)
        if on_dev_fill:
            if lo > 0:
                single[:lo].zero_()
            if hi < F:
                single[hi:].zero_()
            if not (len(kept_local) and on_gpu):
                win.zero_()
            res[nS + nW:].zero_()
        hs = {}
        k0 = int(kept_local[0]) if len(kept_local) else 0
        means, x, loc_off = None, None, None
        if held and F > 0:
            loc_off = np.array([g_lo[held[0]]] + [g_hi[g] for g in held], dtype=np.int64)
            x = r_e[:, :, int(loc_off[0]):int(loc_off[-1])].contiguous()
        # One host wait per step (round 5): the fused spectral lag path finishes on the device, so everything behind the
        # three calls — the halo window, the weighting of the lag means, the all-reduce, the copy to the host — is QUEUED
        # here, at issue time, and finish() only waits for the copy. (Round 6: the batched path for series of more than
        # 16 384 padded points finishes on the device too, so the one-wait order holds for every length.)
        # Measured on ONE GPU with the shards one of eight / four ranks holds, both orders in the same process (tools/
        # c4_shard_cost.py, C4_ORDER=two|one, profiles/r05_c4_shard.txt): 1.17 against 1.16 ms per step at 8, 2.01 against
        # 1.83 at 4 — boxes differ by more than that between leases, so orders are compared inside one run. On a
        # multi-GPU node the single wait also covers the collective's wire time. MDHIP_STEP_ONE_WAIT=0 restores round 4's order.
        one_wait = on_gpu and os.environ.get("MDHIP_STEP_ONE_WAIT", "1") != "0"
        issue_err = None
        if on_gpu:
            try:
                if hi > lo:
                    hs["single"] = backend.msd_origin(r_f, r0, goff, scale=scale, out=single[lo:hi], ctx=ctx, async_=True)
                if len(kept_local):
                    hs["fixed"] = backend.msd_windows(r_f[k0:], tao, scale=scale, out=win, ctx=ctx, async_=True)
                if x is not None:
                    means = torch.empty((n_lags, len(held), 4), dtype=torch.float64, device=dev)
                    # (the caller's own lag_variant is put back afterwards: ADVICE r05 — resetting to the default lost it)
                    user_variant = ctx.get_option("lag_variant", -1)
                    if one_wait:
                        ctx.set_option("lag_variant", 2)  # (the spectral path, no host-side fallback: the status word decides)
                    try:
                        hs["lag"] = backend.lag_msd(x, max_lag, loc_off - loc_off[0], scale=lag_scale, out=means, ctx=ctx,
                                                    async_=True, status_out=res[nS + nW + nL + 1:] if one_wait else None)
                    finally:
                        if one_wait:
                            ctx.set_option("lag_variant", user_variant)
            except Exception as e:  # (this rank still takes part in the all-reduce; every rank raises behind it)
                issue_err = e
                res[nS + nW + nL] = 1.0
        else:
            if hi > lo:
                single[lo:hi] = torch.as_tensor(compute["origin"](r_f.numpy(), r0.numpy(), goff, scale))
            if len(kept_local) > 1:
                win.copy_(torch.as_tensor(compute["windows"](r_f[k0:].numpy(), tao, scale)))
            if x is not None:
                means = torch.as_tensor(np.asarray(compute["lag"](x.numpy(), max_lag, loc_off - loc_off[0], lag_scale)))

        def post_ops():
            """What follows the kernels on the device: the halo window, the lag sums (stream-ordered behind the calls)."""
            if len(kept_local) and halo is not None:
                # the one window that reaches back to the rank below: three planes of arithmetic (the kernel's
                # operations: scale, subtract, square, (dx2 + dy2) + dz2)
                d2 = (r_f[k0] * scale - halo * scale) ** 2
                win.add_(torch.stack([d2[0], d2[1], d2[2], (d2[0] + d2[1]) + d2[2]], dim=1))
            if means is not None:
                w = _lag_weights(origins, g_hi, g_lo, held, dev)
                if len(held) == G:
                    torch.mul(means, w[:, :, None], out=lagsum)  # (one kernel: the product lands in the reduced buffer)
                else:
                    lagsum[:, torch.as_tensor(held, device=dev), :] = means * w[:, :, None]

        ev_done, flat_t, flat_np = None, None, None
        if one_wait:
            ev_k = torch.cuda.Event()
            ev_k.record(torch.cuda.current_stream(dev))
            s2 = _step_stream(dev, ctx, post=True)
            s2.wait_event(ev_k)
            # (on the second stream: the next step's kernels queue behind this step's KERNELS only, not behind its dozen
            # small operations and its collective; everything these read is held by this step until finish())
            with torch.cuda.stream(s2):
                if issue_err is None:
                    post_ops()
                if world > 1:
                    # (the second communicator is opt-in, MDHIP_STEP_POST_GROUP=1: collectives of two communicators in
                    # flight at once have not been run on more than one GPU here — a one-rank RCCL group and gloo pairs
                    # are what this box allows — and a hang would cost a whole multi-GPU run; on the default group the
                    # next step's all-gather queues behind this all-reduce, which is slower but cannot interleave)
                    _allreduce_inplace(res, _post_group() if os.environ.get("MDHIP_STEP_POST_GROUP", "0") == "1" else None)
                flat_np, flat_t = _pinned_like(res)
                flat_t.copy_(res, non_blocking=True)
                ev_done = torch.cuda.Event()
                ev_done.record(s2)
    keep = (r_f, r_e, allf, r0, x, means)  # what the queued kernels read stays alive until the step has been waited for

    def finish():
        if one_wait:
            ev_done.synchronize()  # THE host wait of the step
            flat = flat_np
            err = issue_err
            try:
                for key, h in hs.items():
                    h.wait()  # (complete: the status of each call, its times)
                    stats[key] = h.stats()
            except Exception as e:
                err = err or e
            if world > 1:
                # Errors that only show at COMPLETION (a call's deferred re-run failing, EHIP) are not on the reduced
                # failure flag — that all-reduce ran before the host knew. The ranks agree on them here, before anyone
                # raises or enters another collective (the redo below, the next step's exchange): one word, summed.
                # (a rank raising alone left the others waiting in their next collective until it timed out.)
                agree = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=dev)
                with torch.cuda.stream(_step_stream(dev, ctx, post=True)):
                    _allreduce_inplace(agree, _post_group() if os.environ.get("MDHIP_STEP_POST_GROUP", "0") == "1" else None)
                    n_failed = int(round(float(agree.item())))
                if n_failed and err is None:
                    err = RuntimeError("%d rank(s) failed completing their share of the MSD step; this rank stops with them"
                                       % n_failed)
            if err is not None:
                raise err
            if flat[nS + nW + nL] != 0.0:
                raise RuntimeError("%d rank(s) failed on their share of the MSD step; this rank stops with them"
                                   % int(round(flat[nS + nW + nL])))
            status = flat[nS + nW + nL + 1]  # (the SUM of the ranks' bounds: at least the largest of them)
            if not (status <= 1e-10):
                # some rank's spectral result is not good enough (or was rewritten at completion): every rank sees the same
                # reduced status, so every rank repeats the lag part, together, under lag_variant 3 — the library's own
                # decision per call (round 6): the spectral result stands where the rank's bound holds (the status is the SUM
                # of the ranks' bounds: it can pass the tolerance with every rank inside it), the few lags that miss it are
                # recomputed from the difference form, and only data that misses it broadly goes to the difference kernel
                flat = flat.copy()
                lag_part = torch.zeros(nL, dtype=torch.float64, device=dev)
                with torch.cuda.stream(_step_stream(dev, ctx, post=True)):
                    if x is not None:
                        user_variant = ctx.get_option("lag_variant", -1)
                        ctx.set_option("lag_variant", 3)
                        try:
                            m2 = torch.empty((n_lags, len(held), 4), dtype=torch.float64, device=dev)
                            backend.lag_msd(x, max_lag, loc_off - loc_off[0], scale=lag_scale, out=m2, ctx=ctx)
                        finally:
                            ctx.set_option("lag_variant", user_variant)
                        w = _lag_weights(origins, g_hi, g_lo, held, dev)
                        lv = lag_part.view(n_lags, G, 4)
                        if len(held) == G:
                            lv.copy_(m2 * w[:, :, None])
                        else:
                            lv[:, torch.as_tensor(held, device=dev), :] = m2 * w[:, :, None]
                    if world > 1:
                        _allreduce_inplace(lag_part)
                    flat[nS + nW:nS + nW + nL] = lag_part.cpu().numpy()
                stats["lag_redone"] = float(status)
        else:
            # On a stream of its own: the next step's kernels may already be queued on the issue stream, and nothing
            # here has to wait for them (what it reads is complete: the calls are waited for first).
            with (torch.cuda.stream(_step_stream(dev, ctx, post=True)) if on_gpu else contextlib.nullcontext()):
                err = issue_err
                try:
                    if err is None:
                        for key, h in hs.items():
                            h.wait()  # the calls have completed: `means` is in place (the batched path's finish ran here)
                            stats[key] = h.stats()
                        post_ops()
                except Exception as e:  # agreed on THROUGH the all-reduce (no collective of its own)
                    err = e
                    res[nS + nW + nL] = 1.0
                if world > 1:
                    # (MDHIP_STEP_POST_GROUP=1: the closing all-reduce on the second communicator, as in the one-wait order
                    # above — also on this path so that the two-communicator pattern runs under gloo in the CPU suite)
                    _allreduce_inplace(res, _post_group() if os.environ.get("MDHIP_STEP_POST_GROUP", "0") == "1" else None)
                flat = res.cpu().numpy()
                if err is not None:
                    raise err
                if flat[nS + nW + nL] != 0.0:
                    raise RuntimeError("%d rank(s) failed on their share of the MSD step; this rank stops with them"
                                       % int(round(flat[nS + nW + nL])))
        assert keep is not None
        single_h = flat[:nS].reshape(F, G, 4)
        win_h = flat[nS:nS + nW].reshape(E, 4)
        with np.errstate(invalid="ignore", divide="ignore"):
            lag_h = np.where(lag_counts[:, :, None] > 0,
                             flat[nS + nW:nS + nW + nL].reshape(n_lags, G, 4) / np.maximum(lag_counts, 1.0)[:, :, None],
                             0.0)
        return single_h, win_h, lag_h, stats

    return _Deferred(finish)


def msd_step_sharded(*args, **kwargs):
    """msd_step_sharded_async, waited for: (single [F,G,4], windows [E,4], lag [F,G,4], per-call kernel times)."""
    return msd_step_sharded_async(*args, **kwargs).wait()


class _Deferred:
    """Handle of a step whose calls have been issued: wait() runs what is left (once) and returns the result."""

    def __init__(self, finish):
        self._finish, self._out = finish, None

    def wait(self):
        if self._finish is not None:
            self._out = self._finish()
            self._finish = None
        return self._out


def entity_shard(n_entities, rank=None, world=None):
    """Contiguous block [lo, hi) of entities owned by `rank` (lag_msd_sharded); sizes differ by at most one."""
    return frame_shard(n_entities, rank, world)


def lag_ranges(n, n_lags, world):
    """Boundaries k_0 = 0 < k_1 < ... <= k_world = n_lags of lag ranges of equal work for the direct correlation
    estimator: lag k costs n - k products, so the cumulative work up to lag k is k (2n - k + 1) / 2."""
    n, n_lags, world = int(n), int(n_lags), int(world)
    total = n_lags * (2.0 * n - n_lags + 1.0) / 2.0
    b = [0]
    for r in range(1, world):
        w = total * r / world
        # smallest k with k (2n - k + 1) / 2 >= w
        k = int(np.ceil(((2.0 * n + 1.0) - np.sqrt(max((2.0 * n + 1.0) ** 2 - 8.0 * w, 0.0))) / 2.0))
        b.append(min(max(k, b[-1]), n_lags))
    b.append(n_lags)
    return b


def xcorr_direct_sharded(a, b=None, n_lags=None, compute=None, ctx=None):
    """
    Direct ("brute_force", viscosity.py:103-108) correlation with the LAGS sharded: every rank holds the whole
    series a, b [n] or [P,n] (b=None: autocorrelation), computes the lags of its range (equal work per rank, see
    lag_ranges) and the slices are all-gathered: c [n_lags] or [P,n_lags] on every rank. Every lag is computed by
    exactly one rank with the same kernel; the time slabs of a launch depend on its lag range, so the result equals
    the single-GPU one to rounding (~1e-14 relative), well inside the 1e-10 acf[0] bar.
    Series as CUDA tensors: a rank's slice [P, k1-k0] is written to a device buffer (mdhip_xcorr_lags_dev) and gathered
    from there (8 MB per series at n = 1e6); one copy to the host at the end.
    """
    dev = compute is None and _is_tensor(a) and a.is_cuda
    if compute is None:
        from . import backend

        def compute(aa, bb, k0, nl):
            return backend.xcorr(aa, bb, method=backend.XCORR_DIRECT, n_lags=nl, ctx=ctx, lag_begin=k0)

    single = len(a.shape) == 1
    n = int(a.shape[-1])
    P = 1 if single else int(a.shape[0])
    n_lags = n if n_lags is None else int(n_lags)
    rank, world = rank_world()
    bounds = lag_ranges(n, n_lags, world)
    k0, k1 = bounds[rank], bounds[rank + 1]
    widths = [bounds[r + 1] - bounds[r] for r in range(world)]
    if dev:
        import torch

        from . import backend

        mine = torch.empty((P, k1 - k0), dtype=torch.float64, device=a.device)
        if k1 > k0:
            backend.xcorr(a, b, method=backend.XCORR_DIRECT, n_lags=k1 - k0, ctx=ctx, lag_begin=k0, out=mine)
        rows = mine.t().contiguous()  # rows = lags, so that the gather concatenates along the lag axis
        out = (allgather_var(rows, counts=widths) if is_distributed() else rows).t().contiguous().cpu().numpy()
        return out[0] if single else out
    mine = np.asarray(compute(a, b, k0, k1 - k0)).reshape(P, k1 - k0) if k1 > k0 else np.zeros((P, 0))
    if not is_distributed():
        out = mine
    else:
        out = allgather_var(np.ascontiguousarray(mine.T), counts=widths).T
    out = np.ascontiguousarray(out)
    return out[0] if single else out
