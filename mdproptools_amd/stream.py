"""
Streaming ingest: native dump reader -> pinned staging buffers -> H2D -> kernels, with the
text of the NEXT batch of frames being parsed while the GPU works on the current one.

The reference materialises every frame of a trajectory before its first pair loop
(/root/reference/mdproptools/structural/rdf_cn.py:176 `dumps = list(parse_lammps_dumps(filename))`,
dynamical/diffusion.py:172); at BASELINE C3 size that is 2.4 GB of coordinates behind ~5 GB of text. Here a
producer thread parses files (mdhip_dump_*; several files in parallel, the C calls release the GIL) into a small
ring of page-locked buffers [B,3,N]; the consumer — a drop-in function — hands one buffer at a time to the
library, whose host->device copy of a pinned buffer is a DMA at the PCIe rate, and gives the buffer back. Host
memory in use is `depth` batches, whatever the length of the trajectory.

    for batch in FrameStream("dump.nvt.*.dump"):
        full, part, ov = backend.rdf_loop(batch.xyz, batch.types, batch.lengths, ...)
        batch.release()

Timings for the parse / copy+kernel split are kept in `stream.stats` (parse_s is summed over the reader threads).
"""

import ctypes as C
import queue
import threading
import time

import numpy as np

from . import io as mio

DEFAULT_BATCH_BYTES = 24 << 20  # coordinates per batch (100 frames of 10k atoms, 10 of 100k: a few batches even for short runs)
DEFAULT_DEPTH = 3               # staging buffers in the ring
MIN_BATCH_FRAMES = 32           # ... but at least this many frames per batch while they fit MAX_BATCH_BYTES: the pair
MAX_BATCH_BYTES = 128 << 20     # kernels' per-call costs (pre-pass, launch tails) want tens of frames per call — 10-frame
                                # batches of 100k atoms cost 0.30 s of device time per 1000 frames against 0.17 s in one call


def frames_per_batch(batch_bytes, n_atoms, exact=False):
    """Frames of n_atoms atoms per staging batch: batch_bytes of coordinates — and, unless the caller fixed the batch
    size itself (`exact`), raised to MIN_BATCH_FRAMES while that stays within MAX_BATCH_BYTES."""
    per = max(1, 24 * int(n_atoms))
    cap = max(1, int(batch_bytes) // per)
    if not exact and cap < MIN_BATCH_FRAMES:
        cap = max(cap, min(MIN_BATCH_FRAMES, MAX_BATCH_BYTES // per))
    return max(1, cap)


def _rank_device():
    """The GPU this process works on (one rank per GPU: LOCAL_RANK, wrapped when ranks share a card). The staging
    buffers are allocated from reader threads, whose current HIP device would otherwise be 0 on every rank."""
    import os

    dev = int(os.environ.get("LOCAL_RANK", "0"))
    try:
        import torch

        n = torch.cuda.device_count()
        if n > 0:
            dev %= n
    except Exception:
        pass
    return dev


class _Pinned:
    """A page-locked float64 buffer from libmdhip.so (falls back to pageable numpy memory without a HIP runtime)."""

    def __init__(self, n_doubles, device=None):
        self.ptr = None
        self.lib = None
        self.array = None
        try:
            from . import _lib

            lib = _lib.load()
            p = C.c_void_p()
            dev = _rank_device() if device is None else int(device)
            if lib.mdhip_host_alloc_on(dev, C.c_size_t(n_doubles * 8), C.byref(p)) == 0 and p.value:
                self.ptr, self.lib = p, lib
                self.array = np.ctypeslib.as_array((C.c_double * n_doubles).from_address(p.value))
        except Exception:
            pass
        if self.array is None:
            self.array = np.empty(n_doubles, dtype=np.float64)

    @property
    def pinned(self):
        return self.ptr is not None

    def free(self):
        if self.ptr is not None:
            self.array = None
            self.lib.mdhip_host_free(self.ptr)
            self.ptr = None


# Page-locking memory costs ~10 ms per 64 MB: the staging buffers of finished streams are kept (a few, process-wide)
# and handed to the next stream instead of being unpinned and pinned again.
_CACHE = []
_CACHE_MAX = 4
_CACHE_LOCK = threading.Lock()


def _take_buffer(n_doubles):
    with _CACHE_LOCK:
        for k, b in enumerate(_CACHE):
            if b.array is not None and b.array.size >= n_doubles:
                return _CACHE.pop(k)
    return _Pinned(n_doubles)


def _give_back(buf):
    with _CACHE_LOCK:
        if buf.array is not None and len(_CACHE) < _CACHE_MAX:
            _CACHE.append(buf)
            return
    buf.free()


class Frame:
    """One frame of a batch: views into the batch buffers (valid until the batch is released)."""

    __slots__ = ("timestep", "ids", "types", "xyz", "lengths")

    def __init__(self, timestep, ids, types, xyz, lengths):
        self.timestep, self.ids, self.types, self.xyz, self.lengths = timestep, ids, types, xyz, lengths


class Batch:
    """Consecutive frames with the same atom count: xyz [B,3,N] (one contiguous, page-locked block), ids / types
    [B,N], lengths [B,3], timesteps [B]. Iterating yields `Frame` views."""

    # True when the library's reader threads found the second column (types) of EVERY frame of this batch bit-identical
    # to that of the stream's first frame (`types_ref`): callers then derive labels, counts and densities once
    uniform_types = False
    types_ref = None

    def __init__(self, stream, buf, n_frames, n_atoms, ids, types, lengths, timesteps):
        self._stream, self._buf = stream, buf
        self.xyz = buf.array[: n_frames * 3 * n_atoms].reshape(n_frames, 3, n_atoms)
        self.ids, self.types, self.lengths, self.timesteps = ids, types, lengths, timesteps

    def __len__(self):
        return self.xyz.shape[0]

    def __iter__(self):
        for k in range(len(self)):
            yield Frame(int(self.timesteps[k]), self.ids[k], self.types[k], self.xyz[k], tuple(self.lengths[k]))

    def release(self):
        """Give the staging buffer back to the producer (the library call that read it has returned)."""
        if self._buf is not None:
            self._stream._free.put(self._buf)
            self._buf = None


class FrameStream:
    """Iterator of `Batch` over the frames of `file_pattern` (numeric file order, atoms sorted by id)."""

    def __init__(self, file_pattern, files=None, batch_bytes=None, depth=DEFAULT_DEPTH,
                 columns=("id", "type", "x", "y", "z"), on_frame=None):
        self.pattern, self.files = str(file_pattern), files
        self._exact = batch_bytes is not None  # a caller-given batch size is taken literally
        self.batch_bytes, self.depth = int(batch_bytes or DEFAULT_BATCH_BYTES), max(2, int(depth))
        self.columns = list(columns)
        self.on_frame = on_frame
        self.stats = {"parse_s": 0.0, "wait_for_buffer_s": 0.0, "frames": 0, "batches": 0, "pinned": None,
                      "consumer_wait_s": 0.0}
        self._ready = queue.Queue(maxsize=self.depth)
        self._free = queue.Queue()
        self._bufs = []
        self._error = None
        self._thread = None
        self._closed = False

    # ---- producer -------------------------------------------------------------------------------------------------
    def _get_buffer(self, n_doubles):
        t0 = time.perf_counter()
        while True:
            if len(self._bufs) < self.depth:
                b = _take_buffer(n_doubles)
                self._bufs.append(b)
                if self.stats["pinned"] is None:
                    self.stats["pinned"] = b.pinned
                break
            b = self._free.get()
            if b is None:
                return None
            if b.array.size >= n_doubles:
                break
            self._bufs.remove(b)  # a longer frame than the ring was sized for: replace the buffer
            b.free()
        self.stats["wait_for_buffer_s"] += time.perf_counter() - t0
        return b

    def _produce(self):
        """Two stages, both on a small thread pool (the C calls release the GIL): (A) open + index the next files, a
        bounded number ahead; (B) in file order, give every frame its slot in the current batch and parse its x, y, z
        planes STRAIGHT into that slot of the page-locked buffer (ids and types into small arrays of their own). A
        batch is handed to the consumer when the parses of all its frames are done."""
        import os
        from concurrent.futures import ThreadPoolExecutor

        if any(str(f).endswith(".gz") for f in (self.files or mio._sorted_matches(self.pattern))):
            return self._produce_by_copy()  # compressed text goes through the pandas route
        try:
            files = self.files if self.files is not None else mio._sorted_matches(self.pattern)
            done = self._produce_whole_files(files)
            if done is None:
                return
            files = files[done:]  # (what the batch reader did not take: multi-frame files, changing atom counts)
            workers = int(os.environ.get("MDHIP_STREAM_WORKERS", "0")) or min(max(1, (os.cpu_count() or 1) // 2), 32)
            ahead = 2 * workers  # files opened (indexed) ahead of the one being placed
            with ThreadPoolExecutor(max_workers=workers) as pool:
                def open_file(fn):
                    t0 = time.perf_counter()
                    nd = mio.NativeDumpFile(fn)
                    heads = [nd.header(f) for f in range(nd.n_frames)]
                    return nd, heads, time.perf_counter() - t0

                def parse(nd, f, names, dests, pending):
                    # `pending` = [frames of this file still to be parsed, lock]: the frames of a multi-frame file are
                    # parsed concurrently, whoever finishes LAST closes the mapping
                    t0 = time.perf_counter()
                    try:
                        mio.native_read_into(nd, f, names, self.columns, dests, sort_by="id", n_threads=1)
                    finally:
                        with pending[1]:
                            pending[0] -= 1
                            last = pending[0] == 0
                        if last:
                            nd.close()
                    return time.perf_counter() - t0

                opens = []
                nxt = 0

                def top_up():
                    nonlocal nxt
                    while nxt < len(files) and len(opens) < ahead:
                        opens.append(pool.submit(open_file, files[nxt]))
                        nxt += 1

                cur = None  # [buf, n, cap, ids [cap,n], types [cap,n], lengths, steps, parse futures]

                def flush():
                    # the batch goes to the consumer WITH its parse futures still running (the consumer waits for them):
                    # the producer moves on to place the frames of the next batches, so up to `depth` batches of
                    # frames are being parsed at once
                    nonlocal cur
                    if cur is None:
                        return
                    buf, n, _cap, ids, types, lengths, steps, futs = cur
                    B = len(steps)
                    self._ready.put((Batch(self, buf, B, n, ids[:B], types[:B], np.array(lengths),
                                           np.array(steps, dtype=np.int64)), futs))
                    self.stats["batches"] += 1
                    cur = None

                top_up()
                while opens:
                    if self._closed:
                        return
                    nd, heads, t_open = opens.pop(0).result()
                    self.stats["parse_s"] += t_open
                    top_up()
                    pending = [len(heads), threading.Lock()]
                    for f, (ts, n, bounds, tilt, names) in enumerate(heads):
                        if cur is not None and (cur[1] != n or len(cur[6]) >= cur[2]):
                            flush()
                        if cur is None:
                            cap = frames_per_batch(self.batch_bytes, n, self._exact)
                            buf = self._get_buffer(cap * 3 * n)
                            if buf is None:
                                return
                            cur = [buf, n, cap, np.empty((cap, n)), np.empty((cap, n)), [], [], []]
                        k = len(cur[6])
                        slot = cur[0].array[k * 3 * n:(k + 1) * 3 * n].reshape(3, n)
                        ids, types = cur[3][k], cur[4][k]
                        cur[5].append(mio.LammpsBox(bounds.tolist(), tilt).to_lattice().lengths)
                        cur[6].append(ts)
                        cur[7].append(pool.submit(parse, nd, f, names, [ids, types, slot[0], slot[1], slot[2]], pending))
                        self.stats["frames"] += 1
                    if not heads:
                        nd.close()
                flush()
        except BaseException as e:  # handed to the consumer
            self._error = e
        finally:
            self._ready.put(None)

    def _produce_whole_files(self, files):
        """The common trajectory — one frame per file, the same atoms in every file — a BATCH of files per library
        call (mdhip_dump_read_files): the library's own threads open, index and parse the files of a batch and place
        x, y, z straight into the frame slots of the page-locked buffer, ids and types into their tables; Python does
        per-batch work only (per-file Python — open, header, a future per frame — cost ~0.2 ms a file: as much as the
        parsing itself for 10 000-atom frames). Returns the number of files consumed (the caller's per-frame route takes
        the rest: multi-frame files, a changing atom count, a missing column), or None when the stream was closed."""
        import ctypes as C
        import os

        if not files:
            return 0
        # the batch call below places exactly FIVE columns (id, one per-atom attribute, three planes) and orders rows by
        # "id": any other column set takes the per-frame route, which is general
        if len(self.columns) != 5 or self.columns[0] != "id":
            return 0
        try:
            from . import _lib

            lib = _lib.load()
        except Exception:
            return 0
        threads = int(os.environ.get("MDHIP_STREAM_WORKERS", "0")) or min(max(1, (os.cpu_count() or 1) // 2), 32)
        done = 0
        n = None
        cols = (C.c_char_p * len(self.columns))(*[c.encode() for c in self.columns])
        while done < len(files):
            if self._closed:
                return None
            if n is None:
                nd = mio.NativeDumpFile(files[done])
                try:
                    if nd.n_frames != 1:
                        return done
                    n = nd.header(0)[1]
                finally:
                    nd.close()
                if n < 1:
                    return done
            cap = frames_per_batch(self.batch_bytes, n, self._exact)
            chunk = files[done:done + cap]
            B = len(chunk)
            buf = self._get_buffer(cap * 3 * n)
            if buf is None:
                return None
            ids, types = np.empty((B, n)), np.empty((B, n))
            steps = np.empty(B, dtype=np.int64)
            bounds, tilt = np.empty((B, 6)), np.empty((B, 3))
            tri = np.zeros(B, dtype=np.int32)
            base = buf.array.ctypes.data
            dptr = C.POINTER(C.c_double)
            dst = (dptr * 5)(C.cast(ids.ctypes.data, dptr), C.cast(types.ctypes.data, dptr), C.cast(base, dptr),
                             C.cast(base + 8 * n, dptr), C.cast(base + 16 * n, dptr))
            stride = (C.c_int64 * 5)(n, n, 3 * n, 3 * n, 3 * n)
            paths = (C.c_char_p * B)(*[str(f).encode() for f in chunk])
            err = C.create_string_buffer(512)
            same = np.zeros(B, dtype=np.int32)
            ref = getattr(self, "_types_ref", None)
            if ref is not None and ref.shape[0] != n:
                ref = None
            t0 = time.perf_counter()
            rc = lib.mdhip_dump_read_files(paths, B, 5, cols, b"id", n, dst, stride,
                                           steps.ctypes.data_as(C.POINTER(C.c_int64)),
                                           bounds.ctypes.data_as(dptr), tilt.ctypes.data_as(dptr),
                                           tri.ctypes.data_as(C.POINTER(C.c_int32)), threads, err, 512, 1,
                                           None if ref is None else ref.ctypes.data_as(dptr),
                                           same.ctypes.data_as(C.POINTER(C.c_int32)))
            self.stats["parse_s"] += (time.perf_counter() - t0) * min(threads, B)  # (upper bound: wall x threads)
            if rc == 1 or (rc == 0 and tri.any()):
                # some file of the chunk is not a plain single frame of n atoms in an orthogonal box: the per-frame
                # route from here (it applies the tilt correction of the bounds)
                self._free.put(buf)
                return done
            if rc != 0:
                self._free.put(buf)
                raise ValueError(err.value.decode() or "mdhip_dump_read_files failed (%d)" % rc)
            # orthogonal cell: |row| of the diagonal cell matrix = hi - lo, as LammpsBox.to_lattice().lengths computes it
            # (sqrt(a * a) == |a| exactly)
            lengths = np.abs(np.column_stack([bounds[:, 1] - bounds[:, 0], bounds[:, 3] - bounds[:, 2],
                                              bounds[:, 5] - bounds[:, 4]]))
            if ref is None:
                ref = self._types_ref = types[0].copy()
            b = Batch(self, buf, B, n, ids, types, lengths, steps)
            b.uniform_types, b.types_ref = bool(same.all()), ref
            self._ready.put((b, []))
            self.stats["batches"] += 1
            self.stats["frames"] += B
            done += B
        return done

    def _produce_by_copy(self):
        """Fallback producer (compressed dumps): frames from the generic reader, copied into the staging slots."""
        try:
            cur = None  # (buf, n_atoms, cap, ids, types, lengths, steps)
            t_parse = time.perf_counter()

            def flush():
                nonlocal cur
                if cur is None:
                    return
                buf, n, _cap, ids, types, lengths, steps = cur
                self._ready.put((Batch(self, buf, len(steps), n, np.stack(ids), np.stack(types), np.array(lengths),
                                       np.array(steps, dtype=np.int64)), []))
                self.stats["batches"] += 1
                cur = None

            it = mio.iter_native_frames(self.pattern, self.columns, sort_by="id", files=self.files, workers=1)
            for ts, _bounds, lengths, _names, planes in it:
                if self._closed:
                    return
                self.stats["parse_s"] += time.perf_counter() - t_parse
                n = planes.shape[1]
                if cur is not None and (cur[1] != n or len(cur[6]) >= cur[2]):
                    flush()
                if cur is None:
                    cap = frames_per_batch(self.batch_bytes, n, self._exact)
                    buf = self._get_buffer(cap * 3 * n)
                    if buf is None:
                        return
                    cur = (buf, n, cap, [], [], [], [])
                k = len(cur[6])
                np.copyto(cur[0].array[k * 3 * n:(k + 1) * 3 * n].reshape(3, n), planes[2:5])
                cur[3].append(planes[0])
                cur[4].append(planes[1])
                cur[5].append(lengths)
                cur[6].append(ts)
                self.stats["frames"] += 1
                t_parse = time.perf_counter()
            flush()
        except BaseException as e:  # handed to the consumer
            self._error = e
        finally:
            self._ready.put(None)

    # ---- consumer -------------------------------------------------------------------------------------------------
    def __iter__(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._produce, name="mdhip-frame-stream", daemon=True)
            self._thread.start()
        try:
            while True:
                t0 = time.perf_counter()
                item = self._ready.get()
                if item is None:
                    self.stats["consumer_wait_s"] += time.perf_counter() - t0
                    break
                batch, futs = item
                for fu in futs:
                    self.stats["parse_s"] += fu.result()  # (summed over the pool's threads)
                self.stats["consumer_wait_s"] += time.perf_counter() - t0
                if self.on_frame is not None:
                    for ts in batch.timesteps:
                        self.on_frame(int(ts))
                yield batch
                batch.release()
            if self._error is not None:
                raise self._error
        finally:
            self.close()

    def close(self):
        self._closed = True
        self._free.put(None)
        deadline = time.perf_counter() + 30.0
        while True:  # keep the ready queue drained until the producer (and its parse tasks) are gone
            try:
                while self._ready.get_nowait() is not None:
                    pass
            except queue.Empty:
                pass
            if self._thread is None:
                break
            self._thread.join(timeout=0.05)
            if not self._thread.is_alive() or time.perf_counter() > deadline:
                break
        if self._thread is not None and self._thread.is_alive():
            # parse tasks that were already submitted may still be scattering values into the staging buffers (a
            # slow file system, very large frames): the buffers are LEAKED — neither unpinned nor handed to the next
            # stream — rather than freed under a writer
            self._bufs = []
            return
        for b in self._bufs:
            _give_back(b)
        self._bufs = []
