"""
Radial distribution functions and coordination numbers from LAMMPS dumps —
drop-in for /root/reference/mdproptools/structural/rdf_cn.py (same public
functions, positional order, defaults, return types, CSV side effects and
exception types: rdf_cn.py:385-396, 533-544, 654-665, 759-770).

What runs where
  GPU (libmdhip.so): every pair loop — `_rdf_loop`, `_cn_loop`,
      `_rdf_mol_loop`, `_cn_mol_loop` (rdf_cn.py:72-162) — and the molecule
      centres of mass (`_define_mol_cols`, rdf_cn.py:218-241), for all frames
      of a batch in one call; integer histograms come back per frame.
  Host (numpy, this file): parsing, id sorting, the altered-id remap
      (rdf_cn.py:197-215, vectorised), densities and the per-frame
      normalisation in the reference's operation order (rdf_cn.py:288-291,
      312-328), the frame average and the CSV. Because the histograms are
      exact integers and the normalisation repeats the reference's float
      operations in order, g(r) and CN come out bit-identical.

Differences, all deliberate:
  * pairs whose bin index would be == num_bins (reachable when r_cut/bin_size
    rounds up, e.g. 20/0.05) are dropped and reported instead of written out of
    bounds (the numba build corrupts memory there; plain numpy raises);
  * progress lines are printed only when `rdf_cn.VERBOSE` is true.
"""

from timeit import default_timer as timer

import numpy as np
import pandas as pd

from .. import backend
from ..io import parse_lammps_dumps

CON_CONSTANT = 1.660538921  # amu/A^3 -> g/cm^3 (rdf_cn.py:30)
VERBOSE = False
MAX_BATCH_BYTES = 1 << 30  # coordinates staged per library call
STREAM = True  # parse the next batch of frames (into page-locked staging buffers) while the GPU runs the current one;
               # the frames of a trajectory are then never all resident on the host (mdproptools_amd/stream.py)

_R_LABEL = "r ($\\AA$)"


def _say(*args):
    if VERBOSE:
        print(*args)


# ------------------------------------------------------------------------------------------------
# host-side pieces
# ------------------------------------------------------------------------------------------------


def _initialize(r_cut, bin_size, filename, partial_relations):
    """Bins, radii, parsed frames and relation count (rdf_cn.py:165-180)."""
    if isinstance(r_cut, list):
        num_bins = [int(rc / bin_size) for rc in r_cut]
        radii = [(np.arange(nb) + 0.5) * bin_size for nb in num_bins]
    else:
        num_bins = int(r_cut / bin_size)
        radii = (np.arange(num_bins) + 0.5) * bin_size
    dumps = _load_frames(filename, shard=True, stream=STREAM)
    n_known = len(dumps) if hasattr(dumps, "__len__") else None  # (a stream knows its length only at its end)
    return dumps, num_bins, radii, n_known, len(partial_relations[0])


def _calc_atom_type(ids, num_mols, num_atoms):
    """
    Atom id -> 1-based index of the atom inside its molecule type, offset by the atom counts of
    the preceding molecule types (rdf_cn.py:197-215), vectorised over all atoms.
    """
    ids = np.asarray(ids, dtype=np.float64)
    num_atoms = np.asarray(num_atoms)
    upper = np.cumsum(np.multiply(num_mols, num_atoms))
    which = np.searchsorted(upper, ids, side="left")  # first molecule type whose range holds the id
    out = ids.copy()
    inside = which < len(upper)
    w = which[inside]
    v = np.mod(ids[inside] - upper[w], num_atoms[w])  # Python-style modulo of a non-positive number
    v[v == 0] = num_atoms[w][v == 0]
    out[inside] = v + np.concatenate(([0], np.cumsum(num_atoms)[:-1]))[w]
    return out


def _molecule_layout(num_mols, num_atoms_per_mol):
    """Segment offsets and molecule type labels implied by the sorted-id order (rdf_cn.py:222-230)."""
    counts = np.repeat(np.asarray(num_atoms_per_mol, dtype=np.int64), np.asarray(num_mols, dtype=np.int64))
    seg_off = np.concatenate(([0], np.cumsum(counts))).astype(np.int64)
    seg_type = np.repeat(np.arange(1, len(num_mols) + 1), np.asarray(num_mols, dtype=np.int64))
    return seg_off, seg_type.astype(np.int32)


def _type_counts(labels):
    """{label: number of atoms} — what `np.unique(..., return_counts=True)` gives (rdf_cn.py:253-256), by a counting
    pass when the labels are small non-negative integers (always, for LAMMPS types and altered ids): O(n), no sort."""
    lab = np.asarray(labels).astype(np.int64)
    if lab.size and lab.min() >= 0 and lab.max() < (1 << 20):
        cnt = np.bincount(lab)
        vals = np.flatnonzero(cnt)
        return {int(v): int(cnt[v]) for v in vals}
    vals, cnt = np.unique(lab, return_counts=True)
    return {int(v): int(c) for v, c in zip(vals, cnt)}


_COUNTS_MEMO = {"lab": None, "counts": None}


def _type_counts_memo(labels):
    """_type_counts with a one-entry memo: the frames of a trajectory nearly always carry the same labels, and the
    comparison (a memcmp) is several times cheaper than the count."""
    m = _COUNTS_MEMO
    lab = np.asarray(labels)
    if m["lab"] is not None and m["lab"].shape == lab.shape and np.array_equal(m["lab"], lab):
        return m["counts"]
    m["lab"], m["counts"] = lab.copy(), _type_counts(lab)
    return m["counts"]


def _calc_props(box_lengths, ref_labels, obj_labels, num_types, mass, partial_relations, altered,
                num_atoms_per_mol=None):
    """
    Densities and consistency checks of one frame (rdf_cn.py:244-294).
    Returns (rho, rho_pairs, atom_types, object_types).
    """
    n_objects = len(obj_labels)
    volume = np.prod(box_lengths)
    atom_types = _type_counts_memo(ref_labels)
    object_types = atom_types if obj_labels is ref_labels else _type_counts(obj_labels)
    expected = np.sum(num_atoms_per_mol) if altered else num_types
    if expected != len(atom_types):
        raise ValueError(
            "Consistency check failed: Number of specified atomic types is different from the "
            f"calculated value specified= {num_atoms_per_mol if altered else num_types}, "
            f"calculated= {len(atom_types)}")
    # rdf_cn.py:280-282 — needs the labels 1..num_types to be present (KeyError otherwise)
    total_mass = np.sum([float(mass[i]) * float(atom_types[i + 1]) for i in range(num_types)])
    total_density = float((total_mass / volume) * CON_CONSTANT)
    _say("{0:s}{1:10.8f}".format("Average density=", total_density))
    rho = n_objects / volume
    rho_pairs = np.zeros(len(partial_relations[1]))
    for k, obj in enumerate(partial_relations[1]):
        rho_pairs[k] = object_types[obj] / volume
        if rho_pairs[k] < 1.0e-22:
            raise ValueError("Error: Density is zero for mol type: " + str(obj))
    return rho, rho_pairs, atom_types, object_types


def _shell_volume(bin_size, num_bins):
    edges3 = np.arange(1, num_bins + 1) ** 3 - np.arange(num_bins) ** 3
    return 4 / 3 * np.pi * bin_size ** 3 * edges3  # rdf_cn.py:312-318


def _normalize_rdf(bin_size, rho_pairs, atom_types, partial_relations, num_relations, num_bins, rdf_part,
                   rdf_full=None, num_atoms=None, rho=None):
    """Per-frame normalisation (rdf_cn.py:297-329); operation order kept for bit-identical g(r)."""
    sv = _shell_volume(bin_size, num_bins)
    if rdf_full is not None:
        rdf_full = rdf_full / (num_atoms * rho * sv)
    n_ref = np.array([atom_types[a] for a in partial_relations[0]]).reshape(num_relations, 1)
    n_ref = np.tile(n_ref, num_bins)
    rho_b = np.tile(rho_pairs.reshape(num_relations, 1), num_bins)
    sv_m = np.tile(sv, (num_relations, 1))
    rdf_part = rdf_part / (n_ref * rho_b * sv_m)
    return rdf_full, rdf_part


def _normalize_rdf_batch(bin_size, props, partial_relations, num_relations, num_bins, part, full=None, num_atoms=None):
    """`_normalize_rdf` for all B frames of a batch in ONE numpy expression per output, each element through the
    reference's operations in the reference's order (rdf_cn.py:312-328):
        g_full[k][b]    = h / ((num_atoms_k * rho_k) * shell[b])
        g_part[k][r][b] = h / ((n_ref[k][r] * rho_pair[k][r]) * shell[b])
    (the reference tiles n_ref, rho_pair and the shell volumes to [R, nb] and multiplies left to right: the same two
    products per element, so the doubles are the same). part uint64 [B,R,nb], full uint64 [B,nb] or None.
    Returns rows [B, (1 + R) * nb] (or [B, R * nb] without `full`): g_full | g_part, as the per-frame code stacked them."""
    B = len(props)
    sv = _shell_volume(bin_size, num_bins)
    n_ref = np.array([[p[2][a] for a in partial_relations[0]] for p in props]).reshape(B, num_relations)
    rho_b = np.stack([p[1] for p in props]).reshape(B, num_relations)
    g_part = np.asarray(part).astype(np.float64) / ((n_ref * rho_b)[:, :, None] * sv[None, None, :])
    if full is None:
        return g_part.reshape(B, num_relations * num_bins)
    na_rho = np.array([n * p[0] for n, p in zip(num_atoms, props)])
    g_full = np.asarray(full).astype(np.float64) / (na_rho[:, None] * sv[None, :])
    return np.concatenate([g_full, g_part.reshape(B, num_relations * num_bins)], axis=1)


def _sum_frames(rows):
    """Sum of the per-frame rows in FRAME ORDER, one addition per frame and element as the reference's
    `rdf_full_sum += ...` loop (rdf_cn.py:514-515; numpy's own sum would add pairwise: other roundings)."""
    acc = np.zeros(rows.shape[1])
    for row in rows:
        acc += row
    return acc


_PROPS_MEMO = {"key": None, "val": None}


def _calc_props_memo(box_lengths, ref_labels, obj_labels, num_types, mass, partial_relations, altered,
                     num_atoms_per_mol=None):
    """_calc_props, computed once for consecutive frames with the same box and the same labels (NVT trajectories:
    every frame): the densities are functions of exactly those. Not used with VERBOSE (the density line is printed
    per frame there, as the reference does)."""
    if VERBOSE:
        return _calc_props(box_lengths, ref_labels, obj_labels, num_types, mass, partial_relations, altered,
                           num_atoms_per_mol)
    m = _PROPS_MEMO
    key = (tuple(float(x) for x in box_lengths), num_types, tuple(float(x) for x in mass),
           tuple(map(tuple, partial_relations)), altered, None if num_atoms_per_mol is None else tuple(num_atoms_per_mol))
    # (the SAME array object as last time — the stream hands `types_ref` to every batch whose frames the reader threads
    # found unchanged — needs no comparison; another array is compared value by value)
    same_labels = (m["key"] is not None and m["key"][0] == key
                   and (m.get("obj") is ref_labels
                        or (m["key"][1].shape == np.shape(ref_labels) and np.array_equal(m["key"][1], ref_labels)))
                   and (obj_labels is ref_labels or (m["key"][2] is not None and np.array_equal(m["key"][2], obj_labels))))
    if same_labels:
        return m["val"]
    val = _calc_props(box_lengths, ref_labels, obj_labels, num_types, mass, partial_relations, altered, num_atoms_per_mol)
    m["key"] = (key, np.array(ref_labels, copy=True), None if obj_labels is ref_labels else np.array(obj_labels, copy=True))
    m["obj"] = ref_labels
    m["val"] = val
    return val


_I32_MEMO = {"obj": None, "val": None}


def _labels_and_props(batch, altered, num_mols, num_atoms_per_mol, num_types, mass, partial_relations):
    """(labels for the library — int32 [N], or [F, N] when they change —, the per-frame `_calc_props` tuples) of one
    batch (rdf_cn.py:462-482). A streamed batch whose frames all carry the first frame's types (the reader threads
    compared them while the text was in their caches) takes one label array and, with a constant box, one set of
    densities for all its frames."""
    if not altered and getattr(batch, "uniform_types", False) and batch.types_ref is not None:
        lab = batch.types_ref
        props = [_calc_props_memo(f.lengths, lab, lab, num_types, mass, partial_relations, altered, num_atoms_per_mol)
                 for f in batch]
        if _I32_MEMO["obj"] is not lab:
            _I32_MEMO["obj"], _I32_MEMO["val"] = lab, lab.astype(np.int32)
        return _I32_MEMO["val"], props
    labels = [(_calc_atom_type(f.ids, num_mols, num_atoms_per_mol) if altered else f.types) for f in batch]
    props = [_calc_props_memo(f.lengths, lab, lab, num_types, mass, partial_relations, altered, num_atoms_per_mol)
             for f, lab in zip(batch, labels)]
    return _labels_for(batch, labels), props


def _write_csv(df, path_or_buf):
    """`df.to_csv(path_or_buf, index=False)` — byte for byte — for the all-float64 frames these functions write,
    without pandas' per-cell formatting machinery (20 ms for 400 x 12 values, a quarter of a C2-size call): pandas
    writes the shortest round-trip decimal of every double, which is Python's repr, and an empty field for NaN."""
    simple = (isinstance(path_or_buf, (str, bytes)) or hasattr(path_or_buf, "__fspath__")) and \
        all(str(t) == "float64" for t in df.dtypes) and \
        not any(ch in str(c) for c in df.columns for ch in ',"\n\r')
    if not simple:
        df.to_csv(path_or_buf, index=False)
        return
    lines = [",".join(str(c) for c in df.columns)]
    for row in df.to_numpy().tolist():
        lines.append(",".join("" if v != v else repr(v) for v in row))
    with open(path_or_buf, "w", newline="") as fh:
        fh.write("\n".join(lines) + "\n")


def _normalize_cn(atom_types, partial_relations, cn):
    return cn / [atom_types[a] for a in partial_relations[0]]  # rdf_cn.py:332-338


def _save_rdf(radii, relation_matrix, path_or_buf, save_mode, rdf_part_sum, rdf_full_sum=None):
    """Same columns and CSV behaviour as rdf_cn.py:341-365."""
    cols = [_R_LABEL] + (["g_full(r)"] if rdf_full_sum is not None else [])
    cols += [f"g_{pair[0]}-{pair[1]}" for pair in relation_matrix]
    blocks = (radii, rdf_full_sum, rdf_part_sum) if rdf_full_sum is not None else (radii, rdf_part_sum)
    final_df = pd.DataFrame(np.vstack(blocks).transpose(), columns=cols)
    if save_mode:
        _write_csv(final_df, path_or_buf)
        _say("Results are written to pd.DataFrame and csv file")
    else:
        _say(final_df)
    return final_df


def _save_cn(relation_matrix, path_or_buff, cn_sum, save_mode):
    cols = [f"cn_{pair[0]}-{pair[1]}" for pair in relation_matrix]
    final_df = pd.DataFrame(np.vstack(cn_sum).transpose(), columns=cols)
    if save_mode:
        _write_csv(final_df, path_or_buff)
        _say("CN results are written to pd.DataFrame and csv file")
    else:
        _say(final_df)
    return final_df


# ------------------------------------------------------------------------------------------------
# frame batching: SoA planes for the library
# ------------------------------------------------------------------------------------------------


class _Frame:
    """One parsed frame reduced to what the pair loops need (rdf_cn.py:183-194): id-sorted ids, types,
    xyz planes [3,N] and the box edge lengths."""

    __slots__ = ("timestep", "ids", "types", "xyz", "lengths")

    def __init__(self, timestep, ids, types, xyz, lengths):
        _say("The timestep of the current file is: " + str(timestep))
        self.timestep, self.ids, self.types, self.xyz, self.lengths = timestep, ids, types, xyz, lengths

    @classmethod
    def view(cls, fr):
        """A frame of a streamed batch (views into the staging buffer; no message: the stream printed it)."""
        self = cls.__new__(cls)
        self.timestep, self.ids, self.types, self.xyz, self.lengths = fr.timestep, fr.ids, fr.types, fr.xyz, fr.lengths
        return self

    @classmethod
    def from_dump(cls, dump):
        tbl = dump.data[["id", "type", "x", "y", "z"]].sort_values("id").to_numpy(dtype=np.float64)
        return cls(dump.timestep, tbl[:, 0], tbl[:, 1], np.ascontiguousarray(tbl[:, 2:5].T),
                   dump.box.to_lattice().lengths)


def _load_frames(filename, shard=False, stream=False):
    """Every frame of `filename` (file or '*' pattern, numeric order). The native reader of libmdhip.so
    produces the same doubles as the pandas-based one (tests/test_dump_reader_cpu.py), ~10x faster.

    stream=True (what the public functions ask for): a `stream.FrameStream` instead of a list — `_batches` then
    yields batches as the producer thread finishes parsing them, the frames of the trajectory are never all
    resident on the host (the reference builds the whole list first, rdf_cn.py:176).

    shard=True under torch.distributed (one process per GPU): a rank parses and returns only ITS share of the
    trajectory — a contiguous block of the files when there are at least as many files as ranks, else a
    contiguous block of the frames — so that parsing, the usual bottleneck, scales with the ranks too."""
    from .. import dist as D
    from .. import io as mio

    sharded = shard and D.is_distributed()
    files = None
    if sharded and (isinstance(filename, str) or hasattr(filename, "__fspath__")):
        matches = mio._sorted_matches(str(filename))
        if len(matches) >= D.rank_world()[1]:
            files = D.shard_items(matches)
    is_path = isinstance(filename, str) or hasattr(filename, "__fspath__")
    if stream and mio.USE_NATIVE_READER and is_path and (not sharded or files is not None):
        from ..stream import FrameStream

        return FrameStream(str(filename), files=files,
                           on_frame=lambda ts: _say("The timestep of the current file is: " + str(ts)))
    if mio.USE_NATIVE_READER and is_path:
        frames = [_Frame(ts, planes[0], planes[1], np.ascontiguousarray(planes[2:5]), lengths)
                  for ts, _b, lengths, _names, planes in
                  mio.iter_native_frames(str(filename), ["id", "type", "x", "y", "z"], sort_by="id", files=files)]
    elif files is not None:
        frames = [_Frame.from_dump(d) for fn in files for d in parse_lammps_dumps(fn)]
    else:
        frames = [_Frame.from_dump(d) for d in parse_lammps_dumps(filename)]
    if sharded and files is None:
        frames = D.shard_items(frames)
    return frames


def _all_frames(per_frame_rows):
    """Per-frame result rows of every rank in frame order (identity without torch.distributed). Ranks hold
    contiguous blocks of the trajectory, so the concatenation in rank order is the frame order, and summing
    the gathered rows in that order gives bit for bit what one process gets."""
    from .. import dist as D

    if len(per_frame_rows) and np.ndim(per_frame_rows[0]) == 2:  # per-batch blocks [B, W]
        rows = np.concatenate(per_frame_rows)
    else:
        rows = np.stack(per_frame_rows) if len(per_frame_rows) else None
    if not D.is_distributed():
        return [] if rows is None else rows
    # Fewer frames than ranks: the ranks without a frame contribute no rows (round 6; they used to make every rank raise).
    # The row width of an empty rank comes from the others, with the counts, in one small all-gather.
    mine = (0, 0) if rows is None else (int(rows.shape[0]), int(rows.shape[1]))
    both = D.allgather_var(np.array([mine], dtype=np.int64), counts=[1] * D.rank_world()[1])
    counts, width = [int(c) for c in both[:, 0]], int(both[:, 1].max())
    if sum(counts) == 0:
        return []
    if rows is None:
        rows = np.zeros((0, width))
    return D.allgather_var(rows, counts=counts)


def _is_writer():
    """Only rank 0 writes files under torch.distributed (every rank returns the same DataFrame)."""
    from .. import dist as D

    return D.rank_world()[0] == 0


class _Batch(list):
    """The frames of one library call; `block` = their coordinates as ONE array [B,3,N] when they already sit in a
    staging buffer (streamed batches), else None."""

    block = None
    uniform_types = False
    types_ref = None
    lengths_block = None


def _lengths_block(batch):
    lb = getattr(batch, "lengths_block", None)
    return lb if lb is not None else np.array([f.lengths for f in batch])


def _xyz_block(batch):
    return batch.block if getattr(batch, "block", None) is not None else np.stack([f.xyz for f in batch])


def _batches(frames):
    """Consecutive frames with the same atom count, capped at MAX_BATCH_BYTES of coordinates. A FrameStream
    yields its own batches (the staging buffer goes back to the producer when the loop asks for the next one:
    everything a caller keeps from a batch must be a copy)."""
    if not isinstance(frames, list):
        for sb in frames:
            b = _Batch(_Frame.view(fr) for fr in sb)
            b.block = sb.xyz
            b.uniform_types, b.types_ref = getattr(sb, "uniform_types", False), getattr(sb, "types_ref", None)
            b.lengths_block = np.asarray(sb.lengths, dtype=np.float64)
            yield b
        return
    start = 0
    while start < len(frames):
        n = frames[start].xyz.shape[1]
        cap = max(1, MAX_BATCH_BYTES // max(1, 24 * n))
        stop = start + 1
        while stop < len(frames) and stop - start < cap and frames[stop].xyz.shape[1] == n:
            stop += 1
        yield frames[start:stop]
        start = stop


def _labels_for(frames, labels_per_frame):
    """[N] when every frame carries the same labels, else [F, N]."""
    first = labels_per_frame[0]
    if all(np.array_equal(first, lab) for lab in labels_per_frame[1:]):
        return first.astype(np.int32)
    return np.stack(labels_per_frame).astype(np.int32)


# ------------------------------------------------------------------------------------------------
# public functions
# ------------------------------------------------------------------------------------------------


def calc_atomic_rdf(r_cut, bin_size, num_types, mass, partial_relations, filename, num_mols=None,
                    num_atoms_per_mol=None, path_or_buff="rdf.csv", save_mode=True):
    """
    Full and partial atom-atom g(r) averaged over the frames of `filename` (rdf_cn.py:385-530).

    Args follow the reference: r_cut (float), bin_size (float), num_types (int), mass (list of
    float, one per atom type), partial_relations ([[reference types], [other types]]), filename
    (dump file or '*' pattern), num_mols / num_atoms_per_mol (give both to use altered atom ids:
    the index of an atom inside its molecule type), path_or_buff, save_mode.
    Returns a DataFrame with columns r, g_full(r), g_a-b ...
    """
    dumps, num_bins, radii, num_files, num_relations = _initialize(r_cut, bin_size, filename, partial_relations)
    altered = bool(num_mols and num_atoms_per_mol)
    relation_matrix = np.asarray(partial_relations).transpose()
    rdf_full_sum = np.zeros(num_bins)
    rdf_part_sum = np.zeros((num_relations, num_bins))
    frames = dumps
    dropped = 0
    rows = []  # normalised g(r) of every frame this process holds: [g_full | g_part]
    for batch in _batches(frames):
        start = timer()
        lab_arg, props = _labels_and_props(batch, altered, num_mols, num_atoms_per_mol, num_types, mass,
                                           partial_relations)
        full, part, ov = backend.rdf_loop(_xyz_block(batch), lab_arg, _lengths_block(batch), relation_matrix, r_cut,
                                          bin_size, num_bins, per_frame=True)
        dropped += ov
        # every frame normalised with ITS box and densities (rdf_cn.py:502-513), all frames of the batch at once
        rows.append(_normalize_rdf_batch(bin_size, props, partial_relations, num_relations, num_bins, part, full,
                                         [f.xyz.shape[1] for f in batch]))
        if VERBOSE:
            for f in batch:
                _say("Finished computing RDF for timestep", f.timestep)
        _say("Trajectory loop took:", timer() - start, "s")
    rows = _all_frames(rows)  # every rank's frames, in frame order (identity in a single process)
    num_files = len(rows)
    if num_files:
        acc = _sum_frames(rows)
        rdf_full_sum, rdf_part_sum = acc[:num_bins], acc[num_bins:].reshape(num_relations, num_bins)
    if dropped:
        print(f"calc_atomic_rdf: {dropped} pair(s) fell in bin index {num_bins} (== num_bins) and were dropped")
    rdf_full_sum = rdf_full_sum / num_files
    rdf_part_sum = rdf_part_sum / num_files
    return _save_rdf(radii, relation_matrix, path_or_buff, save_mode and _is_writer(), rdf_part_sum,
                     rdf_full_sum=rdf_full_sum)


def calc_atomic_cn(r_cut, bin_size, num_types, mass, partial_relations, filename, num_mols=None,
                   num_atoms_per_mol=None, path_or_buff="cn.csv", save_mode=True):
    """
    Atom-atom coordination numbers, one cutoff per relation (rdf_cn.py:533-651). r_cut is a list.
    Returns a one-row DataFrame with columns cn_a-b.
    """
    dumps, _, _, num_files, num_relations = _initialize(r_cut, bin_size, filename, partial_relations)
    altered = bool(num_mols and num_atoms_per_mol)
    relation_matrix = np.asarray(partial_relations).transpose()
    cn_sum = np.zeros(num_relations)
    frames = dumps
    rows = []
    for batch in _batches(frames):
        lab_arg, props = _labels_and_props(batch, altered, num_mols, num_atoms_per_mol, num_types, mass,
                                           partial_relations)
        raw = backend.cn_loop(_xyz_block(batch), lab_arg, _lengths_block(batch), relation_matrix, list(r_cut),
                              per_frame=True)
        for k, f in enumerate(batch):
            rows.append(np.asarray(_normalize_cn(props[k][2], partial_relations, raw[k].astype(np.float64)),
                                   dtype=np.float64))
            _say("Finished computing CN for timestep", f.timestep)
    rows = _all_frames(rows)
    num_files = len(rows)
    for row in rows:
        cn_sum += row
    cn_sum = cn_sum / num_files
    return _save_cn(relation_matrix, path_or_buff, cn_sum, save_mode and _is_writer())


def calc_atomic_rdf_cn(r_cut, cn_r_cut, bin_size, num_types, mass, partial_relations, filename, num_mols=None,
                       num_atoms_per_mol=None, rdf_path_or_buff="rdf.csv", cn_path_or_buff="cn.csv", save_mode=True):
    """
    `calc_atomic_rdf(r_cut, ...)` and `calc_atomic_cn(cn_r_cut, ...)` (rdf_cn.py:385-651) of the same trajectory in
    ONE pass: the dump files are read once and every frame goes through one pair sweep that yields the histograms and
    the coordination counts (mdhip_rdf_cn_atomic). Returns (rdf DataFrame, cn DataFrame) — bit for bit the frames the
    two separate calls return, and the same two CSV files. (Not in the reference, which runs the two functions one
    after the other over the same pairs: BASELINE config 3 asks for both.)
    """
    dumps, num_bins, radii, num_files, num_relations = _initialize(r_cut, bin_size, filename, partial_relations)
    if len(cn_r_cut) != num_relations:
        raise ValueError("one coordination cutoff per relation is required")
    altered = bool(num_mols and num_atoms_per_mol)
    relation_matrix = np.asarray(partial_relations).transpose()
    dropped = 0
    rows, cn_rows = [], []
    for batch in _batches(dumps):
        lab_arg, props = _labels_and_props(batch, altered, num_mols, num_atoms_per_mol, num_types, mass,
                                           partial_relations)
        full, part, ov, raw = backend.rdf_cn_loop(_xyz_block(batch), lab_arg, _lengths_block(batch), relation_matrix,
                                                  r_cut, bin_size, num_bins, list(cn_r_cut), per_frame=True)
        dropped += ov
        rows.append(_normalize_rdf_batch(bin_size, props, partial_relations, num_relations, num_bins, part, full,
                                         [f.xyz.shape[1] for f in batch]))
        for k, f in enumerate(batch):
            cn_rows.append(np.asarray(_normalize_cn(props[k][2], partial_relations, raw[k].astype(np.float64)),
                                      dtype=np.float64))
            _say("Finished computing RDF and CN for timestep", f.timestep)
    rows, cn_rows = _all_frames(rows), _all_frames(cn_rows)
    n_frames = len(rows)
    acc = _sum_frames(rows) if n_frames else np.zeros((1 + num_relations) * num_bins)
    rdf_full_sum, rdf_part_sum = acc[:num_bins], acc[num_bins:].reshape(num_relations, num_bins)
    cn_sum = _sum_frames(cn_rows) if n_frames else np.zeros(num_relations)
    if dropped:
        print(f"calc_atomic_rdf_cn: {dropped} pair(s) fell in bin index {num_bins} (== num_bins) and were dropped")
    g = _save_rdf(radii, relation_matrix, rdf_path_or_buff, save_mode and _is_writer(), rdf_part_sum / n_frames,
                  rdf_full_sum=rdf_full_sum / n_frames)
    c = _save_cn(relation_matrix, cn_path_or_buff, cn_sum / n_frames, save_mode and _is_writer())
    return g, c


def _same_types(batch):
    return all(np.array_equal(batch[0].types, f.types) for f in batch[1:])


def _molecular_inputs(batch, num_mols, num_atoms_per_mol, mass):
    """Device-side COM of wrapped coordinates for every frame of the batch (rdf_cn.py:218-241)."""
    seg_off, seg_type = _molecule_layout(num_mols, num_atoms_per_mol)
    n = batch[0].xyz.shape[1]
    if seg_off[-1] != n:
        raise ValueError(f"Length of values ({int(seg_off[-1])}) does not match length of index ({n})")
    xyz = _xyz_block(batch)
    atom_mass = np.asarray(mass, dtype=np.float64)[batch[0].types.astype(np.int64) - 1]
    if _same_types(batch):
        sites, _, _ = backend.segment_com(xyz, atom_mass, seg_off)
    else:  # per-frame masses: one call per frame
        sites = np.concatenate([
            backend.segment_com(f.xyz[None], np.asarray(mass, dtype=np.float64)[f.types.astype(np.int64) - 1],
                                seg_off)[0] for f in batch])
    return xyz, sites, seg_type


def calc_molecular_rdf(r_cut, bin_size, num_types, mass, partial_relations, filename, num_mols,
                       num_atoms_per_mol, path_or_buff="rdf_mol.csv", save_mode=True):
    """
    Partial g(r) between atoms (first list of partial_relations) and molecule centres of mass
    (second list: molecule type numbers) (rdf_cn.py:654-756).
    """
    dumps, num_bins, radii, num_files, num_relations = _initialize(r_cut, bin_size, filename, partial_relations)
    relation_matrix = np.asarray(partial_relations).transpose()
    rdf_part_sum = np.zeros((num_relations, num_bins))
    frames = dumps
    dropped = 0
    rows = []
    for batch in _batches(frames):
        xyz, sites, seg_type = _molecular_inputs(batch, num_mols, num_atoms_per_mol, mass)
        props = [_calc_props(f.lengths, f.types, seg_type, num_types, mass, partial_relations, False)
                 for f in batch]
        if _same_types(batch):
            part, ov = backend.rdf_mol_loop(xyz, batch[0].types.astype(np.int32), sites, seg_type,
                                            np.array([f.lengths for f in batch]), relation_matrix, r_cut,
                                            bin_size, num_bins, per_frame=True)
        else:
            part, ov = _per_frame_mol_rdf(batch, sites, seg_type, relation_matrix, r_cut, bin_size, num_bins)
        dropped += ov
        for k, f in enumerate(batch):
            _, rho_pairs, atom_types, _ = props[k]
            _, g_part = _normalize_rdf(bin_size, rho_pairs, atom_types, partial_relations, num_relations,
                                       num_bins, part[k].astype(np.float64))
            rows.append(np.ravel(g_part))
            _say("Finished computing RDF for timestep", f.timestep)
    rows = _all_frames(rows)
    num_files = len(rows)
    for row in rows:
        rdf_part_sum += row.reshape(num_relations, num_bins)
    if dropped:
        print(f"calc_molecular_rdf: {dropped} pair(s) fell in bin index {num_bins} (== num_bins) and were dropped")
    rdf_part_sum = rdf_part_sum / num_files
    return _save_rdf(radii, relation_matrix, path_or_buff, save_mode and _is_writer(), rdf_part_sum)


def _per_frame_mol_rdf(batch, sites, seg_type, relation_matrix, r_cut, bin_size, num_bins):
    parts, ov = [], 0
    for k, f in enumerate(batch):
        p, o = backend.rdf_mol_loop(f.xyz[None], f.types.astype(np.int32), sites[k:k + 1], seg_type,
                                    np.array([f.lengths]), relation_matrix, r_cut, bin_size, num_bins)
        parts.append(p[0])
        ov += o
    return np.stack(parts), ov


def calc_molecular_cn(r_cut, bin_size, num_types, mass, partial_relations, filename, num_mols,
                      num_atoms_per_mol, path_or_buff="cn_mol.csv", save_mode=True):
    """Atom - molecule-COM coordination numbers, one cutoff per relation (rdf_cn.py:759-855)."""
    dumps, _, _, num_files, num_relations = _initialize(r_cut, bin_size, filename, partial_relations)
    relation_matrix = np.asarray(partial_relations).transpose()
    cn_sum = np.zeros(num_relations)
    frames = dumps
    rows = []
    for batch in _batches(frames):
        xyz, sites, seg_type = _molecular_inputs(batch, num_mols, num_atoms_per_mol, mass)
        props = [_calc_props(f.lengths, f.types, seg_type, num_types, mass, partial_relations, False)
                 for f in batch]
        if _same_types(batch):
            raw = backend.cn_mol_loop(xyz, batch[0].types.astype(np.int32), sites, seg_type,
                                      np.array([f.lengths for f in batch]), relation_matrix, list(r_cut))
        else:
            raw = np.stack([backend.cn_mol_loop(f.xyz[None], f.types.astype(np.int32), sites[k:k + 1], seg_type,
                                                np.array([f.lengths]), relation_matrix, list(r_cut))[0]
                            for k, f in enumerate(batch)])
        for k, f in enumerate(batch):
            rows.append(np.asarray(_normalize_cn(props[k][2], partial_relations, raw[k].astype(np.float64)),
                                   dtype=np.float64))
            _say("Finished computing CN for timestep", f.timestep)
    rows = _all_frames(rows)
    num_files = len(rows)
    for row in rows:
        cn_sum += row
    cn_sum = cn_sum / num_files
    return _save_cn(relation_matrix, path_or_buff, cn_sum, save_mode and _is_writer())


def calc_intermolecular_rdf(r_cut, bin_size, num_types, mass, partial_relations, filename, num_mols,
                            num_atoms_per_mol, path_or_buff="rdf_mol.csv", save_mode=True):
    """
    Molecule-COM to molecule-COM partial g(r) (rdf_cn.py:857-903; undocumented upstream, a molecule is
    paired with itself as there). partial_relations holds molecule type numbers on both sides.
    """
    dumps, num_bins, radii, num_files, num_relations = _initialize(r_cut, bin_size, filename, partial_relations)
    relation_matrix = np.asarray(partial_relations).transpose()
    rdf_part_sum = np.zeros((num_relations, num_bins))
    frames = dumps
    rows = []
    for batch in _batches(frames):
        _, sites, seg_type = _molecular_inputs(batch, num_mols, num_atoms_per_mol, mass)
        props = [_calc_props(f.lengths, seg_type, seg_type, num_types, mass, partial_relations, False)
                 for f in batch]
        part, _ = backend.rdf_mol_loop(sites, seg_type, sites, seg_type, np.array([f.lengths for f in batch]),
                                       relation_matrix, r_cut, bin_size, num_bins, per_frame=True)
        for k in range(len(batch)):
            _, rho_pairs, atom_types, _ = props[k]
            _, g_part = _normalize_rdf(bin_size, rho_pairs, atom_types, partial_relations, num_relations,
                                       num_bins, part[k].astype(np.float64))
            rows.append(np.ravel(g_part))
    rows = _all_frames(rows)
    num_files = len(rows)
    for row in rows:
        rdf_part_sum += row.reshape(num_relations, num_bins)
    rdf_part_sum = rdf_part_sum / num_files
    return _save_rdf(radii, relation_matrix, path_or_buff, save_mode and _is_writer(), rdf_part_sum)
