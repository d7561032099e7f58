"""
Seeded synthetic inputs for the BASELINE.json configurations (numpy PCG64, float64).

C2: N=10 000, F=200, cubic L=50 A, uniform positions per frame, 4 types (1 + (id-1) mod 4),
    all 10 unordered type pairs as relations, r_cut 20, bin 0.05 (400 bins).
C3: N=100 000, F=1000, L=104 A, as C2 plus CN with per-relation cutoffs 2.325 + 0.5 kl.
C4: N=50 000, F=5000 unwrapped random walk, sigma 0.1 A/frame, r(0) uniform in L=82.8 A.
C5: n=1e6 samples, three AR(1) (phi 0.99) series x100.
Frames are generated independently from (seed, frame index) so that any rank can make its own
shard without generating the others.
"""

import numpy as np

BASE_SEED = 20250328

ALL_PAIRS_4 = [(a, b) for a in range(1, 5) for b in range(a, 5)]  # 10 unordered pairs of 4 types


def rdf_types(n_atoms, n_types=4):
    return (1 + (np.arange(n_atoms) % n_types)).astype(np.int32)


# The reference's own example (data/mg_tfsi_dme): 591 DME x 16 atoms, 66 TFSI x 15 atoms, 33 Mg in a cubic box.
C1_ATOMS, C1_BOX = 10_479, 49.182348836183905
C1_TYPE_COUNTS = {1: 1182, 2: 2364, 3: 5910, 4: 66, 5: 132, 6: 264, 7: 132, 8: 396, 9: 33}
C1_RELATIONS = [(9, 1), (9, 4), (9, 6), (9, 9), (1, 3)]  # the relations of the example notebook
C1_MOLS, C1_ATOMS_PER_MOL = [591, 66, 33], [16, 15, 1]
C1_ALT_RELATIONS = [(32, 17), (32, 32)]                   # altered ids (rdf_cn.py:197-215): Mg - first TFSI atom, Mg - Mg


def c1_types(alt=False):
    """Type column of a C1-shaped frame: the nine atom types with the example's populations, or (alt) the 32 pseudo-types
    of the altered-id mode — the index of an atom inside its molecule type, offset by the preceding types' atoms."""
    if not alt:
        return np.concatenate([np.full(c, t, np.int32) for t, c in C1_TYPE_COUNTS.items()])
    out, first = [], 1
    for mols, per in zip(C1_MOLS, C1_ATOMS_PER_MOL):
        out.append(np.tile(np.arange(first, first + per, dtype=np.int32), mols))
        first += per
    return np.concatenate(out)


def rdf_frames(n_atoms, frame_ids, box_len, seed_offset=2, dtype=np.float64):
    """Ideal-gas frames [len(frame_ids), 3, n_atoms] in [0, L)."""
    out = np.empty((len(frame_ids), 3, n_atoms), dtype=dtype)
    for k, f in enumerate(frame_ids):
        rng = np.random.default_rng([BASE_SEED + seed_offset, int(f)])
        out[k] = rng.random((3, n_atoms)) * box_len
    return out


def rdf_config(name):
    if name == "C2":
        return dict(n_atoms=10_000, n_frames=200, box_len=50.0, r_cut=20.0, bin_size=0.05, seed_offset=2)
    if name == "C1":  # synthetic positions at the example's size and density (the real frames: tests/golden)
        return dict(n_atoms=C1_ATOMS, n_frames=200, box_len=C1_BOX, r_cut=20.0, bin_size=0.05, seed_offset=1)
    if name == "C3":
        return dict(n_atoms=100_000, n_frames=1000, box_len=104.0, r_cut=20.0, bin_size=0.05, seed_offset=3)
    raise KeyError(name)


def cn_cutoffs(n_rel):
    return [2.325 + 0.5 * kl for kl in range(n_rel)]


def random_walk(n_ent, n_frames, box_len=82.8, sigma=0.1, seed_offset=4, chunk=250):
    """Unwrapped random walk [n_frames, 3, n_ent] (C4)."""
    rng = np.random.default_rng(BASE_SEED + seed_offset)
    out = np.empty((n_frames, 3, n_ent))
    out[0] = rng.random((3, n_ent)) * box_len
    for f0 in range(1, n_frames, chunk):
        f1 = min(n_frames, f0 + chunk)
        steps = rng.normal(0.0, sigma, size=(f1 - f0, 3, n_ent))
        np.cumsum(steps, axis=0, out=steps)
        out[f0:f1] = out[f0 - 1] + steps
    return out


def ar1_series(n, n_series=3, phi=0.99, scale=100.0, seed_offset=5):
    """AR(1) pressure-like series [n_series, n] (C5)."""
    from scipy.signal import lfilter

    rng = np.random.default_rng(BASE_SEED + seed_offset)
    e = rng.standard_normal((n_series, n))
    return lfilter([1.0], [1.0, -phi], e, axis=1) * scale


def residence_walk(n_frames=1000, n_central=315, n_shell=11_280, box_len=104.0, sigma=0.1, seed_offset=7):
    """The residence leg's trajectory (bench.py, tools/run_secondary.py): the example's shares of central (Mg) and shell
    (ether O) atoms of 100 000 atoms at C3's density, unwrapped random walks; -> (central [F,3,n_c], shell [F,3,n_s])."""
    rng = np.random.default_rng(BASE_SEED + seed_offset)
    n = n_central + n_shell
    r = np.empty((n_frames, 3, n))
    r[0] = rng.random((3, n)) * box_len
    for f0 in range(1, n_frames, 100):
        steps = rng.normal(0.0, sigma, (min(n_frames, f0 + 100) - f0, 3, n))
        np.cumsum(steps, axis=0, out=steps)
        r[f0:f0 + len(steps)] = r[f0 - 1] + steps
    return np.ascontiguousarray(r[:, :, :n_central]), np.ascontiguousarray(r[:, :, n_central:])
