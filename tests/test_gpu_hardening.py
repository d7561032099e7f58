"""
Round-2 hardening of the pin and of the boundary (GPU side):
  * the ctypes stub of INTEGRATION.md section 2 is executed VERBATIM — in a fresh interpreter that imports neither this
    package nor torch — and its `_rdf_loop` must reproduce the reference's integers on the mg_tfsi_dme golden frame;
  * more than 255 histogram classes (ids no longer fit a byte anywhere on the host);
  * the packed-f32 sweep on adversarial inputs at C2 scale: a lattice whose distances sit exactly ON bin edges and on
    the cutoff, all atoms coincident (every pair in bin 0: the fullest LDS words a frame can produce), and a slice
    of the soak generator (tests/bench/soak_pk.py) inside the suite.
"""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from oracle import cref as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from mdproptools_amd import backend

    return backend


def test_integration_md_stub_verbatim(tmp_path):
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    (stub,) = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert "mdhip_rdf_atomic" in stub and "def _rdf_loop" in stub and "mdproptools_amd" not in stub
    driver = '''
import sys, json, numpy as np
STUB = open(sys.argv[1]).read()
exec(compile(STUB, "INTEGRATION.md#2", "exec"))
assert "torch" not in sys.modules and "mdproptools_amd" not in sys.modules
g = np.load(sys.argv[2])
fr = g["frames"][0]
fr = fr[np.argsort(fr[:, 0], kind="stable")]                       # rdf_cn.py:192 sort_values("id")
data = np.ascontiguousarray(fr[:, 1:5])                             # [type, x, y, z] as _rdf_loop receives it
lengths = g["bounds"][0][:, 1] - g["bounds"][0][:, 0]
rel = np.ascontiguousarray(g["rdf_def_rel"].T)                      # relation_matrix rows (a, b)
rdf_full = np.zeros(400); rdf_part = np.zeros((len(rel), 400))
_rdf_loop(data, rel, len(rel), lengths, 20.0, 0.05, rdf_full, rdf_part)
ok = bool(np.array_equal(rdf_full, g["rdf_def_full"][0]) and np.array_equal(rdf_part, g["rdf_def_part"][0]))
print(json.dumps({"ok": ok, "sum": int(rdf_full.sum())}))
'''
    (tmp_path / "stub.py").write_text(stub)
    (tmp_path / "driver.py").write_text(driver)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(REPO, "mdproptools_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(tmp_path / "driver.py"), str(tmp_path / "stub.py"),
                        os.path.join(GOLDEN, "c1_rdf.npz")], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res == {"ok": True, "sum": 30926986}  # SURVEY.md's known answer for frame 0


def test_more_than_255_classes(B):
    """24 atom types, all 300 unordered type pairs as relations (301 classes with 'other'): several class passes,
    class ids beyond a byte. Against the C oracle, per relation."""
    rng = np.random.default_rng(255)
    n, L, T = 2600, 30.0, 24
    xyz = rng.uniform(0, L, (2, 3, n))
    ty = rng.integers(1, T + 1, n).astype(np.int32)
    rel = np.array([[a, b] for a in range(1, T + 1) for b in range(a, T + 1)], dtype=np.int32)
    assert len(rel) == 300
    box = np.full((2, 3), L)
    full, part, ov = B.rdf_loop(xyz, ty, box, rel, 9.0, 0.1, 90)
    cuts = list(2.0 + 6.5 * rng.random(len(rel)))
    cn = B.cn_loop(xyz, ty, box, rel, cuts)
    for f in range(2):
        cf, cp, cov = C.rdf_pairs(xyz[f], ty, rel, box[f], 81.0, 0.1, 90)
        np.testing.assert_array_equal(full[f], cf)
        np.testing.assert_array_equal(part[f], cp)
        np.testing.assert_array_equal(cn[f], C.cn_pairs(xyz[f], ty, rel, box[f], [c * c for c in cuts]))
    # a frame large enough for the culled sweep (>= 8 tiles) with the same 300 relations
    n2 = 4200
    x2 = rng.uniform(0, 40.0, (1, 3, n2))
    t2 = rng.integers(1, T + 1, n2).astype(np.int32)
    f2, p2, _ = B.rdf_loop(x2, t2, np.full((1, 3), 40.0), rel, 9.0, 0.1, 90)
    cf, cp, _ = C.rdf_pairs(x2[0], t2, rel, [40.0] * 3, 81.0, 0.1, 90)
    np.testing.assert_array_equal(f2[0], cf)
    np.testing.assert_array_equal(p2[0], cp)


def test_lattice_on_bin_edges_at_c2_scale(B):
    """10 000 atoms on a simple cubic lattice whose spacing is a whole number of bins (2.5 A = 50 bins of 0.05): a
    large share of all distances sits exactly ON a bin edge, and lattice vectors such as (8,0,0) a = 20 A sit exactly
    on the cutoff. C2's cutoff, bin size and type pattern; packed-f32 sweep == all-f64 sweep == C oracle (frame 0)."""
    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context

    g, a = 22, 2.5
    L = g * a  # 55 A
    rng = np.random.default_rng(20)
    frames = []
    for _ in range(3):
        idx = rng.choice(g ** 3, 10_000, replace=False)
        cell = np.stack([idx % g, (idx // g) % g, idx // (g * g)]).astype(np.float64)
        frames.append(cell * a)
    xyz = np.stack(frames)
    ty = synth.rdf_types(10_000)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((3, 3), L)
    pk, f64 = Context(0), Context(0)
    f64.set_option("rdf_pk", 0)
    a_ = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, ctx=pk)
    assert "<3," in pk.last_kernel_name(), pk.last_kernel_name()
    b_ = B.rdf_loop(xyz, ty, box, rel, 20.0, 0.05, 400, ctx=f64)
    assert "<2," in f64.last_kernel_name()
    np.testing.assert_array_equal(a_[0], b_[0])
    np.testing.assert_array_equal(a_[1], b_[1])
    cf, cp, cov = C.rdf_pairs_threaded(xyz[0], ty, rel, box[0], 400.0, 0.05, 400, 16)
    np.testing.assert_array_equal(a_[0][0], cf)
    np.testing.assert_array_equal(a_[1][0], cp)
    # nearly every populated bin is a multiple of the lattice's sqrt(k) pattern: the edge-sitting pairs are many
    on_edge = sum(int(cf[int(round(np.sqrt(k) * a / 0.05))]) for k in (1, 4, 9, 16, 25, 36, 49)
                  if abs(np.sqrt(k) * a / 0.05 - round(np.sqrt(k) * a / 0.05)) < 1e-9)
    assert on_edge > 100_000
    # the coordination counts from the same sweep, cutoffs ON lattice distances (2.5, 5.0 A ...) and between them
    cuts = [2.5, 5.0, 7.5, 3.0, 3.5355339059327378, 4.330127018922194, 6.0, 2.5, 10.0, 12.5]
    f_, p_, ov, cn = B.rdf_cn_loop(xyz, ty, box, rel, 20.0, 0.05, 400, cuts, ctx=pk)
    np.testing.assert_array_equal(f_, a_[0])
    np.testing.assert_array_equal(cn[0], C.cn_pairs(xyz[0], ty, rel, box[0], [c * c for c in cuts]))
    pk.set_option("cn_pk", 1)  # mdhip_cn_atomic through the packed sweep (coarse histogram + split bins)
    np.testing.assert_array_equal(cn, B.cn_loop(xyz, ty, box, rel, cuts, ctx=f64))
    np.testing.assert_array_equal(cn, B.cn_loop(xyz, ty, box, rel, cuts, ctx=pk))  # coarse histogram + split bins
    pk.close()
    f64.close()


def test_all_atoms_coincident_many_frames(B):
    """Degenerate input: every atom of every frame at one point, so that every pair of a frame lands in ONE histogram
    word (bin 0). 4096 atoms x 600 frames = 5.0e9 increments of a single counter overall — more than 2^32: the
    per-block 32-bit LDS words must be flushed before they wrap (the library bounds the frames a block may sum)."""
    import torch

    n, F = 4096, 600
    xyz = torch.full((F, 3, n), 7.25, dtype=torch.float64, device="cuda")
    ty = np.ones(n, dtype=np.int32)
    rel = np.array([[1, 1]])
    box = np.full((F, 3), 30.0)
    pairs = n * (n - 1) // 2
    full, part, ov = B.rdf_loop(xyz, ty, box, rel, 10.0, 0.05, 200, per_frame=False)
    assert int(full[0]) == 2 * pairs * F and int(full[1:].sum()) == 0 and ov == 0
    assert int(part[0, 0]) == 2 * pairs * F
    assert 2 * pairs * F > 2 ** 32
    pf, pp, _ = B.rdf_loop(xyz[:3], ty, box[:3], rel, 10.0, 0.05, 200, per_frame=True)
    assert [int(v) for v in pf[:, 0]] == [2 * pairs] * 3
    cn = B.cn_loop(xyz, ty, box, rel, [1.0], per_frame=False)
    assert int(cn[0]) == 2 * pairs * F


def test_soak_slice_packed_vs_f64_and_oracle(B):
    """300 cases of the soak generator (tests/bench/soak_pk.py, the first of its seeded stream) inside the suite:
    packed-f32 sweep == all-f64 sweep everywhere, and every 10th case also == the C oracle on frame 0."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_gpu_parity import _pk_case
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(20250328)
    f64, pk = Context(0), Context(0)
    for ctx, v in ((f64, 0), (pk, 1)):
        ctx.set_option("rdf_cull", 1)
        ctx.set_option("rdf_pk", v)
    engaged = 0
    for trial in range(300):
        xyz, ty, box, rel, r_cut, bin_size, nbins = _pk_case(rng, trial)
        xyz = xyz[:2]
        box = box[:2]
        per_frame = bool(trial % 2) or trial % 10 == 0
        a = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=f64)
        b = B.rdf_loop(xyz, ty, box, rel, r_cut, bin_size, nbins, per_frame=per_frame, ctx=pk)
        engaged += any(t in pk.last_kernel_name() for t in ("<3,", "<4,", "<5,", "<6,"))
        msg = "trial %d kernel %s" % (trial, pk.last_kernel_name())
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], msg
        if trial % 10 == 0:
            cf, cp, _ = C.rdf_pairs(xyz[0], ty, rel, box[0], r_cut * r_cut, bin_size, nbins)
            assert np.array_equal(b[0][0], cf) and np.array_equal(b[1][0], cp), msg
    assert engaged >= 200
    f64.close()
    pk.close()


def test_overflow_guard_splits_the_batch(B):
    """The guard of the 32-bit LDS words: with its threshold lowered (rdf_guard) a persistent block that swept more
    neighbour tiles than allowed raises the flag, the host halves the batch until every launch passes, and the sums
    are what the unguarded run gives — frame-summed, per frame, with device-resident sums and with CN riding along."""
    import torch

    from mdproptools_amd import synth
    from mdproptools_amd._lib import Context

    n, L, F = 6000, 42.0, 96  # (many frames: a block's tile count at 96 frames is far above its count at one frame,
    #                            so that there is a wide range of thresholds that split the batch and still pass)
    xyz = synth.rdf_frames(n, range(F), L, 11)
    ty = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4)
    box = np.full((F, 3), L)
    ref = B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 240, per_frame=False)
    refp = B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 240, per_frame=True)
    ctx = Context(0)
    # lower the number of neighbour tiles a block may sweep per launch until the 96-frame batch has to be halved
    # (a threshold even one frame exceeds gives a clean error instead, see the end of the test)
    split_at = None
    for guard in (2048, 1024, 768, 512, 384, 256, 192, 128, 96, 64, 48, 32, 24, 16, 12, 8, 6, 4):
        ctx.set_option("rdf_guard", guard)
        try:
            a = B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 240, per_frame=False, ctx=ctx)
        except Exception as e:
            assert "overflow the 32-bit" in str(e)
            break
        np.testing.assert_array_equal(a[0], ref[0])
        np.testing.assert_array_equal(a[1], ref[1])
        if ctx.last_kernel_ms()[1] > 1 and split_at is None:
            # the LARGEST threshold that splits: the blocks' tile counts vary from run to run with the work queue, so
            # the calls below need the margin the smaller thresholds do not have
            split_at = guard
    assert split_at is not None, "no threshold made the host split the batch"
    # per-frame output: a frame's blocks share its items, so a block sweeps more tiles than in the persistent grid —
    # the first threshold that passes must still give the right rows
    for k in (1, 2, 4, 8, 16, 64):
        ctx.set_option("rdf_guard", split_at * k)
        try:
            b = B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 240, per_frame=True, ctx=ctx)
        except Exception as e:
            assert "overflow the 32-bit" in str(e)
            continue
        np.testing.assert_array_equal(b[0], refp[0])
        break
    else:
        raise AssertionError("per-frame output never passed the guard")
    ctx.set_option("rdf_guard", split_at)
    out = torch.empty(11 * 240 + 1, dtype=torch.int64, device="cuda")
    B.rdf_loop_dev(torch.from_numpy(xyz).cuda(), ty, box, rel, 12.0, 0.05, 240, out, ctx=ctx)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64)[:240], ref[0])
    cuts = synth.cn_cutoffs(len(rel))
    f_, p_, ov, cn = B.rdf_cn_loop(xyz, ty, box, rel, 12.0, 0.05, 240, cuts, per_frame=False, ctx=ctx)
    np.testing.assert_array_equal(f_, ref[0])
    np.testing.assert_array_equal(cn, B.cn_loop(xyz, ty, box, rel, cuts, per_frame=False))
    ctx.set_option("rdf_guard", 1)  # not even one frame passes: a clean error, not wrong sums
    with pytest.raises(Exception, match="overflow the 32-bit"):
        B.rdf_loop(xyz, ty, box, rel, 12.0, 0.05, 240, per_frame=False, ctx=ctx)
    ctx.close()


def test_soak_slices_inside_the_suite():
    """A slice of the long differential runs (tests/bench/soak_pk.py, soak_cn.py) inside `-m gpu`, behind a time budget of
    about a minute: 700 cases of the packed-f32 sweep against the all-f64 sweep (fresh seed: not the stream of
    test_soak_slice_packed_vs_f64_and_oracle) and 400 cases of the one-sweep RDF + CN call against the two calls, every
    fifth one also against oracle/cpu_ref.c. The full soaks (12 000 + 3000 cases) are run by hand per round and
    logged under profiles/."""
    import subprocess
    import time

    t0 = time.perf_counter()
    for script, args in (("soak_pk.py", ["700", "909"]), ("soak_cn.py", ["400", "17", "oracle"])):
        r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench", script)] + args, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        tail = "\n".join(r.stdout.splitlines()[-5:])
        assert r.returncode == 0 and "identical" in tail, tail
    assert time.perf_counter() - t0 < 300.0


def test_soak_slice_full_lag_sources():
    """A slice of tests/bench/soak_lag.py inside `-m gpu`: ten random shapes (frames 2049 .. 8192, entities 1 .. 1500, odd
    and even column counts, up to six groups with empty ones) through the transposed-copy, read-in-place and in-kernel
    transposition forms of the fused full-lag MSD kernel — agreement within the reported bounds, every call reproducible
    bit for bit (the in-kernel form hands tiles from block to block)."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench", "soak_lag.py"), "10", "77"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-5:])
    assert r.returncode == 0 and "agree within their bounds" in tail, tail


def test_soak_slice_long_trajectories():
    """A slice of tests/bench/soak_lag_long.py inside `-m gpu` (round 6): six random shapes of 8193 .. 26 000 frames through
    the residue-class kernels (padded length 24 576 / 49 152, both forms) against the batched transforms, batches of a few
    MB so that batches, blocks and segments straddle — agreement within the reported bounds, reproducible bit for bit."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "bench", "soak_lag_long.py"), "6", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-5:])
    assert r.returncode == 0 and "agree with the batched transforms" in tail, tail


def test_every_pair_ambiguous_fills_the_queues(B):
    """The deferred-pair queues at their limit (the push has no capacity test). Atoms sit on TWO points a
    whole number of bins apart: every pair is either at distance 0 or exactly on a bin edge, i.e. inside the guard band of
    the packed-f32 guess — every lane of every group pushes, 4 x 64 entries on top of whatever a wave still holds. If an
    entry ever landed beyond a wave's 320 words it would overwrite the next wave's queue or the CN tables and these counts
    could not all be right: histogram, coordination counts (cutoffs on and off the distance) and the one-sweep call
    against the C oracle, packed sweep against the all-f64 sweep."""
    n, L = 8192, 30.0
    rng = np.random.default_rng(8)
    side = rng.integers(0, 2, n)
    xyz = np.empty((2, 3, n))
    for f, d in enumerate((4.0, 6.35)):  # 80 and 127 bins of 0.05
        xyz[f, 0] = 10.0 + d * side
        xyz[f, 1] = 11.0
        xyz[f, 2] = 12.0
    ty = (1 + np.arange(n) % 3).astype(np.int32)
    rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3]])
    box = np.full((2, 3), L)
    ctx = B.default_context()
    full, part, ov = B.rdf_loop(xyz, ty, box, rel, 10.0, 0.05, 200, ctx=ctx)
    assert "<3" in ctx.last_kernel_name() or "<4" in ctx.last_kernel_name()  # the packed sweep took it
    cuts = [4.0, 6.35, 3.9999, 9.0]
    cn = B.cn_loop(xyz, ty, box, rel, cuts, ctx=ctx)
    f2, p2, o2, cn2 = B.rdf_cn_loop(xyz, ty, box, rel, 10.0, 0.05, 200, cuts, ctx=ctx)
    ctx.set_option("rdf_pk", 0)
    try:
        f64 = B.rdf_loop(xyz, ty, box, rel, 10.0, 0.05, 200, ctx=ctx)
    finally:
        ctx.set_option("rdf_pk", -1)
    for f in range(2):
        cf, cp, cov = C.rdf_pairs(xyz[f], ty, rel, box[f], 100.0, 0.05, 200)
        np.testing.assert_array_equal(full[f], cf)
        np.testing.assert_array_equal(part[f], cp)
        np.testing.assert_array_equal(cn[f], C.cn_pairs(xyz[f], ty, rel, box[f], [c * c for c in cuts]))
    assert int(full.sum()) == 2 * 2 * (n * (n - 1) // 2) and ov == 0
    np.testing.assert_array_equal(f2, full)
    np.testing.assert_array_equal(p2, part)
    np.testing.assert_array_equal(cn2, cn)
    np.testing.assert_array_equal(f64[0], full)
    np.testing.assert_array_equal(f64[1], part)


def test_queue_limit_cases_through_the_capacity_check_build():
    """the packed sweep's pushes carry no capacity test in the shipped build (+1-2 % for
    a check that cannot fire); the invariant `a wave never holds more than PK_QCAP entries` is checked by the DEBUG build
    (-DPK_CAPCHECK, mdproptools_amd/build.py:build_capcheck — entries beyond the queue are counted and the call fails).
    Through that build, in a fresh interpreter: the every-pair-ambiguous case above (queues at their limit) and a slice of
    the randomised packed-vs-f64 soak. Both must pass with the check armed."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "_bin", "libmdhip_capcheck.so")
    if not os.path.exists(lib):
        from mdproptools_amd import build as bld

        lib = bld.build_capcheck()
    env = dict(os.environ, MDHIP_LIB=lib)
    chk = ("import ctypes, os; from mdproptools_amd import _lib; h = _lib.load(); "
           "assert os.path.realpath(h._name) == os.path.realpath(os.environ['MDHIP_LIB']), h._name")
    r = subprocess.run([sys.executable, "-c", chk], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_hardening.py::test_every_pair_ambiguous_fills_the_queues"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "bench", "soak_pk.py"), "60", "17"], cwd=root, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_lag_msd_staged_integer_ramp_exact_and_reproducible():
    """The in-kernel transposition of the round-5 full-lag kernel (msd_fft_w12.h: clusters of workgroups stage their tiles
    through a ring in device memory) on an input where every wrong sample shows: x[t][column] = 16384 column + t — integers,
    so every series' mean and every centred sample is exact, MSD(k) = k^2 for every series, and a sample taken from another
    row or column moves the result by orders of magnitude more than rounding. (A pipelined form of the staging prologue
    delivered wrong rows on this GPU: found this way.) Against the transposed-copy source, against k^2, and bit-identical
    from call to call."""
    import torch

    from mdproptools_amd import backend as B
    from mdproptools_amd._lib import Context

    ctx = Context(0)
    try:
        # (2500, 3071: the SHORT instance of round 6, 1536 <= F < 3072 with F + max_lag in (4096, 8192])
        for F, E in ((5000, 4096), (6144, 1024), (4097, 2048), (2500, 2048), (3071, 1024)):
            t = torch.arange(F, dtype=torch.float64, device="cuda")[:, None, None]
            c = torch.arange(3 * E, dtype=torch.float64, device="cuda").reshape(1, 3, E)
            r = (16384.0 * c + t).contiguous()
            ctx.set_option("lag_variant", 2)
            ctx.set_option("lag_direct", 0)
            ref = B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx)
            assert ctx.last_kernel_name() == "msd_power_w12_kernel"
            ctx.set_option("lag_direct", 2)
            outs = [B.lag_msd(r, F - 1, [0, E], scale=1.0, ctx=ctx) for _ in range(4)]
            assert ctx.last_kernel_name() == "msd_power_w12_kernel" and ctx.fallbacks() == 0
            k2 = np.arange(F, dtype=np.float64) ** 2
            for o in outs:
                np.testing.assert_array_equal(o, outs[0])
                # (rounding: ~1e-5 at lag F - 1, where the mean is taken over one origin; a wrong sample: >= 0.5)
                np.testing.assert_allclose(o[1:, 0, :3], ref[1:, 0, :3], rtol=0, atol=1e-3)
                np.testing.assert_allclose(o[1:, 0, 0], k2[1:], rtol=0, atol=1e-3)
    finally:
        ctx.close()


def test_read_once_sort_every_instance_and_small_copy_paths(B):
    """Round 5's pre-pass changes, against the forms they replace: the spatial sort that keeps a frame's atoms in
    registers (cull_sort_reg_kernel<2..12>: one instance per 2048 atoms, frames of up to 12288; beyond that the loop
    form) against the multi-block sort (rdf_sort 0) and the loop form (rdf_sort 3), and the small host<->device copies
    made by a kernel (small_copy 1) against hipMemcpyAsync (0) — frame-summed and per-frame results, host and the
    one-sweep RDF+CN: the integers must be identical, whatever order the atoms of a cell end up in."""
    from mdproptools_amd._lib import Context

    rng = np.random.default_rng(5)
    ref_ctx, new_ctx = Context(0), Context(0)
    ref_ctx.set_option("rdf_sort", 0)
    ref_ctx.set_option("small_copy", 0)
    for c in (ref_ctx, new_ctx):
        c.set_option("rdf_cull", 1)
    try:
        for n in (700, 2048, 2049, 4100, 6200, 8191, 10000, 12288, 12289, 13000):
            F, L = 3, 46.0
            xyz = rng.uniform(-3.0, L + 3.0, (F, 3, n))  # (atoms outside the cell too: the sort wraps, the sweep does not)
            ty = rng.integers(1, 4, n).astype(np.int32)
            box = np.full((F, 3), L)
            rel = np.array([[1, 1], [1, 2], [2, 3], [3, 3]])
            for per_frame in (False, True):
                a = B.rdf_loop(xyz, ty, box, rel, 9.0, 0.05, 180, per_frame=per_frame, ctx=ref_ctx)
                b = B.rdf_loop(xyz, ty, box, rel, 9.0, 0.05, 180, per_frame=per_frame, ctx=new_ctx)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (n, per_frame)
            new_ctx.set_option("rdf_sort", 3)
            c = B.rdf_loop(xyz, ty, box, rel, 9.0, 0.05, 180, per_frame=False, ctx=new_ctx)
            new_ctx.set_option("rdf_sort", -1)
            assert np.array_equal(a[0].sum(axis=0) if a[0].ndim > 1 else a[0], c[0]), n
            ra = B.rdf_cn_loop(xyz, ty, box, rel, 9.0, 0.05, 180, [3.1, 4.0, 5.5, 2.2], per_frame=False, ctx=ref_ctx)
            rb = B.rdf_cn_loop(xyz, ty, box, rel, 9.0, 0.05, 180, [3.1, 4.0, 5.5, 2.2], per_frame=False, ctx=new_ctx)
            assert all(np.array_equal(u, v) for u, v in zip(ra[:2], rb[:2])) and np.array_equal(ra[2], rb[2]), n
    finally:
        ref_ctx.close()
        new_ctx.close()
