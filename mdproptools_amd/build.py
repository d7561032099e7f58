"""
Build recipe for libmdhip.so (hand-written HIP for gfx950 behind the C-ABI of include/mdhip.h).

    python -m mdproptools_amd.build            # incremental
    python -m mdproptools_amd.build --force

hipcc cross-compiles gfx950 without a GPU. The library is built IN-TREE
(mdproptools_amd/libmdhip.so) so that it travels to the GPU box with the
repository snapshot. -ffp-contract=off is set for every translation unit: the
RDF/CN path needs the reference's unfused double arithmetic for bit-exact bin
counts, and the kernels that may fuse (correlation, lag sums) call fma explicitly.
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libmdhip.so")
SOURCES = ["mdhip_ctx.hip", "pair_hist.hip", "pair_dense.hip", "pair_cull.hip", "pair_sj.hip", "segment_com.hip", "msd.hip", "msd_fft.hip", "xcorr.hip", "fft_pow2.hip", "scan.hip", "residence.hip",
           "dump_reader.cpp"]
HEADERS = [os.path.join(CSRC, "ctx.h"), os.path.join(CSRC, "pair_common.h"), os.path.join(CSRC, "msd_fft_w12.h"), os.path.join(CSRC, "msd_fft_w12r.h"), os.path.join(os.path.dirname(HERE), "include", "mdhip.h")]
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=" + ARCH,
          "-Wall", "-Wno-unused-function"]
LDFLAGS = ["-shared", "-fPIC", "--offload-arch=" + ARCH, "-L/opt/rocm/lib", "-lpthread",
           "-Wl,-rpath,/opt/rocm/lib"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmdhip.so cannot be built")
    return exe


def source_id():
    """sha256 (16 hex digits) over every source the library is compiled from, in a fixed order: what
    `mdhip_build_id()` of a library built by this recipe returns. bench.py prints both, so that a number can be tied
    to the code that produced it even though the shipped .so is prebuilt."""
    import hashlib

    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0")
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libmdhip.so; returns its path."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    build_id = source_id()
    id_file = os.path.join(OBJ, "build_id.txt")
    try:
        with open(id_file) as fh:
            id_stale = fh.read().strip() != build_id
    except OSError:
        id_stale = True
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        carries_id = src == "mdhip_ctx.hip"  # the one unit that holds the build id: rebuilt whenever any source changed
        if force or _stale(o, [s] + HEADERS) or (carries_id and id_stale):
            if src.endswith(".hip"):
                cmd = [hipcc] + CFLAGS + (['-DMDHIP_BUILD_ID="%s"' % build_id] if carries_id else []) + ["-c", s, "-o", o]
            else:  # host-only translation unit
                cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-Wall", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc] + objs + LDFLAGS + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    with open(id_file, "w") as fh:
        fh.write(build_id + "\n")
    return LIB


VARIANT_DIR = os.path.join(os.path.dirname(HERE), "tools", "_bin")


def build_variant(name, src, extra_flags, force=False):
    """A variant BUILD of the library — `src` recompiled with `extra_flags`, every other object taken from the regular
    build — as tools/_bin/libmdhip_<name>.so (git-ignored; travels to the GPU box with the snapshot). Load it with
    MDHIP_LIB=<path> (mdproptools_amd/_lib.py). Used for A/B measurements and for the debug build the GPU suite runs
    (-DPK_CAPCHECK: the packed sweep's queue-capacity check, tests/test_gpu_hardening.py)."""
    build()
    os.makedirs(os.path.join(VARIANT_DIR, "obj_" + name), exist_ok=True)
    s = os.path.join(CSRC, src)
    o = os.path.join(VARIANT_DIR, "obj_" + name, os.path.splitext(src)[0] + ".o")
    lib = os.path.join(VARIANT_DIR, "libmdhip_%s.so" % name)
    hipcc = _hipcc()
    if force or _stale(o, [s] + HEADERS):
        r = subprocess.run([hipcc] + CFLAGS + list(extra_flags) + ["-c", s, "-o", o], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s (%s):\n%s" % (src, name, r.stdout.decode(errors="replace")))
    objs = [o if src_k == src else os.path.join(OBJ, os.path.splitext(src_k)[0] + ".o") for src_k in SOURCES]
    if force or _stale(lib, objs):
        r = subprocess.run([hipcc] + objs + LDFLAGS + ["-o", lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed (%s):\n%s" % (name, r.stdout.decode(errors="replace")))
    return lib


def build_capcheck():
    """The debug build with the packed sweep's queue-capacity check compiled in (pair_sj.hip, PK_CAPCHECK)."""
    return build_variant("capcheck", "pair_sj.hip", ["-DPK_CAPCHECK"])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
