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
    assert lib.mdhip_version() == 100


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
    # bench.py: only inside the cpu_baseline functions
    src = open(os.path.join(REPO, "bench.py")).read()
    for m in re.finditer(r"^\s*(?:from|import)\s+oracle\b", src, flags=re.M):
        enclosing = re.findall(r"^def (\w+)\(", src[: m.start()], flags=re.M)[-1]
        assert enclosing.startswith("cpu_baseline"), enclosing


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
