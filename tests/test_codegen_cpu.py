"""
A static guard for the hand-scheduled code: the exactness of the default RDF kernel rests on inline
asm that rewrites `exec` behind the compiler's back, on `-ffp-contract=off` / `#pragma clang fp contract(off)` holding on
the reference's f64 chain (/root/reference/mdproptools/structural/rdf_cn.py:44-69: subtract, wrap, square, add — no FMA),
and the roofline's occupancy assumptions rest on register budgets. hipcc cross-compiles gfx950 without a GPU, so all of
that is checked HERE from the compiler's own assembly (`-S --cuda-device-only`, the flags of mdproptools_amd/build.py):
a compiler bump that fuses the chain, moves a VALU instruction between an exec-rewriting block and its restore, or
halves the occupancy fails this test instead of silently changing results or speed.
The two translation units take ~30 s each to compile; the assembly is cached under /tmp by source hash.
"""
import hashlib
import os
import re
import subprocess

import pytest

from mdproptools_amd import build as bld

CSRC = bld.CSRC
FLAGS = [f for f in bld.CFLAGS if f != "-Wall"] + ["-S", "--cuda-device-only", "-Wno-unused-command-line-argument"]


def _asm(src):
    path = os.path.join(CSRC, src)
    h = hashlib.sha256()
    for p in [path] + bld.HEADERS:
        with open(p, "rb") as fh:
            h.update(fh.read())
    out = "/tmp/mdhip_codegen_%s_%s.s" % (os.path.splitext(src)[0], h.hexdigest()[:16])
    if not os.path.exists(out):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.check_call([bld._hipcc()] + FLAGS + ["-o", tmp, path], cwd="/tmp", stderr=subprocess.DEVNULL)
        os.replace(tmp, out)
    with open(out) as fh:
        return fh.read()


def _kernels(text):
    """name -> (body lines, metadata dict) for every kernel of an assembly file."""
    lines = text.split("\n")
    bodies = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z[A-Za-z0-9_]*):", ln)
        if m:
            e = next(x for x in range(i, len(lines)) if lines[x].startswith(".Lfunc_end"))
            bodies[m.group(1)] = lines[i:e]
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:\s+\.\w+:.*\n)+)", text):
        d = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
        meta[m.group(1)] = d
    return {k: (v, meta.get(k, {})) for k, v in bodies.items()}


@pytest.fixture(scope="module")
def pair_sj():
    return _kernels(_asm("pair_sj.hip"))


def _find(kern, sub):
    hits = [k for k in kern if sub in k]
    assert len(hits) == 1, (sub, hits)
    return kern[hits[0]]


def _ops(body):
    for ln in body:
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        yield s.split()[0], s


def test_reference_f64_chain_is_not_fused(pair_sj):
    """pair_hist_sj_kernel<2, ., false> is the all-f64 sweep: d = xi - xj, the single +-L wrap, (dx2 + dy2) + dz2, the
    edge-table compare — the reference's operations one by one. Not one fused multiply-add may appear in it."""
    for rows in ("Lb1", "Lb0"):
        body, meta = _find(pair_sj, "pair_hist_sj_kernelILi2E%sELb0ELb0EE" % rows)
        fused = [s for op, s in _ops(body) if re.match(r"v_(fma|fmac|mad)_f64", op)]
        assert fused == [], fused[:3]
        assert any(op == "v_mul_f64" for op, _ in _ops(body)) and any(op == "v_add_f64" for op, _ in _ops(body))


def test_packed_kernels_f64_fma_count_is_the_sqrt_and_division_expansions(pair_sj):
    """The packed sweeps (MODE 3..6) resolve ambiguous pairs with the same f64 chain + sqrt(rsq) / ddr + trunc. The
    chain is the function the all-f64 sweep uses (unfused: the test above); correctly rounded f64 sqrt and division
    are expanded by the compiler into Newton steps that DO use v_fma_f64 — 23 of them with ROCm 7.2. More than that means
    something else got fused: look before raising the number."""
    for mode in (3, 4, 5, 6):
        for rows in ("Lb1", "Lb0"):
            for cn in ("Lb0", "Lb1"):
                body, _ = _find(pair_sj, "pair_hist_sj_kernelILi%dE%sE%sELb0EE" % (mode, rows, cn))
                n = sum(1 for op, _ in _ops(body) if re.match(r"v_(fma|fmac|mad)_f64", op))
                assert 0 < n <= 23, (mode, rows, cn, n)


EXEC_WRITE = re.compile(r"^(v_cmpx_\w+|s_\w+_saveexec_b64)\b|^s_\w+\s+exec\b|^s_\w+\s+exec_(lo|hi)\b")


def test_every_inline_block_that_rewrites_exec_puts_it_back(pair_sj):
    """Inline asm appears between ';;#ASMSTART' and ';;#ASMEND'. A block that writes exec (v_cmpx, s_mov exec, ...) must
    END with exec restored from a scalar register pair (`s_mov_b64 exec, s[a:b]`: the sweep's `full` mask) — i.e. the
    restore is INSIDE the block, so no compiler-scheduled vector instruction can ever sit between the rewrite and the
    restore. Out-of-line parts of a block (.subsection 1) must end in a branch back into it."""
    checked = 0
    for name, (body, _) in pair_sj.items():
        if "pair_hist_sj_kernel" not in name:
            continue
        i = 0
        while i < len(body):
            if "#ASMSTART" not in body[i]:
                i += 1
                continue
            j = next(x for x in range(i, len(body)) if "#ASMEND" in body[x])
            block = [s.strip() for s in body[i + 1:j] if s.strip()]
            i = j + 1
            main, sub, cur = [], [], None
            for s in block:
                if s.startswith(".subsection"):
                    cur = sub if s.split()[1] != "0" else None
                    continue
                (sub if cur is not None else main).append(s)
            instr = [s for s in main if not s.endswith(":") and not s.startswith((";", "."))]
            writes = [k for k, s in enumerate(instr) if EXEC_WRITE.match(s)]
            if not writes and not any(EXEC_WRITE.match(s) for s in sub):
                continue
            checked += 1
            assert writes, (name, block)
            last = instr[writes[-1]]
            assert re.match(r"^s_mov_b64\s+exec,\s*s\[\d+:\d+\]", last), (name, last)
            # nothing vector-side after the restore inside the block either
            assert not any(s.startswith(("v_", "ds_")) for s in instr[writes[-1] + 1:]), (name, block)
            if sub:
                tail = [s for s in sub if not s.endswith(":") and not s.startswith((";", "."))][-1]
                assert tail.startswith("s_branch"), (name, tail)
    assert checked > 100  # (the packed sweeps hold hundreds of such blocks; zero would mean the parser saw nothing)


def test_resource_budgets_of_the_headline_kernels(pair_sj):
    """What the roofline's occupancy assumptions depend on: the C2 / C3 headline kernel <3, true, false> at <= 80 VGPRs
    (6 waves per SIMD: profiles/r02_ubench_valu.json's roof is priced at 6) with at most a few spilled registers outside
    the sweep (ROCm 7.2: 2 registers, 12 bytes of scratch)."""
    for mode in (3, 4):
        _, meta = _find(pair_sj, "pair_hist_sj_kernelILi%dELb1ELb0ELb0EE" % mode)
        assert int(meta["vgpr_count"]) <= 80, meta
        assert int(meta["vgpr_spill_count"]) <= 3, meta
        assert int(meta["private_segment_fixed_size"]) <= 16, meta
    # the one-sweep RDF + CN kernel keeps the same occupancy (a few spilled registers in its rare CN-check path)
    _, meta = _find(pair_sj, "pair_hist_sj_kernelILi3ELb1ELb1ELb0EE")
    assert int(meta["vgpr_count"]) <= 80 and int(meta["private_segment_fixed_size"]) <= 32, meta
    # round 6: the 16-wave instance (one block per CU, the whole LDS for one histogram) at <= 128 VGPRs — 4 waves per SIMD —
    # and without a single spilled register
    for mode in (3, 4):
        _, meta = _find(pair_sj, "pair_hist_sj_kernelILi%dELb1ELb0ELb1EE" % mode)
        assert int(meta["vgpr_count"]) <= 128 and int(meta["vgpr_spill_count"]) == 0, meta
    # the all-f64 sweep (f64_only leg)
    _, meta = _find(pair_sj, "pair_hist_sj_kernelILi2ELb1ELb0ELb0EE")
    assert int(meta["vgpr_spill_count"]) == 0 and int(meta["private_segment_fixed_size"]) == 0, meta


def _regs(tok):
    """'v86' -> {86}; 'v[86:89]' -> {86..89}; anything else -> empty."""
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def wide_store_hazards(body, idle=2):
    """(store, writer) pairs: a buffer store of more than 8 bytes and a vector-ALU instruction that writes one of its DATA
    registers fewer than `idle` cycles behind it (every instruction is one cycle, `s_nop k` is k + 1)."""
    ops = list(_ops(body))
    out = []
    for i, (op, s) in enumerate(ops):
        if not re.match(r"buffer_store_(dwordx[34]|format_xyzw?)", op):
            continue
        data = _regs(s.split()[1].rstrip(","))
        gap = 0
        for op2, s2 in ops[i + 1:]:
            if gap >= idle:
                break
            if op2.startswith("v_") and not op2.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                dst = _regs(s2.split()[1].rstrip(","))
                if op2.startswith("v_swap"):
                    dst |= _regs(s2.split()[2].rstrip(","))
                if dst & data:
                    out.append((s, s2))
            m = re.match(r"s_nop\s+(\d+)", s2)
            gap += int(m.group(1)) + 1 if m else 1
    return out


def test_scanner_sees_the_pattern_that_went_wrong():
    bad = ["k:", "\tbuffer_store_dwordx4 v[86:89], v114, s[56:59], s74 offen sc1", "\tv_cndmask_b32_e32 v86, v148, v126, vcc"]
    assert len(wide_store_hazards(bad)) == 1
    ok = [bad[0], bad[1], "\ts_nop 1", bad[2]]
    assert wide_store_hazards(ok) == []
    ok2 = [bad[0], bad[1], "\ts_mov_b32 s55, s59", "\tv_add_f64 v[90:91], v[86:87], v[88:89]", bad[2]]
    assert wide_store_hazards(ok2) == []


def test_staging_stores_keep_their_data_registers_for_two_cycles():
    """The full-lag MSD kernels' staging stores (16 bytes, scalar offset in a register): the compiler does not guard
    that form against a vector write of the data registers in the next cycle, and on gfx950 such a write changes what
    some lanes store (msd_fft_w12.h, W12_STORE_GUARD: found on an integer ramp at 5120 < F <= 6144). Every kernel of
    msd_fft.hip is scanned."""
    kern = _kernels(_asm("msd_fft.hip"))
    assert any("msd_power_w12_kernel" in k for k in kern)
    found = {k: wide_store_hazards(body) for k, (body, _) in kern.items()}
    found = {k: v for k, v in found.items() if v}
    assert found == {}, found


def test_long_series_kernels_stay_inside_their_register_budgets():
    """Round 6. msd_power_w12p_kernel (msd_fft_w12r.h) runs twelve waves per block — 168 registers each — and requests its
    next samples a transform ahead: a spilled register's reload waits for those loads (scratch and global memory share a
    counter), so the kernel is only as fast as measured while (almost) nothing spills: <= 12 registers (ROCm 7.2: 8), all in
    its prologue and epilogue. The first form of the kernel hoisted every table entry out of the series loop and spilled
    206. fft_power_pass_kernel (fft_pow2.hip) likewise: three waves per SIMD, <= 8 spilled registers (hoisted, it spilled
    79-147 and ran at a quarter of the speed)."""
    kern = _kernels(_asm("msd_fft.hip"))
    for inst in ("Lb1", "Lb0"):
        _, meta = _find(kern, "msd_power_w12p_kernelI%sE" % inst)
        assert int(meta["vgpr_count"]) <= 168 and int(meta["vgpr_spill_count"]) <= 12, meta
    # No FLAT memory instruction anywhere in msd_fft.hip: a table pointer passed through an opaque asm operand (to keep
    # loop-invariant reads from being hoisted) loses its address space, every read of it becomes flat_load + a wait for ALL
    # memory counters — the prefetched samples included. That cost 0.5-1.2 ms per long call until the opaque operand became an
    # integer offset.
    flat = {k: [s_ for op, s_ in _ops(body) if op.startswith("flat_")][:2] for k, (body, _) in kern.items()}
    flat = {k: v for k, v in flat.items() if v}
    assert flat == {}, flat
    # msd_power_w1_kernel (one wave per series, F <= 1536): the one- and two-transform instances spill nothing; the
    # three-transform one keeps 24 samples in flight per class next to three sets of sums and spills ~40 registers (ROCm 7.2:
    # 40) — measured as it is; more would say something else got hoisted
    for d2, cap in ((1, 0), (2, 0), (3, 48)):
        _, meta = _find(kern, "msd_power_w1_kernelILi%dE" % d2)
        assert int(meta["vgpr_count"]) <= 168 and int(meta["vgpr_spill_count"]) <= cap, (d2, meta)
    fft = _kernels(_asm("fft_pow2.hip"))
    hits = [k for k in fft if "fft_power_pass_kernel" in k]
    assert len(hits) >= 5
    for k in hits:
        meta = fft[k][1]
        assert int(meta["vgpr_count"]) <= 168 and int(meta["vgpr_spill_count"]) <= 8, (k, meta)
