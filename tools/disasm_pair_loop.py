#!/usr/bin/env python3
"""
tools/disasm_pair_loop.py [kernel-substring] — opcode histogram of the hot loop of the packed-f32 pair kernel
(pair_hist_sj_kernel<3, true, false>), from the compiler's own assembly of mdproptools_amd/csrc/pair_sj.hip (hipcc -S, the
flags of mdproptools_amd/build.py; no GPU needed).

The hot loop is the software-pipelined sweep of the common variant (no per-pair wrap): two unrolled copies of
sweep_group_pk<false, 0, true, ...>, each holding one group of four j atoms = 4 pair slots of 64 pairs, recognisable as
the innermost loop whose body carries the prefetch (s_load_dwordx4/x8/x16 of the next group's records) between
v_pk_* instructions. Printed: per loop body the instruction counts by class, normalised per pair slot (64 pairs), which
is what DESIGN.md 4.1b's "8.1 VALU per 64 pairs" claims; the conditional part (bin guess under the in-cutoff mask) is
listed separately, as the instructions between s_and_saveexec and the s_mov exec restore.
"""
import collections
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(HERE, "mdproptools_amd", "csrc", "pair_sj.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
         "-Wno-unused-function", "-S", "--cuda-device-only"]


def classify(op):
    if op.startswith("v_pk_"):
        return "VALU packed f32"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "VALU compare"
    if op in ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32"):
        return "VALU transcendental"
    if op.startswith("v_") and "f64" in op:
        return "VALU f64"
    if op.startswith("v_"):
        return "VALU other"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"):
        return "VMEM"
    return "other"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "pair_hist_sj_kernelILi3ELb1ELb0EE"
    out = "/tmp/pair_sj_disasm.s"
    subprocess.check_call(["hipcc"] + FLAGS + ["-o", out, SRC], stderr=subprocess.DEVNULL, cwd="/tmp")
    lines = open(out).read().split("\n")
    start = next(k for k, ln in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(want), ln))
    end = next(k for k in range(start, len(lines)) if "s_endpgm" in lines[k])
    body = lines[start:end + 1]
    # basic blocks
    blocks, cur, name = collections.OrderedDict(), [], "entry"
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks[name] = cur
            name, cur = m.group(1), []
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        cur.append(t.split(";")[0].strip())
    blocks[name] = cur
    # loops: a block sequence from a label to a backward branch to that label; the hot one holds packed ops AND a
    # scalar prefetch
    names = list(blocks)
    loops = []
    for i, nm in enumerate(names):
        for ins in blocks[nm]:
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", ins)
            if m and m.group(1) in names and names.index(m.group(1)) <= i:
                j = names.index(m.group(1))
                seq = [x for k in range(j, i + 1) for x in blocks[names[k]]]
                npk = sum(1 for x in seq if x.startswith("v_pk_"))
                nld = sum(1 for x in seq if x.startswith("s_load"))
                if npk >= 24 and nld >= 1:  # two unrolled sweeps of 12 packed instructions + the prefetch
                    loops.append((len(seq), names[j], nm, seq))
    if not loops:
        print("no loop with packed ops and a scalar prefetch found")
        return
    loops.sort()
    size, first, last, seq = loops[0]  # the innermost such loop: the software-pipelined sweep, two groups per trip
    ops = collections.Counter(x.split()[0] for x in seq)
    npk = sum(v for k, v in ops.items() if k.startswith("v_pk_"))
    slots = npk / 3.0  # 6 packed instructions per two pair slots
    print("kernel %s\nloop %s .. %s: %d instructions per trip, %d packed -> %.0f pair slots of 64 pairs"
          % (want, first, last, size, npk, slots))
    # The pair path proper = the packed distance chain + the hand-written pair block (bin_pair): its opcodes occur
    # nowhere else in the loop, so their counts per pair slot can be read off the histogram. Everything else in the
    # loop's address range is the wave-uniformly skipped ambiguous-pair code (queue push: v_mbcnt, ds_write; lost-entry
    # atomics), the queue-drain check and the loop control / prefetch.
    sig_uncond = ["v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32", "v_cmpx_gt_f32_e64"]
    sig_cond = ["v_sqrt_f32", "v_fma_f32", "v_fract_f32", "v_cvt_i32_f32", "v_cmpx_ge_f32_e64", "v_lshl_add_u32"]
    sig_other = ["ds_add_u32", "s_andn2_b64", "s_mov_b64", "s_and_saveexec_b64", "s_cbranch_execz", "s_ff1_i32_b64", "s_bitset0_b64",
                 "s_load_dwordx16", "s_load_dwordx8", "s_load_dwordx4"]
    print("\npair path, per trip and per pair slot (64 pairs):")
    u = c = 0
    for k in sig_uncond:
        print("  %-22s %3d   %.2f   every slot" % (k, ops.get(k, 0), ops.get(k, 0) / slots))
        u += ops.get(k, 0)
    for k in sig_cond:
        # (v_fma_f32 and v_lshl_add_u32 also occur in the inlined exact chain of the queue drain, which sits inside the
        # loop's address range: the pair block itself — bin_pair's asm text — holds exactly one of each per slot)
        n_k = min(ops.get(k, 0), int(slots))
        print("  %-22s %3d   %.2f   under the in-cutoff mask (skipped by s_cbranch_execz when no lane is inside)%s"
              % (k, n_k, n_k / slots, "" if n_k == ops.get(k, 0) else "   [+%d in the drain code]" % (ops[k] - n_k)))
        c += n_k
    for k in sig_other:
        if ops.get(k, 0):
            print("  %-22s %3d   %.2f" % (k, ops[k], ops[k] / slots))
    print("\nVALU of the pair path per pair slot: %.2f every slot + %.2f under the in-cutoff mask; at BASELINE C2 ~0.69 of "
          "the (wave, slot) pairs have a lane inside: %.2f + 0.69 x %.2f = %.2f VALU per 64 pairs"
          % (u / slots, c / slots, u / slots, c / slots, u / slots + 0.69 * c / slots))
    tot = collections.Counter(classify(x.split()[0]) for x in seq)
    print("\nwhole loop range by class (includes the skipped ambiguous-pair code):")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %-22s %4d" % (k, v))
    print("\nall opcodes of the loop range:")
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]):
        print("  %-28s %4d" % (k, v))


if __name__ == "__main__":
    main()
