#!/usr/bin/env python3
"""
tools/pmc_summarize.py <dir> <tag> <workload> — per-kernel means of the rocprofv3 --pmc passes tools/pmc.sh wrote,
as text (<dir>/<tag>_summary.txt) and as entries "<kernel>|<workload>" of <dir>/<tag>_kernels.json, which
`tools/pmc_summarize.py --merge <json> ...` folds into profiles/pmc_kernels.json (the file bench.py prices the pair
kernel's instruction stream from). Every entry carries the hash of the pair-kernel sources it was measured on, so
that bench.py can refuse a count taken from other code.

HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE from separate passes, KB units, FETCH_SIZE doubled
(gfx950 tallies 128-byte read requests at 64 bytes; the factor was checked here on msd_pairs_kernel against its known
24*E*F bytes).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIR_SOURCES = ["pair_sj.hip", "pair_common.h", "pair_hist.hip", "pair_cull.hip"]


def source_hash():
    h = hashlib.sha256()
    for name in PAIR_SOURCES:
        with open(os.path.join(HERE, "mdproptools_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def short_name(k):
    """'void mdpair::(anonymous namespace)::pair_hist_sj_kernel<3, true>(mdpair::PairArgs)' -> the name the
    library reports (mdhip_last_kernel_name)."""
    k = k.strip().replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*\)$", "", k)                   # argument list
    k = re.sub(r"^void\s+", "", k)
    k = re.sub(r"^(\w+::)+", "", k)
    k = k.replace(" [clone .kd]", "").replace(".kd", "")
    # (round 6: the pair kernel has a fourth template parameter, BIG; the library reports the instances without it by their
    # three-parameter names — "<3, true, false>" — and the profiler prints the default)
    return re.sub(r"(pair_hist_sj_kernel<\d+, (?:true|false), (?:true|false)), false>", r"\1>", k)


def merge(dst, srcs):
    db = {}
    if os.path.exists(dst):
        db = json.load(open(dst))
    for s in srcs:
        db.update(json.load(open(s)))
    json.dump(db, open(dst, "w"), indent=1, sort_keys=True)
    print("merged", len(srcs), "file(s) into", dst, "->", len(db), "entries")


def main():
    if sys.argv[1] == "--merge":
        return merge(sys.argv[2], sys.argv[3:])
    out, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob("%s/%s_p*_counter_collection.csv" % (out, tag))):
        for row in csv.DictReader(open(f)):
            acc[short_name(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    entries = {}
    with open("%s/%s_summary.txt" % (out, tag), "w") as fh:
        for k, d in sorted(acc.items()):
            if not any(s in k for s in ("pair_hist", "msd_", "segment_", "xcorr_", "lag_msd", "merge_slices")):
                continue
            n = max(len(v) for v in d.values())
            fh.write("[%s] mean per dispatch over %d dispatches\n" % (k, n))
            m = {c: sum(v) / len(v) for c, v in d.items()}
            for c, v in sorted(m.items()):
                fh.write("  %-28s %.6g\n" % (c, v))
            if "pair_hist_sj_kernel" in k and "SQ_INSTS_VALU" in m:
                e = dict(m)
                e["tag"] = tag
                e["workload"] = workload
                e["source_hash"] = source_hash()
                if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                    e["fetch_bytes_corrected"] = 2.0 * m["FETCH_SIZE"] * 1024.0
                    e["write_bytes"] = m["WRITE_SIZE"] * 1024.0
                    e["hbm_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
                if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
                    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
                    e["valu_busy"] = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m:
                    e["wait_inst_any_over_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
                entries["%s|%s" % (k, workload)] = e
    json.dump(entries, open("%s/%s_kernels.json" % (out, tag), "w"), indent=1, sort_keys=True)
    print(open("%s/%s_summary.txt" % (out, tag)).read())


if __name__ == "__main__":
    main()
