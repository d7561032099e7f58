#!/usr/bin/env python3
"""
tools/pmc_secondary_summarize.py <dir> <tag> — per-workload, per-kernel summary of the passes tools/pmc_secondary.sh
wrote: <dir>/<tag>_secondary_summary.txt (text) and <dir>/<tag>_secondary.json, keyed by workload:

  {"reps": R, "kernels": {name: {"calls_per_rep": n, "avg_us": ..., "total_us_per_rep": ..., counters (mean per
   dispatch) ...}}, "per_call": {"kernel_us": ..., "fetch_bytes_corrected": ..., "write_bytes": ...,
   "hbm_bytes": ..., "lds_idx_active": ..., ...}}

`per_call` sums over every libmdhip kernel of ONE library call (all dispatches of the process divided by the number of
repetitions). HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE from passes of their own, KB units,
FETCH_SIZE doubled on gfx950. Only kernels of libmdhip.so are kept (torch's generators and copies are dropped).
bench.py reads profiles/pmc_secondary.json for the `traffic` fields of the c4 / c5 rooflines.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

KEEP = ("msd_", "segment_", "type_sum", "mol_flux", "fft_", "xcorr_", "lag_msd", "transpose", "cumtrapz", "scan",
        "r2c_", "c2r_", "conj_copy", "spectrum", "col_sum", "col_mean", "frame_sq", "power_", "fold_items",
        "real_to_complex", "trapz", "shell_pairs", "run_starts", "residence_lag", "DeviceRadixSort", "rocprim")


HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECONDARY_SOURCES = ["msd.hip", "msd_fft.hip", "msd_fft_w12.h", "msd_fft_w12r.h", "segment_com.hip", "xcorr.hip", "fft_pow2.hip", "scan.hip",
                     "residence.hip"]


def source_hash():
    """Hash of the sources of the kernels measured here: bench.py uses a counter only when it was taken from the code
    that is present (same rule as profiles/pmc_kernels.json for the pair kernels)."""
    import hashlib

    h = hashlib.sha256()
    for name in SECONDARY_SOURCES:
        with open(os.path.join(HERE, "mdproptools_amd", "csrc", name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def ours(raw):
    """libmdhip kernels only: torch's generators, copies and scans (at::native::...) are dropped."""
    # (the residence call's sort is the one vendor-library kernel family in the product: kept, so that its share shows)
    return "at::" not in raw and "tensor_kernel" not in raw and "rocclr" not in raw and any(s in raw for s in KEEP)


def short_name(k):
    k = k.strip().replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*\)$", "", k)
    k = re.sub(r"^void\s+", "", k)
    k = re.sub(r"^(\w+::)+", "", k)
    return k.replace(" [clone .kd]", "").replace(".kd", "")


def main():
    out, tag = sys.argv[1], sys.argv[2]
    workloads = sorted({re.match(r".*/%s_(.+)_p0_kernel_trace\.csv$" % re.escape(tag), f).group(1)
                        for f in glob.glob("%s/%s_*_p0_kernel_trace.csv" % (out, tag))})
    db = {}
    lines = []
    for w in workloads:
        reps = 0
        for ln in open("%s/%s_%s_p0.log" % (out, tag, w)):
            if ln.startswith(w + " "):
                reps += 1
        reps = max(reps, 1)
        kern = collections.defaultdict(lambda: {"n": 0, "dur": 0.0})
        for row in csv.DictReader(open("%s/%s_%s_p0_kernel_trace.csv" % (out, tag, w))):
            if not ours(row["Kernel_Name"]):
                continue
            name = short_name(row["Kernel_Name"])
            kern[name]["n"] += 1
            kern[name]["dur"] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3  # us
        ctr = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in sorted(glob.glob("%s/%s_%s_p[1-9]_counter_collection.csv" % (out, tag, w))):
            for row in csv.DictReader(open(f)):
                if ours(row["Kernel_Name"]):
                    ctr[short_name(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        entry = {"reps": reps, "kernels": {}, "per_call": collections.defaultdict(float), "tag": tag,
                 "source_hash": source_hash()}
        lines.append("== %s (%d repetitions) ==" % (w, reps))
        for name, k in sorted(kern.items(), key=lambda kv: -kv[1]["dur"]):
            e = {"calls_per_rep": k["n"] / reps, "avg_us": k["dur"] / k["n"], "total_us_per_rep": k["dur"] / reps}
            entry["per_call"]["kernel_us"] += k["dur"] / reps
            for c, v in sorted(ctr.get(name, {}).items()):
                e[c] = sum(v) / len(v)
                entry["per_call"][c] += sum(v) / reps
            entry["kernels"][name] = e
            lines.append("[%s] %.3g launches per call, avg %.2f us, %.2f us per call" % (
                name, e["calls_per_rep"], e["avg_us"], e["total_us_per_rep"]))
            for c in sorted(e):
                if c.isupper() or c.startswith("SQ_") or c.startswith("GRBM"):
                    lines.append("    %-26s %.6g" % (c, e[c]))
        pc = entry["per_call"]
        if "FETCH_SIZE" in pc or "WRITE_SIZE" in pc:
            pc["fetch_bytes_corrected"] = 2.0 * pc.get("FETCH_SIZE", 0.0) * 1024.0
            pc["write_bytes"] = pc.get("WRITE_SIZE", 0.0) * 1024.0
            pc["hbm_bytes"] = pc["fetch_bytes_corrected"] + pc["write_bytes"]
        entry["per_call"] = dict(pc)
        lines.append("per call: " + ", ".join("%s=%.6g" % (c, v) for c, v in sorted(entry["per_call"].items())))
        lines.append("")
        db[w] = entry
    with open("%s/%s_secondary_summary.txt" % (out, tag), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    json.dump(db, open("%s/%s_secondary.json" % (out, tag), "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
