#!/usr/bin/env python
"""tools/show_steps.py LINE.json — the per-step timing of a bench.py line, in one screen."""
import json
import sys

for path in sys.argv[1:]:
    try:
        with open(path) as fh:
            line = [l for l in fh.read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as e:  # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    st = d.get("roofline", {}).get("step_ms") or {}
    print("%s: ms_per_step %.3f value %.4g launch_ms %s" % (path, d["ms_per_step"], d["value"],
                                                          d.get("roofline", {}).get("launch_ms")))
    if st:
        print("  step_ms min %.3f median %.3f p90 %.3f max %.3f | kernel+prepass median %.3f | ratio %.3f" % (
            st["min"], st["median"], st["p90"], st["max"], st["median_kernel_plus_prepass"],
            st["median_over_kernel_plus_prepass"] or 0))
        print("  raw   ", " ".join("%.2f" % v for v in st["raw"]))
        print("  kernel", " ".join("%.2f" % v for v in st["kernel_plus_prepass_ms"]))
        print("  call  ", " ".join("%.2f" % v for v in st["call_ms"]))
    print("  kernel:", d.get("config", {}).get("kernel"), "| workload:", d.get("config", {}).get("workload"))
    for leg in ("c1", "c1_alt", "c1_full", "residence"):
        v = d.get(leg)
        if isinstance(v, dict):
            print("  %s:" % leg, {k: v.get(k) for k in ("error", "value", "kernel", "kernel_ms", "prepass_ms", "ms_per_step",
                                                      "cost_per_pair_over_c2") if v.get(k) is not None})
