#!/usr/bin/env python
"""tools/trace_overlap.py DIR — from a rocprofv3 --kernel-trace --memory-copy-trace CSV output: the timeline of the large
host-to-device copies against the pair kernel."""
import csv
import glob
import sys

d = sys.argv[1]
kern = [f for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)]
mem = [f for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)]
ev = []
for f in kern:
    for r in csv.DictReader(open(f)):
        if "pair_hist_sj" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERNEL pair_hist_sj"))
for f in mem:
    for r in csv.DictReader(open(f)):
        # (the trace carries no byte count: the staging copies of a C2 step — 48 MB — are the ones that take > 0.3 ms)
        if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 300000:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY %s (stream %s)" % (
                r.get("Direction", "").replace("MEMORY_COPY_", ""), r.get("Stream_Id", "?"))))
ev.sort()
t0 = ev[0][0] if ev else 0
for s, e, what in ev[-30:]:
    print("%10.3f ms  +%7.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, what))
