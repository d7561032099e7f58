#!/usr/bin/env python
"""tools/trace_gaps.py DIR [n_last] — from a rocprofv3 --kernel-trace --memory-copy-trace CSV output: the GPU timeline of the
last operations (kernels and copies, all streams), with the idle gap before each: where a step's time goes between its
kernels."""
import csv
import glob
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].split("::")[-1][:44]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", "")))
ev.sort()
ev = ev[-n_last:]
busy_until = ev[0][0]
tot_busy = tot_idle = 0
for s, e, name in ev:
    gap = s - busy_until
    if gap > 0:
        tot_idle += gap
    print("%9.1f us  idle before %6.1f us  dur %7.1f us  %s" % ((s - ev[0][0]) / 1e3, max(gap, 0) / 1e3, (e - s) / 1e3, name))
    if e > busy_until:
        tot_busy += e - max(s, busy_until)
        busy_until = e
print("span %.1f us: busy %.1f us, idle %.1f us" % ((busy_until - ev[0][0]) / 1e3, tot_busy / 1e3, tot_idle / 1e3))
