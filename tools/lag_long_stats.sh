#!/bin/bash
# tools/lag_long_stats.sh <tag> [option=value ...] — rocprofv3 --kernel-trace --stats over tools/run_secondary.py lag_long
# (10 000 frames x 50k entities through the batched full-lag path); the per-kernel table lands in
# gpurun_out/prof/<tag>_lag_long_kernel_stats.csv.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
MDHIP_OPTS="$*" timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${TAG}_lag_long -- python3 $R/tools/run_secondary.py lag_long 3 > $O/${TAG}_lag_long.log 2>&1
echo "rc=$?"; grep -v amdgpu $O/${TAG}_lag_long.log | tail -4
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/${TAG}_lag_long_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %6s total_ms %9.3f avg_us %9.1f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
