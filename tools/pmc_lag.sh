#!/bin/bash
# tools/pmc_lag.sh <tag> <kernel-substring> [run_lag_fft.py args] — kernel trace + three PMC passes over one C4-size full-lag
# call; per-dispatch means of the named kernel -> gpurun_out/pmc/<tag>_summary.txt
set -u
TAG=$1; KSUB=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_trace -- python3 $R/tools/run_lag_fft.py "$@" > $OUT/${TAG}_trace.log 2>&1
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
P3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o ${TAG}_p$i -- python3 $R/tools/run_lag_fft.py "$@" > $OUT/${TAG}_p$i.log 2>&1
done
python3 - "$OUT" "$TAG" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, tag, ksub = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(f"{out}/{tag}_p*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if ksub in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"{out}/{tag}_summary.txt", "w") as fh:
    fh.write(f"[{ksub}] mean per dispatch over {max((len(v) for v in acc.values()), default=0)} dispatches\n")
    for c, v in sorted(acc.items()):
        fh.write(f"  {c:28s} {sum(v)/len(v):.6g}\n")
    for f in glob.glob(f"{out}/{tag}_trace_kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            fh.write("  %-60s calls %s avg %.1f us\n" % (row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e3))
print(open(f"{out}/{tag}_summary.txt").read())
PY
