#!/bin/bash
# tools/pmc_kernel.sh <tag> <kernel-name-substring> <run_kernel.py args...> — rocprofv3 PMC passes (one counter
# group per run, --kernel-trace only) over tools/run_kernel.py; prints per-dispatch means for the named kernel.
set -u
TAG=$1; KSUB=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
P3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_RD SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o ${TAG}_p$i -- python3 $R/tools/run_kernel.py "$@" > $OUT/${TAG}_p$i.log 2>&1
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
print(open(f"{out}/{tag}_summary.txt").read())
PY
