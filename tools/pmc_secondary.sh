#!/bin/bash
# tools/pmc_secondary.sh <tag> [workloads...] — rocprofv3 over tools/run_secondary.py, one workload per process:
#   pass 0  --kernel-trace --stats           (per-kernel durations of the call)
#   pass 1-5 --pmc <group> --kernel-trace    (one counter group per run, as gpurun requires; FETCH_SIZE and WRITE_SIZE
#                                             in passes of their own, as MI355X_MICROARCH.md prescribes)
# Output: gpurun_out/pmc2/<tag>_<workload>_pN_*.csv; tools/pmc_secondary_summarize.py turns them into
# <tag>_secondary_summary.txt and <tag>_secondary.json (copied to profiles/ by hand).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
WL=${@:-"msd_pairs msd_windows com flux lag_fft lag_long lag_diff acf_fft acf_direct cumtrapz residence"}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
P3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
for w in $WL; do
  reps=3; case $w in lag_diff|acf_direct) reps=2 ;; esac
  echo "== $w $(date +%T)"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ${TAG}_${w}_p0 -- python3 $R/tools/run_secondary.py $w $reps > $OUT/${TAG}_${w}_p0.log 2>&1 || echo "stats pass failed for $w"
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o ${TAG}_${w}_p$i -- python3 $R/tools/run_secondary.py $w $reps > $OUT/${TAG}_${w}_p$i.log 2>&1 || echo "pmc pass $i failed for $w"
  done
  tail -1 $OUT/${TAG}_${w}_p0.log
done
python3 $R/tools/pmc_secondary_summarize.py $OUT $TAG
