#!/bin/bash
# tools/pmc.sh <tag> <workload C2|C3> [bench args...] — rocprofv3 PMC passes over bench.py's headline loop on the GPU
# box (one counter group per run, --kernel-trace only, as gpurun requires; separate passes for FETCH_SIZE and
# WRITE_SIZE as MI355X_MICROARCH.md prescribes). Output: gpurun_out/pmc/<tag>_pN_*.csv, a per-kernel summary
# gpurun_out/pmc/<tag>_summary.txt and the entries tools/pmc_summarize.py merges into profiles/pmc_kernels.json.
set -u
TAG=$1; shift
WL=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"
P3="GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
P6="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT -o ${TAG}_p$i -- python3 $R/bench.py --no-cpu-baseline --no-legs --steps 3 --warmup 1 "$@" > $OUT/${TAG}_p$i.log 2>&1
done
python3 $R/tools/pmc_summarize.py $OUT $TAG $WL
