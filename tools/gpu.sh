#!/bin/bash
# tools/gpu.sh [steps...] — GPU-box sequences (one script for every round; ROUND names the output files, default r06);
# every step writes under gpurun_out/. Usage on this side: gpurun --timeout N -- 'bash tools/gpu.sh tests bench'.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${TAG:-x}
ROUND=${ROUND:-r06}
L=mdproptools_amd/libmdhip.so
LP=${LP:-tools/_bin/libmdhip_prev.so}   # the library of the round before (A/B), built by tools/build_variant.sh
for s in "$@"; do
  echo "== $s $(date +%T)"
  case $s in
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/gpu_tests.log; [ $rc -eq 0 ] || exit 1 ;;
    tests_k=*) K=${s#tests_k=}; timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$K" > $O/gpu_tests_k.log 2>&1; rc=$?; echo "tests_k rc=$rc"; tail -15 $O/gpu_tests_k.log; [ $rc -eq 0 ] || exit 1 ;;
    bench) timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_${TAG}.json 2> $O/bench_err_${TAG}.log; echo "bench rc=$?"; tail -3 $O/bench_err_${TAG}.log; python3 tools/show_steps.py $O/bench_line_${TAG}.json ;;
    bench_legs=*) LG=${s#bench_legs=}; timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --legs $LG > $O/bench_${LG//,/_}_${TAG}.json 2> $O/bench_err_${TAG}.log; echo "bench rc=$?"; tail -3 $O/bench_err_${TAG}.log; python3 tools/show_steps.py $O/bench_${LG//,/_}_${TAG}.json ;;
    head1) timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/fresh_${TAG}.json 2> $O/fresh_${TAG}.err; echo "head rc=$?"; python3 tools/show_steps.py $O/fresh_${TAG}.json ;;
    head3) for k in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/head_${TAG}_$k.json 2> $O/head_${TAG}_$k.err; echo "head $k rc=$?"; python3 tools/show_steps.py $O/head_${TAG}_$k.json; done ;;
    shape=*) SH=${s#shape=}; for o in "" "--option rdf_disp=0"; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --shape $SH $o > $O/shape_${SH}_${TAG}.json 2> $O/shape_err.log; echo "shape $SH [$o] rc=$?"; python3 tools/show_steps.py $O/shape_${SH}_${TAG}.json; done ;;
    bench_c4) timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_n1_${TAG}.json 2> $O/bench_c4_n1_err.log; echo "c4 rc=$?"; tail -3 $O/bench_c4_n1_err.log ;;
    bench_c4_2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --workload c4 --steps 5 --warmup 1 > $O/bench_c4_gloo2_${TAG}.json 2> $O/bench_c4_gloo2_err.log; echo "c4x2 rc=$?"; tail -3 $O/bench_c4_gloo2_err.log ;;
    bench_c4_rccl1) MDHIP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_rccl1_${TAG}.json 2> $O/bench_c4_rccl1_err.log; echo "c4 rccl1 rc=$?"; tail -3 $O/bench_c4_rccl1_err.log ;;
    bench2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2_gloo_${TAG}.json 2> $O/bench_gpus2_err.log; echo "bench2 rc=$?"; tail -3 $O/bench_gpus2_err.log ;;
    pmc_c2) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c2 C2 > $O/pmc_c2.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_c2.log ;;
    pmc_c1) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c1 C1 --shape C1 > $O/pmc_c1.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_c1.log ;;
    pmc_c1alt) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c1alt C1alt --shape C1alt > $O/pmc_c1alt.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_c1alt.log ;;
    pmc_c1full) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c1full C1full --shape C1full > $O/pmc_c1full.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_c1full.log ;;
    pmc_c2_f64) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c2_f64 C2 --option rdf_pk=0 > $O/pmc_c2_f64.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c3 C3 --scaling strong > $O/pmc_c3.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_cn) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c3_cn C3/cn --scaling strong --op cn > $O/pmc_c3_cn.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_both) timeout -k 10 900 bash tools/pmc.sh ${ROUND}_c3_both C3/rdf_cn --scaling strong --op rdf_cn > $O/pmc_c3_both.log 2>&1; echo "pmc rc=$?" ;;
    pmc2) timeout -k 10 1150 bash tools/pmc_secondary.sh ${ROUND} > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"; tail -3 $O/pmc2.log ;;
    soaks) ( timeout -k 10 500 python tests/bench/soak_pk.py 2500 5 oracle 2>&1 | tail -2; timeout -k 10 400 python tests/bench/soak_cn.py 1000 2>&1 | tail -2; timeout -k 10 300 python tests/bench/soak_cull.py 1000 2>&1 | tail -2; timeout -k 10 400 python tests/bench/soak_lag.py 300 2>&1 | tail -2; timeout -k 10 600 python tests/bench/soak_lag_long.py 120 2>&1 | tail -3; timeout -k 10 600 python tests/bench/soak_lag_short.py 600 2>&1 | tail -2; timeout -k 10 600 python tests/bench/soak_lag_ends.py 150 2>&1 | tail -2; timeout -k 10 200 python tests/bench/soak_fft.py 2>&1 | tail -2 ) > $O/${ROUND}_soaks.txt 2>&1; cat $O/${ROUND}_soaks.txt ;;
    stats) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${ROUND}_bench -- python3 $R/bench.py --no-cpu-baseline --no-legs > $O/bench_line_rocprof.json 2> $O/rocprof_err.log); echo "stats rc=$?" ;;
    stats_legs) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ${ROUND}_bench_legs -- python3 $R/bench.py --no-cpu-baseline > $O/bench_line_rocprof_legs.json 2> $O/rocprof_legs_err.log); echo "stats_legs rc=$?"; tail -2 $O/rocprof_legs_err.log ;;
    ab_lag) timeout -k 10 400 python tools/ab_libs_lag.py $LP $L 2>&1 | grep -v amdgpu > $O/${ROUND}_ab_lag_${TAG}.txt; cat $O/${ROUND}_ab_lag_${TAG}.txt ;;
    ab_pair) timeout -k 10 400 python tools/ab_libs.py $LP $L 2>&1 | grep -v amdgpu > $O/${ROUND}_ab_pair_${TAG}.txt; cat $O/${ROUND}_ab_pair_${TAG}.txt ;;
    w12_check) timeout -k 10 400 python tools/w12.py check 2>&1 | grep -v amdgpu > $O/${ROUND}_w12_check.txt; tail -12 $O/${ROUND}_w12_check.txt ;;
    w12_exp=*) timeout -k 10 400 python tools/w12.py exp $L ${s#w12_exp=} $L 2>&1 | grep -v amdgpu > $O/${ROUND}_w12_exp_${TAG}.txt; cat $O/${ROUND}_w12_exp_${TAG}.txt ;;
    lag_sizes) timeout -k 10 900 python tools/lag_sizes.py 2>&1 | grep -v amdgpu > $O/${ROUND}_lag_sizes.txt; cat $O/${ROUND}_lag_sizes.txt ;;
    c4_shard) ( for rep in 1 2 3; do for n in 8 4; do for o in two one; do C4_ORDER=$o timeout -k 10 300 python tools/c4_shard_cost.py $n 2>&1 | grep -v amdgpu | tail -3; done; done; done ) > $O/${ROUND}_c4_shard.txt 2>&1; cat $O/${ROUND}_c4_shard.txt ;;
    *) if [ -f "$s" ]; then timeout -k 10 600 python "$s" > $O/$(basename $s .py)_${TAG}.txt 2>&1; echo "$s rc=$?"; tail -40 $O/$(basename $s .py)_${TAG}.txt; else echo "unknown step $s"; fi ;;
  esac
done
