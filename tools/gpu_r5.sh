#!/bin/bash
# tools/gpu_r5.sh [steps...] — round-5 GPU-box sequences; every step writes under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${TAG:-x}
L=mdproptools_amd/libmdhip.so
L4=tools/_bin/libmdhip_r4.so
for s in "$@"; do
  echo "== $s $(date +%T)"
  case $s in
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/gpu_tests.log; [ $rc -eq 0 ] || exit 1 ;;
    tests_new) timeout -k 10 900 python -m pytest -m gpu -x -q tests/test_gpu_async.py tests/test_gpu_hardening.py "tests/test_gpu_parity.py::test_cumtrapz_golden_and_sizes" > $O/gpu_tests_new.log 2>&1; rc=$?; echo "tests_new rc=$rc"; tail -15 $O/gpu_tests_new.log; [ $rc -eq 0 ] || exit 1 ;;
    ab_scan) timeout -k 10 300 python tools/ab_libs_scan.py $L4 $L 2>&1 | grep -v amdgpu > $O/r05_ab_scan.txt; cat $O/r05_ab_scan.txt ;;
    bench) timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_${TAG}.json 2> $O/bench_err_${TAG}.log; echo "bench rc=$?"; tail -3 $O/bench_err_${TAG}.log; python3 tools/show_steps.py $O/bench_line_${TAG}.json ;;
    head1) timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/fresh_${TAG}.json 2> $O/fresh_${TAG}.err; echo "head rc=$?"; python3 tools/show_steps.py $O/fresh_${TAG}.json ;;
    bench_c4) timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_n1_${TAG}.json 2> $O/bench_c4_n1_err.log; echo "c4 rc=$?"; tail -3 $O/bench_c4_n1_err.log ;;
    ab_lag) timeout -k 10 400 python tools/ab_libs_lag.py $L4 $L 2>&1 | grep -v amdgpu > $O/r05_ab_lag_${TAG}.txt; cat $O/r05_ab_lag_${TAG}.txt ;;
    pmc_c2) timeout -k 10 900 bash tools/pmc.sh r05_c2 C2 > $O/pmc_c2.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_c2.log ;;
    pmc_c2_f64) timeout -k 10 900 bash tools/pmc.sh r05_c2_f64 C2 --option rdf_pk=0 > $O/pmc_c2_f64.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3) timeout -k 10 900 bash tools/pmc.sh r05_c3 C3 --scaling strong > $O/pmc_c3.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_cn) timeout -k 10 900 bash tools/pmc.sh r05_c3_cn C3/cn --scaling strong --op cn > $O/pmc_c3_cn.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_both) timeout -k 10 900 bash tools/pmc.sh r05_c3_both C3/rdf_cn --scaling strong --op rdf_cn > $O/pmc_c3_both.log 2>&1; echo "pmc rc=$?" ;;
    pmc2) timeout -k 10 1150 bash tools/pmc_secondary.sh r05 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"; tail -3 $O/pmc2.log ;;
    head3) for k in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/head_${TAG}_$k.json 2> $O/head_${TAG}_$k.err; echo "head $k rc=$?"; python3 tools/show_steps.py $O/head_${TAG}_$k.json; done ;;
    bench_c4_2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --workload c4 --steps 5 --warmup 1 > $O/bench_c4_gloo2_${TAG}.json 2> $O/bench_c4_gloo2_err.log; echo "c4x2 rc=$?"; tail -3 $O/bench_c4_gloo2_err.log ;;
    bench_c4_rccl1) MDHIP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_rccl1_${TAG}.json 2> $O/bench_c4_rccl1_err.log; echo "c4 rccl1 rc=$?"; tail -3 $O/bench_c4_rccl1_err.log ;;
    bench2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2_gloo_${TAG}.json 2> $O/bench_gpus2_err.log; echo "bench2 rc=$?"; tail -3 $O/bench_gpus2_err.log ;;
    soaks) ( timeout -k 10 500 python tests/bench/soak_pk.py 2500 5 oracle 2>&1 | tail -2; timeout -k 10 400 python tests/bench/soak_cn.py 1000 2>&1 | tail -2; timeout -k 10 300 python tests/bench/soak_cull.py 1000 2>&1 | tail -2; timeout -k 10 400 python tests/bench/soak_lag.py 300 2>&1 | tail -2; timeout -k 10 200 python tests/bench/soak_fft.py 2>&1 | tail -2 ) > $O/r05_soaks.txt 2>&1; cat $O/r05_soaks.txt ;;
    stats) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r05_bench -- python3 $R/bench.py --no-cpu-baseline --no-legs > $O/bench_line_rocprof.json 2> $O/rocprof_err.log); echo "stats rc=$?" ;;
    stats_legs) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r05_bench_legs -- python3 $R/bench.py --no-cpu-baseline > $O/bench_line_rocprof_legs.json 2> $O/rocprof_legs_err.log); echo "stats_legs rc=$?"; tail -2 $O/rocprof_legs_err.log ;;
    w12_check) timeout -k 10 400 python tools/w12_check.py 2>&1 | grep -v amdgpu > $O/r05_w12_check.txt; tail -12 $O/r05_w12_check.txt ;;
    w12_exp) timeout -k 10 400 python tools/w12_exp.py $L tools/_bin/libmdhip_w12e256.so tools/_bin/libmdhip_w12e512.so tools/_bin/libmdhip_w12e128.so tools/_bin/libmdhip_w12e1024.so $L 2>&1 | grep -v amdgpu > $O/r05_w12_exp.txt; cat $O/r05_w12_exp.txt ;;
    fft_mid) timeout -k 10 400 python tools/fft_mid_check.py 2>&1 | grep -v amdgpu > $O/r05_fft_mid.txt; cat $O/r05_fft_mid.txt ;;
    ab_small_copy) bash tools/ab_opt.sh small_copy 1 0 "golden" > $O/r05_ab_small_copy.txt 2>&1; cat $O/r05_ab_small_copy.txt ;;
    ab_sort) bash tools/ab_opt.sh rdf_sort -1 3 "golden" > $O/r05_ab_sort.txt 2>&1; cat $O/r05_ab_sort.txt ;;
    c4_shard) ( for rep in 1 2 3; do for n in 8 4; do for o in two one; do C4_ORDER=$o timeout -k 10 300 python tools/c4_shard_cost.py $n 2>&1 | grep -v amdgpu | tail -3; done; done; done ) > $O/r05_c4_shard.txt 2>&1; cat $O/r05_c4_shard.txt ;;
    ab_msd) timeout -k 10 400 python tools/ab_libs_msd.py $L4 $L 2>&1 | grep -v amdgpu > $O/r05_ab_msd.txt; cat $O/r05_ab_msd.txt ;;
    ubench_hbm) timeout -k 10 300 tools/_bin/ubench_hbm $O/r05_ubench_hbm.json > $O/r05_ubench_hbm.txt 2>&1; cat $O/r05_ubench_hbm.txt ;;
    *) if [ -f "$s" ]; then timeout -k 10 600 python "$s" > $O/$(basename $s .py)_${TAG}.txt 2>&1; echo "$s rc=$?"; tail -40 $O/$(basename $s .py)_${TAG}.txt; else echo "unknown step $s"; fi ;;
  esac
done
