#!/bin/bash
# tools/gpu_r3.sh [steps...] — round-3 GPU-box sequences; every step writes under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
for s in "$@"; do
  echo "== $s $(date +%T)"
  case $s in
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/gpu_tests.log; [ $rc -eq 0 ] || exit 1 ;;
    tests_dist) timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -m gpu -x -q -k "two_ranks or sharded or bench" > $O/gpu_tests_dist.log 2>&1; rc=$?; echo "tests_dist rc=$rc"; tail -15 $O/gpu_tests_dist.log; [ $rc -eq 0 ] || exit 1 ;;
    bench) timeout -k 10 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; tail -3 $O/bench_err.log ;;
    bench_head) timeout -k 10 300 python bench.py --no-legs > $O/bench_head.json 2> $O/bench_head_err.log; echo "bench rc=$?"; tail -3 $O/bench_head_err.log ;;
    bench_c4) timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_n1.json 2> $O/bench_c4_n1_err.log; echo "c4 rc=$?"; tail -3 $O/bench_c4_n1_err.log; cat $O/bench_c4_n1.json ;;
    bench_c4_2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --workload c4 --steps 5 --warmup 1 > $O/bench_c4_gloo2.json 2> $O/bench_c4_gloo2_err.log; echo "c4x2 rc=$?"; tail -3 $O/bench_c4_gloo2_err.log; cat $O/bench_c4_gloo2.json ;;
    bench2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_err.log; echo "bench2 rc=$?"; tail -3 $O/bench_gpus2_err.log; cat $O/bench_gpus2_gloo.json ;;
    pmc2) timeout -k 10 1100 bash tools/pmc_secondary.sh r03 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"; tail -5 $O/pmc2.log ;;
    stats_legs) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03_bench_legs -- python3 $R/bench.py --no-cpu-baseline > $O/bench_line_rocprof_legs.json 2> $O/rocprof_legs_err.log); echo "stats_legs rc=$?"; tail -2 $O/rocprof_legs_err.log ;;
    stats) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03_bench -- python3 $R/bench.py --no-cpu-baseline --no-legs > $O/bench_line_rocprof.json 2> $O/rocprof_err.log); echo "stats rc=$?" ;;
    mem) free -g; nproc; df -h /tmp | tail -1 ;;
    tests_dropin) timeout -k 10 900 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q > $O/gpu_tests_dropin.log 2>&1; rc=$?; echo "tests_dropin rc=$rc"; tail -5 $O/gpu_tests_dropin.log; [ $rc -eq 0 ] || exit 1 ;;
    e2e_c2) timeout -k 10 600 python tests/bench/bench_e2e.py 10000 200 > $O/bench_e2e_c2.txt 2> $O/bench_e2e_err.log; echo "e2e_c2 rc=$?"; tail -2 $O/bench_e2e_err.log; cat $O/bench_e2e_c2.txt ;;
    e2e_c3) timeout -k 10 1100 python tests/bench/bench_e2e.py 100000 1000 > $O/bench_e2e_c3_full.txt 2> $O/bench_e2e_c3_err.log; echo "e2e_c3 rc=$?"; tail -2 $O/bench_e2e_c3_err.log; cat $O/bench_e2e_c3_full.txt ;;
    e2e_c4) timeout -k 10 1100 python tests/bench/bench_e2e_msd.py 50000 5000 allatom > $O/bench_e2e_c4_full.txt 2> $O/bench_e2e_c4_err.log; echo "e2e_c4 rc=$?"; tail -2 $O/bench_e2e_c4_err.log; cat $O/bench_e2e_c4_full.txt ;;
    e2e_c4com) timeout -k 10 1100 python tests/bench/bench_e2e_msd.py 50000 5000 com > $O/bench_e2e_c4com_full.txt 2> $O/bench_e2e_c4com_err.log; echo "e2e_c4com rc=$?"; tail -2 $O/bench_e2e_c4com_err.log; cat $O/bench_e2e_c4com_full.txt ;;
    tests_fft) timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "xcorr or fft or lag or c5 or c4 or viscos or conduct" > $O/gpu_tests_fft.log 2>&1; rc=$?; echo "tests_fft rc=$rc"; tail -5 $O/gpu_tests_fft.log; [ $rc -eq 0 ] || exit 1 ;;
    soak_fft) timeout -k 10 900 python tests/bench/soak_fft.py > $O/soak_fft.log 2>&1; echo "soak_fft rc=$?"; tail -4 $O/soak_fft.log ;;
    sec_fft) for w in acf_fft lag_fft; do timeout -k 10 300 python tools/run_secondary.py $w 5 2>&1 | tail -3; done ;;
    tests_pair) timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not bench and not two_ranks and not c4 and not c5 and not xcorr and not msd" > $O/gpu_tests_pair.log 2>&1; rc=$?; echo "tests_pair rc=$rc"; tail -4 $O/gpu_tests_pair.log; [ $rc -eq 0 ] || exit 1 ;;
    soak_pk) timeout -k 10 900 python tests/bench/soak_pk.py 1500 31 oracle > $O/soak_pk.log 2>&1; echo "soak_pk rc=$?"; tail -3 $O/soak_pk.log ;;
    soak_cn) timeout -k 10 900 python tests/bench/soak_cn.py 800 5 oracle > $O/soak_cn.log 2>&1; echo "soak_cn rc=$?"; tail -3 $O/soak_cn.log ;;
    soak_cull) timeout -k 10 900 python tests/bench/soak_cull.py 600 > $O/soak_cull.log 2>&1; echo "soak_cull rc=$?"; tail -3 $O/soak_cull.log ;;
    lag_sizes) for l in tools/_bin/libmdhip_f3m9.so tools/_bin/libmdhip_f2a.so mdproptools_amd/libmdhip.so; do timeout -k 10 300 python tools/check_lag_sizes.py $l 2>&1 | grep -v amdgpu | grep 'BAD\|bad' ; done ;;
    pmc_lag) timeout -k 10 600 bash tools/pmc_secondary.sh r03c lag_fft > $O/pmc_lag.log 2>&1; echo "pmc_lag rc=$?"; grep -A30 'msd_power_lds' $O/pmc2/r03c_secondary_summary.txt | head -34 ;;
    tests_seg) timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "com or flux or segment or conduct or diffus or c4 or dropin" > $O/gpu_tests_seg.log 2>&1; rc=$?; echo "tests_seg rc=$rc"; tail -5 $O/gpu_tests_seg.log; [ $rc -eq 0 ] || exit 1 ;;
    tests_rccl1) timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "rccl" > $O/gpu_tests_rccl1.log 2>&1; rc=$?; echo "tests_rccl1 rc=$rc"; tail -15 $O/gpu_tests_rccl1.log; [ $rc -eq 0 ] || exit 1 ;;
    overhead) timeout -k 10 300 python tools/call_overhead.py 2>&1 | grep -v amdgpu | tail -8 ;;
    ab_scan) timeout -k 10 300 python tools/ab_libs_scan.py $(ls tools/_bin/libmdhip_scan*.so) mdproptools_amd/libmdhip.so 2>&1 | grep -v amdgpu | tail -20 ;;
    ab_fft) L=mdproptools_amd/libmdhip.so; for n in 1000000 300000; do timeout -k 10 300 python tools/ab_libs_xcorr.py $L:fft_logr=8 $L:fft_specfuse=0 $L fft $n 2>&1 | grep -v amdgpu | tail -3; done ;;
    ab_seg) timeout -k 10 300 python tools/ab_libs_seg.py $(ls tools/_bin/libmdhip_seg*.so) mdproptools_amd/libmdhip.so 2>&1 | grep -v amdgpu | tail -16 ;;
    ab_lag) timeout -k 10 300 python tools/ab_libs_lag.py mdproptools_amd/libmdhip.so:lag_fft_kernel=1 mdproptools_amd/libmdhip.so:lag_fft_kernel=2 2>&1 | grep -v amdgpu | tail -12 ;;
    ab_pair) for w in C2 C3; do for op in rdf cn rdf_cn; do timeout -k 10 300 python tools/ab_libs.py tools/_bin/libmdhip_cur.so mdproptools_amd/libmdhip.so $w $op 2>&1 | grep -v amdgpu | tail -2; done; done ;;
    pmc_c2) timeout -k 10 900 bash tools/pmc.sh r03_c2 C2 > $O/pmc_c2.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c2_f64) timeout -k 10 900 bash tools/pmc.sh r03_c2_f64 C2 --option rdf_pk=0 > $O/pmc_c2_f64.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3) timeout -k 10 900 bash tools/pmc.sh r03_c3 C3 --scaling strong > $O/pmc_c3.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_cn) timeout -k 10 900 bash tools/pmc.sh r03_c3_cn C3/cn --scaling strong --op cn > $O/pmc_c3_cn.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_both) timeout -k 10 900 bash tools/pmc.sh r03_c3_both C3/rdf_cn --scaling strong --op rdf_cn > $O/pmc_c3_both.log 2>&1; echo "pmc rc=$?" ;;
    *) echo "unknown step $s" ;;
  esac
done
