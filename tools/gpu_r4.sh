#!/bin/bash
# tools/gpu_r4.sh [steps...] — round-4 GPU-box sequences; every step writes under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
TAG=${TAG:-x}
for s in "$@"; do
  echo "== $s $(date +%T)"
  case $s in
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/gpu_tests.log; [ $rc -eq 0 ] || exit 1 ;;
    # the driver's own command, three times on one lease (the first is the cold one), headline only
    head3) for k in 1 2 3; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/head_${TAG}_$k.json 2> $O/head_${TAG}_$k.err; echo "head $k rc=$?"; python3 tools/show_steps.py $O/head_${TAG}_$k.json; done ;;
    # the same under a concurrent rocm-smi poll (the driver samples the GPU every ~5 s while bench.py runs; here every 0.3 s)
    head_smi) ( while true; do rocm-smi --showuse --showmemuse --showpower --json > /dev/null 2>&1; sleep 0.3; done ) & SMI=$!
      for k in 1 2 3 4; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/headsmi_${TAG}_$k.json 2> $O/headsmi_${TAG}_$k.err; echo "head_smi $k rc=$?"; python3 tools/show_steps.py $O/headsmi_${TAG}_$k.json; done
      kill $SMI; wait $SMI 2>/dev/null; which rocm-smi amd-smi; ( time rocm-smi --showuse --json ) 2>&1 | tail -5 ;;
    head1) timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/fresh_${TAG}.json 2> $O/fresh_${TAG}.err; echo "head rc=$?"; python3 tools/show_steps.py $O/fresh_${TAG}.json ;;
    bench) timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_${TAG}.json 2> $O/bench_err_${TAG}.log; echo "bench rc=$?"; tail -3 $O/bench_err_${TAG}.log; python3 tools/show_steps.py $O/bench_line_${TAG}.json ;;
    tests_async) timeout -k 10 900 python -m pytest tests/test_gpu_async.py -m gpu -x -q > $O/gpu_tests_async.log 2>&1; rc=$?; echo "tests_async rc=$rc"; tail -30 $O/gpu_tests_async.log; [ $rc -eq 0 ] || exit 1 ;;
    bench_c4) timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_n1_${TAG}.json 2> $O/bench_c4_n1_err.log; echo "c4 rc=$?"; tail -3 $O/bench_c4_n1_err.log; python3 -c "
import json,sys; d=json.loads(open('$O/bench_c4_n1_${TAG}.json').read().strip().splitlines()[-1]); m=d['msd']; print('c4 value %.4g ms/step %.3f kernels %.3f ratio %.3f' % (d['value'], d['ms_per_step'], m['kernel_ms_per_step'], m['step_over_kernels'])); print(m['step_ms']['raw']); print({k:m[k]['kernel_ms'] for k in ('single_origin','fixed_lag_tao4','full_lag')})" ;;
    bench_c4_2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --workload c4 --steps 5 --warmup 1 > $O/bench_c4_gloo2_${TAG}.json 2> $O/bench_c4_gloo2_err.log; echo "c4x2 rc=$?"; tail -3 $O/bench_c4_gloo2_err.log ;;
    bench_c4_rccl1) MDHIP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4_rccl1_${TAG}.json 2> $O/bench_c4_rccl1_err.log; echo "c4 rccl1 rc=$?"; tail -3 $O/bench_c4_rccl1_err.log; python3 -c "
import json,sys; d=json.loads(open('$O/bench_c4_rccl1_${TAG}.json').read().strip().splitlines()[-1]); m=d['msd']; print('c4 rccl1 value %.4g ms/step %.3f kernels %.3f ratio %.3f' % (d['value'], d['ms_per_step'], m['kernel_ms_per_step'], m['step_over_kernels'])); print(m['step_ms']['raw'])" ;;
    pmc_c2) timeout -k 10 900 bash tools/pmc.sh r04_c2 C2 > $O/pmc_c2.log 2>&1; echo "pmc rc=$?"; tail -3 $O/pmc_c2.log ;;
    pmc_c2_f64) timeout -k 10 900 bash tools/pmc.sh r04_c2_f64 C2 --option rdf_pk=0 > $O/pmc_c2_f64.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3) timeout -k 10 900 bash tools/pmc.sh r04_c3 C3 --scaling strong > $O/pmc_c3.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_cn) timeout -k 10 900 bash tools/pmc.sh r04_c3_cn C3/cn --scaling strong --op cn > $O/pmc_c3_cn.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_both) timeout -k 10 900 bash tools/pmc.sh r04_c3_both C3/rdf_cn --scaling strong --op rdf_cn > $O/pmc_c3_both.log 2>&1; echo "pmc rc=$?" ;;
    pmc2) timeout -k 10 1100 bash tools/pmc_secondary.sh r04 > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"; tail -5 $O/pmc2.log ;;
    evidence) L=mdproptools_amd/libmdhip.so
      timeout -k 10 200 python tools/ab_libs_scan.py tools/_bin/libmdhip_scan3.so $L 2>&1 | grep -v amdgpu > $O/r04_ab_scan.txt
      timeout -k 10 300 python tools/ab_libs_lag.py $L:lag_direct=0 $L:lag_direct=1 2>&1 | grep -v amdgpu > $O/r04_ab_lag_direct.txt
      ( for w in C3 C2; do for op in cn rdf; do timeout -k 10 250 python tools/ab_pair.py rdf_jsplit=0,4,2,1 $w $op 2>&1 | grep -v amdgpu; done; done ) > $O/r04_ab_cn_slices.txt
      timeout -k 10 120 python tools/gk_diag.py 2>&1 | grep -v amdgpu > $O/r04_gk_diag.txt
      timeout -k 10 120 python tools/pin_cost.py 2>&1 | grep -v amdgpu > $O/r04_pin_cost.txt
      timeout -k 10 120 python tools/h2d_pipe.py 16 2>&1 | grep -v amdgpu > $O/r04_h2d_pipe.txt
      ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_h2d -- python3 $R/tools/h2d_pipe.py 8 > /dev/null 2>&1 ); python3 tools/trace_overlap.py $O/trace_h2d >> $O/r04_h2d_pipe.txt
      for f in r04_ab_scan r04_ab_lag_direct r04_gk_diag; do tail -n 3 $O/$f.txt; done ;;
    stats) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r04_bench -- python3 $R/bench.py --no-cpu-baseline --no-legs > $O/bench_line_rocprof.json 2> $O/rocprof_err.log); echo "stats rc=$?" ;;
    stats_legs) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r04_bench_legs -- python3 $R/bench.py --no-cpu-baseline > $O/bench_line_rocprof_legs.json 2> $O/rocprof_legs_err.log); echo "stats_legs rc=$?"; tail -2 $O/rocprof_legs_err.log ;;
    overhead) timeout -k 10 300 python tools/call_overhead.py 2>&1 | grep -v amdgpu | tail -8 ;;
    *) echo "unknown step $s" ;;
  esac
done
