#!/bin/bash
# tools/gpu_round.sh [steps...] — the GPU-box sequence of a development round; every step writes under gpurun_out/.
# steps: ubench tests bench bench2 pmc_c2 pmc_c3 stats
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
for s in "$@"; do
  echo "== $s $(date +%T)"
  case $s in
    ubench) timeout -k 10 300 ./tools/_bin/ubench_valu $O/ubench_valu.json > $O/ubench_valu.txt 2>&1 || exit 1 ;;
    tests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/gpu_tests.log ;;
    bench) timeout -k 10 900 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; tail -3 $O/bench_err.log ;;
    bench_head) timeout -k 10 300 python bench.py --no-legs > $O/bench_head.json 2> $O/bench_head_err.log; echo "bench rc=$?"; tail -3 $O/bench_head_err.log ;;
    bench2) MDHIP_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 1 > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_err.log; echo "bench2 rc=$?"; tail -3 $O/bench_gpus2_err.log ;;
    bench_strong) timeout -k 10 600 python bench.py --scaling strong --steps 3 --warmup 1 > $O/bench_strong.json 2> $O/bench_strong_err.log; echo "strong rc=$?"; tail -3 $O/bench_strong_err.log ;;
    pmc_c2) timeout -k 10 900 bash tools/pmc.sh r02_c2 C2 > $O/pmc_c2.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c2_f64) timeout -k 10 900 bash tools/pmc.sh r02_c2_f64 C2 --option rdf_pk=0 > $O/pmc_c2_f64.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3) timeout -k 10 900 bash tools/pmc.sh r02_c3 C3 --scaling strong > $O/pmc_c3.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_cn) timeout -k 10 900 bash tools/pmc.sh r02_c3_cn C3/cn --scaling strong --op cn > $O/pmc_c3_cn.log 2>&1; echo "pmc rc=$?" ;;
    pmc_c3_both) timeout -k 10 900 bash tools/pmc.sh r02_c3_both C3/rdf_cn --scaling strong --op rdf_cn > $O/pmc_c3_both.log 2>&1; echo "pmc rc=$?" ;;
    stats) (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02_bench -- python3 $R/bench.py --no-cpu-baseline --no-legs > $O/bench_line_rocprof.json 2> $O/rocprof_err.log); echo "stats rc=$?" ;;
    *) echo "unknown step $s" ;;
  esac
done
