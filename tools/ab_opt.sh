# tools/ab_opt.sh KEY V1 V2 [test-filter] — headline bench (no legs) with context option KEY at two values, twice each
set -u
O=gpurun_out; K=$1; A=$2; B=$3; T=${4:-"rdf or pair or cn or async"}
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$T" > $O/gpu_tests_ab.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/gpu_tests_ab.log; [ $rc -eq 0 ] || exit 1
for k in 1 2; do
for v in $A $B; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --option $K=$v > $O/ab_${K}_$v.json 2> $O/ab_${K}_$v.err || exit 1
  python3 -c "
import json;d=json.load(open('$O/ab_${K}_$v.json'));print('$K=$v ms_per_step %.4f value %.4e kernel_ms %.4f prepass %.4f'%(d['ms_per_step'],d['value'],d['roofline'].get('launch_ms',0) or 0,d['roofline'].get('prepass_ms_per_step',0)))"
done; done
