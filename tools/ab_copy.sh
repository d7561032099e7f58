set -u
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rdf or pair or cn or async or cumtrapz or result" > $O/gpu_tests_copy.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/gpu_tests_copy.log; [ $rc -eq 0 ] || exit 1
for k in 1 2; do
for v in 1 0; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --option small_copy=$v > $O/ab_copy_$v.json 2> $O/ab_copy_$v.err || exit 1
  python3 -c "
import json;d=json.load(open('$O/ab_copy_$v.json'));print('small_copy=$v ms_per_step %.4f value %.4e kernel_ms %.4f prepass %.4f'%(d['ms_per_step'],d['value'],d['roofline'].get('launch_ms',0) or 0,d['roofline'].get('prepass_ms_per_step',0)))"
done; done
