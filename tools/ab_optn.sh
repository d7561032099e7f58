# tools/ab_optn.sh KEY V1 V2 ... — the driver's command (no legs) with context option KEY at every value, twice, one lease
O=gpurun_out; K=$1; shift
for k in 1 2; do for v in "$@"; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline --option $K=$v > $O/ab_${K}_$v.json 2> $O/ab_${K}_$v.err || { tail -3 $O/ab_${K}_$v.err; exit 1; }
  python3 -c "
import json;d=json.load(open('$O/ab_${K}_$v.json'));print('$K=$v ms_per_step %.4f value %.4e kernel_ms %.4f prepass %.4f'%(d['ms_per_step'],d['value'],d['roofline'].get('launch_ms',0) or 0,d['roofline'].get('prepass_ms_per_step',0)))"
done; done
