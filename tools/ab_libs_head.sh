# tools/ab_libs_head.sh LIB... — headline bench (no legs) through several builds of the library (MDHIP_LIB)
O=gpurun_out
for k in 1 2; do for L in "$@"; do
  MDHIP_LIB=$L timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-legs --no-cpu-baseline > $O/ab_lib.json 2> $O/ab_lib.err || { tail -3 $O/ab_lib.err; exit 1; }
  python3 -c "
import json;d=json.load(open('$O/ab_lib.json'));print('%-40s ms_per_step %.4f value %.4e kernel_ms %.4f prepass %.4f'%('$L'.split('/')[-1],d['ms_per_step'],d['value'],d['roofline'].get('launch_ms',0) or 0,d['roofline'].get('prepass_ms_per_step',0)))"
done; done
