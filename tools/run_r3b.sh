mkdir -p gpurun_out/r3b; O=$PWD/gpurun_out/r3b
for cfg in "cluster2 2.5" "cluster_exp 2.5" "cluster_exp 6" "cluster2_exp 2.5"; do
  set -- $cfg
  JQC_SPATIAL_SORT=$1 JQC_EXP_CLASS=$2 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-grid > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r3b/bench_$1_$2.json').read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['realistic_density']['ms_per_step'])
P
done
