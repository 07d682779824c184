mkdir -p gpurun_out/r3c; O=$PWD/gpurun_out/r3c
for m in cluster3 cluster2; do
  JQC_SPATIAL_SORT=$m timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-grid > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r3c/bench_$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['realistic_density']['ms_per_step'])
P
done
JQC_SPATIAL_SORT=cluster2 timeout 600 python tools/big_check.py 0166-ionic-bulky-valinomycin def2-svp 2>&1 | grep -v amdgpu | tail -3
