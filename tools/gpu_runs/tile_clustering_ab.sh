mkdir -p gpurun_out/r3a; O=$PWD/gpurun_out/r3a
for m in morton cluster; do
  JQC_SPATIAL_SORT=$m timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-grid > $O/bench_$m.json 2> $O/bench_$m.err
  python - <<P
import json
d=json.loads(open('gpurun_out/r3a/bench_$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['whole_path']['serial_kernel_sum_ms'], d['realistic_density']['ms_per_step'])
P
  JQC_SPATIAL_SORT=$m timeout 600 python bench.py --workload benzene --no-cpu-baseline --no-grid > $O/benzene_$m.json 2>/dev/null
  python - <<P
import json
d=json.loads(open('gpurun_out/r3a/benzene_$m.json').read().strip().splitlines()[-1]); print('$m benzene', d['ms_per_step'])
P
done
JQC_SPATIAL_SORT=cluster timeout 900 python tools/class_profile.py 0112-elongated-nitrogenous > $O/class_profile_cluster.txt 2>&1
