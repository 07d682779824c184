mkdir -p gpurun_out/r3g; O=$PWD/gpurun_out/r3g
for o in rows kdtree; do JQC_GRID_ORDER=$o timeout 600 python tools/dft_host_time.py 2>&1 | grep -v amdgpu | grep "GGA:" | cut -c1-210 | sed "s/^/$o: /"; done
for cfg in "16 4096" "32 4096" "8 4096" "16 8192" "16 2048" "64 4096"; do
  set -- $cfg
  JQC_KCHUNK_MAX=$1 JQC_TARGET_WGS=$2 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-grid > $O/b_$1_$2.json 2>/dev/null
  python - <<P
import json
d=json.loads(open('gpurun_out/r3g/b_$1_$2.json').read().strip().splitlines()[-1]); print('kchunk $1 target $2', round(d['ms_per_step'],1), round(d['realistic_density']['ms_per_step'],1))
P
done
