mkdir -p gpurun_out/r2j; O=$PWD/gpurun_out/r2j
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r2j/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d.get('grid_path')))
P
