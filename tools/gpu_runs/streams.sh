mkdir -p gpurun_out/r3n; O=$PWD/gpurun_out/r3n
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3n/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_path']['frac'], d['realistic_density']['ms_per_step'])
P
timeout 300 python bench.py --workload benzene --no-cpu-baseline --no-grid 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('benzene', d['ms_per_step'])"
timeout 1200 python -m pytest tests/test_jk_gpu.py tests/test_jk_fullsize_gpu.py -q -m gpu --timeout=900 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
