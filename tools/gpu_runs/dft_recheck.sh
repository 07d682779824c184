mkdir -p gpurun_out/r3h; O=$PWD/gpurun_out/r3h
timeout 1200 python -m pytest tests/test_dft_gpu.py tests/test_dft_fullsize_gpu.py tests/test_grad_gpu.py tests/test_boundary_gpu.py -q -m gpu --timeout=900 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python - <<'P'
import json
d=json.loads(open('gpurun_out/r3h/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['traffic'], {k:round(v['ms'],2) for k,v in d['grid_path'].items() if isinstance(v,dict)})
P
