set -x
mkdir -p gpurun_out/r2h; O=$PWD/gpurun_out/r2h; R=$PWD
timeout 1200 python -m pytest tests/test_dft_gpu.py tests/test_dft_fullsize_gpu.py -q -m gpu --timeout=600 -v > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python tools/dft_bench.py 0112-elongated-nitrogenous def2-tzvpp 344064 > $O/dft_bench_tzvpp.log 2>&1
timeout 600 python tools/dft_bench.py 0112-elongated-nitrogenous def2-svp 400000 > $O/dft_bench_svp.log 2>&1
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_112_short.json 2> $O/bench_112_short.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dft -o d -- python3 $R/tools/dft_host_time.py > $O/dft_host_time.log 2>&1
cd $R; grep -v amdgpu $O/dft_host_time.log | grep "GGA:"; grep -v amdgpu $O/dft_bench_tzvpp.log; grep -v amdgpu $O/dft_bench_svp.log; rm -f $O/dft/d_kernel_trace.csv
python - <<'P'
import json
d=json.loads(open('gpurun_out/r2h/bench_112_short.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('grid_path'))
P
