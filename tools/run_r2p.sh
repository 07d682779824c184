mkdir -p gpurun_out/r2p; O=$PWD/gpurun_out/r2p
timeout 2400 python -m pytest tests/test_grad_gpu.py tests/test_dft_gpu.py -q -m gpu --timeout=1200 > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 1500 python tools/grad_bench.py 0112-elongated-nitrogenous def2-tzvpp > $O/grad_bench_tzvpp.log 2>&1; grep -v amdgpu $O/grad_bench_tzvpp.log
timeout 600 python tools/grad_bench.py 0112-elongated-nitrogenous def2-svp > $O/grad_bench_svp.log 2>&1; grep -v amdgpu $O/grad_bench_svp.log
