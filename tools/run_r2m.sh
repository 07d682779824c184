mkdir -p gpurun_out/r2m; O=$PWD/gpurun_out/r2m
timeout 1200 python -m pytest tests/test_grad_gpu.py -q -m gpu --timeout=1200 -v -k "rks_forces or rhf_forces" > $O/pytest.log 2>&1; tail -30 $O/pytest.log
timeout 900 python tools/grad_bench.py 0112-elongated-nitrogenous def2-svp > $O/grad_bench_svp.log 2>&1; grep -v amdgpu $O/grad_bench_svp.log
timeout 1500 python tools/grad_bench.py 0112-elongated-nitrogenous def2-tzvpp > $O/grad_bench_tzvpp.log 2>&1; grep -v amdgpu $O/grad_bench_tzvpp.log
