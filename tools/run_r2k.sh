mkdir -p gpurun_out/r2k; O=$PWD/gpurun_out/r2k
timeout 2400 python -m pytest tests/test_grad_gpu.py -q -m gpu --timeout=1200 -v > $O/pytest.log 2>&1; tail -40 $O/pytest.log
