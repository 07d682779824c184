mkdir -p gpurun_out/r2l; O=$PWD/gpurun_out/r2l
timeout 2400 python -m pytest tests/test_grad_gpu.py -q -m gpu --timeout=1200 -v -k xc > $O/pytest.log 2>&1; tail -40 $O/pytest.log
