mkdir -p gpurun_out/r2w; O=$PWD/gpurun_out/r2w
timeout 1200 python -m pytest tests/test_grad_gpu.py tests/test_dft_gpu.py -q -m gpu --timeout=900 -k "xc or dft" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
