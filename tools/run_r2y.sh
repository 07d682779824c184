mkdir -p gpurun_out/r2y; O=$PWD/gpurun_out/r2y
timeout 1200 python -m pytest tests/test_jk_gpu.py tests/test_dft_gpu.py -q -m gpu --timeout=900 -k "general_contraction" -v > $O/pytest.log 2>&1; tail -10 $O/pytest.log
