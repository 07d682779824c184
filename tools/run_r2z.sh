mkdir -p gpurun_out/r2z; O=$PWD/gpurun_out/r2z
timeout 1200 python -m pytest tests/test_jk_gpu.py tests/test_boundary_gpu.py -q -m gpu --timeout=900 -k "shell_block_max or rks_reset" -v > $O/pytest.log 2>&1; tail -10 $O/pytest.log
