mkdir -p gpurun_out/r2q; O=$PWD/gpurun_out/r2q
timeout 1200 python -m pytest tests/test_boundary_gpu.py -q -m gpu --timeout=900 -v > $O/pytest.log 2>&1; tail -14 $O/pytest.log
