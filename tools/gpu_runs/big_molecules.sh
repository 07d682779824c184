mkdir -p gpurun_out/r3m; O=$PWD/gpurun_out/r3m
timeout 1500 python tools/big_check.py 0425-globular-nitrogenous def2-svp check > $O/big_0425.log 2>&1; grep -v amdgpu $O/big_0425.log | tail -11
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
