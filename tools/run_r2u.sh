mkdir -p gpurun_out/r2u; O=$PWD/gpurun_out/r2u
timeout 1500 python -m pytest tests/test_jk_gpu.py tests/test_jk_pair_gpu.py -q -m gpu --timeout=900 -k "every_angular_class or variant or pair" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
JQC_PROFILE_MODE=j timeout 900 python tools/class_profile.py 0112-elongated-nitrogenous > $O/class_profile_j.txt 2>&1; grep -v amdgpu $O/class_profile_j.txt | head -12
timeout 900 python tools/jk_parts.py > $O/jk_parts.txt 2>&1; grep -v amdgpu $O/jk_parts.txt
