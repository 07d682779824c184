mkdir -p gpurun_out/r2n; O=$PWD/gpurun_out/r2n
timeout 900 python -m pytest tests/test_dft_gpu.py tests/test_dft_fullsize_gpu.py tests/test_grad_gpu.py -q -m gpu --timeout=600 -k "not directional and not forces and not every_class" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for v in hip abl1 abl2; do
  JQC_LIB_PATH=$PWD/joltqc_amd/csrc/libjqc_$v.so timeout 600 python tools/dft_host_time.py > $O/vxc_$v.log 2>&1
  echo "== $v"; grep -v amdgpu $O/vxc_$v.log | grep "GGA:" | cut -c1-120
done
