set -x
mkdir -p gpurun_out/r2g; O=$PWD/gpurun_out/r2g; R=$PWD
timeout 900 python -m pytest tests/test_boundary_gpu.py tests/test_dft_gpu.py tests/test_jk_pair_gpu.py -q -m gpu --timeout=600 -v > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dft -o d -- python3 $R/tools/dft_host_time.py > $O/dft_host_time.log 2>&1
cd $R; grep -v amdgpu $O/dft_host_time.log | head -60; rm -f $O/dft/d_kernel_trace.csv
