mkdir -p gpurun_out/r2o; O=$PWD/gpurun_out/r2o; R=$PWD
timeout 600 python -m pytest tests/test_dft_gpu.py -q -m gpu --timeout=600 -k "build_grids" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/grad -o g -- python3 $R/tools/grad_bench.py 0112-elongated-nitrogenous def2-tzvpp > $O/grad_bench_tzvpp.log 2>&1
cd $R; rm -f $O/grad/g_kernel_trace.csv; head -25 $O/grad/g_kernel_stats.csv | cut -c1-120
