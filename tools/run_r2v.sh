mkdir -p gpurun_out/r2v; O=$PWD/gpurun_out/r2v; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/grad -o g -- python3 $R/tools/grad_bench.py 0112-elongated-nitrogenous def2-tzvpp > $O/grad_bench_tzvpp.log 2>&1
cd $R; rm -f $O/grad/g_kernel_trace.csv; grep jk_grad $O/grad/g_kernel_stats.csv | head -30 | cut -c1-110
