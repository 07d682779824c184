set -x
mkdir -p gpurun_out/r2f; O=$PWD/gpurun_out/r2f; R=$PWD
timeout 600 python tools/host_time.py benzene > $O/host_time_benzene.log 2>&1
timeout 900 python tools/jk_parts.py > $O/jk_parts_112.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o bz -- python3 $R/bench.py --workload benzene --steps 10 --warmup 3 --no-cpu-baseline --no-grid > $O/bench_benzene_trace.json 2> $O/bench_benzene_trace.err
cd $R; python tools/timeline.py $O/trace/bz_kernel_trace.csv > $O/timeline_benzene.txt 2>&1; rm -f $O/trace/bz_kernel_trace.csv.keep
tail -5 $O/host_time_benzene.log; cat $O/jk_parts_112.log | grep -v amdgpu; head -20 $O/timeline_benzene.txt
