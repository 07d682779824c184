mkdir -p gpurun_out/r2r; O=$PWD/gpurun_out/r2r
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 -v --durations=25 > $O/pytest.log 2>&1; tail -45 $O/pytest.log
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 3000 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; tail -c 1500 $O/bench_benzene.json
