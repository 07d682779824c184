mkdir -p gpurun_out/r2x; O=$PWD/gpurun_out/r2x
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 -v --durations=12 > $O/pytest.log 2>&1; tail -18 $O/pytest.log
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 600 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; tail -c 300 $O/bench_benzene.json
JQC_BENCH_BACKEND=gloo JQC_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks_gloo.err; tail -c 1200 $O/bench_2ranks_gloo.json | head -c 700
