#!/bin/bash
# round 6, run B': the suite as the driver runs it on the final tree (g classes re-routed, cost weights refreshed, manifest regenerated),
# the final bench lines (default, two ranks on one device, benzene)
mkdir -p gpurun_out/r06d; O=$PWD/gpurun_out/r06d
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 300 $O/bench_112.json
JQC_BENCH_BACKEND=gloo JQC_BENCH_ONE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; head -c 200 $O/bench_2ranks.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; head -c 200 $O/bench_benzene.json
