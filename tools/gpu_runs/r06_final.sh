#!/bin/bash
# round 6, run B: the suite exactly as the driver runs it (verified manifest of the new sources in place), the round's profiles (kernel-trace
# stats of the bench, FP64 / traffic counter passes, three SQ passes), the bench lines (default workload, benzene, two ranks on one device,
# config-4 size)
mkdir -p gpurun_out/r06b; O=$PWD/gpurun_out/r06b
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
bash tools/final_profile.sh > $O/final_profile.log 2>&1; tail -4 $O/final_profile.log
cp gpurun_out/final/pmc_traffic.json profiles/r06_pmc_traffic_112atoms_tzvpp.json
bash tools/pmc_profile.sh r06_pmc_final 0112-elongated-nitrogenous > $O/pmc_final.log 2>&1; head -8 gpurun_out/r06_pmc_final/summary.txt | cut -c1-220
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 600 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; head -c 300 $O/bench_benzene.json
JQC_BENCH_BACKEND=gloo JQC_BENCH_ONE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; head -c 300 $O/bench_2ranks.json; tail -3 $O/bench_2ranks.err
timeout 900 python bench.py --workload 0166-irregular-nitrogenous --steps 2 --warmup 1 --no-grid --no-cpu-baseline > $O/bench_166.json 2> $O/bench_166.err; head -c 300 $O/bench_166.json
