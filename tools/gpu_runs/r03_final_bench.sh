#!/bin/bash
# round 3: bench lines + round profile of the final tree (the GPU suite of the same tree: tools/gpu_runs/r03_suite3.sh)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r03_final; mkdir -p $O
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 300 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; tail -c 200 $O/bench_benzene.json
bash tools/final_profile.sh > $O/final_profile.log 2>&1; tail -3 $O/final_profile.log
