#!/bin/bash
# Final GPU pass of round 5: ECP tests + timing, final profiles (kernel-trace stats of the bench, flop / traffic passes, three SQ passes), benches
mkdir -p gpurun_out/r05
python -m pytest tests/test_ecp_gpu.py -x -q > gpurun_out/r05/t_ecp2.log 2>&1; tail -3 gpurun_out/r05/t_ecp2.log
python tools/ecp_bench.py 2 > gpurun_out/r05/ecp_bench.log 2>&1; python tools/ecp_bench.py 3 >> gpurun_out/r05/ecp_bench.log 2>&1; cat gpurun_out/r05/ecp_bench.log
bash tools/final_profile.sh > gpurun_out/r05/final_profile.log 2>&1; tail -5 gpurun_out/r05/final_profile.log
bash tools/pmc_profile.sh r05_pmc_final 0112-elongated-nitrogenous > gpurun_out/r05/pmc_final.log 2>&1; head -12 gpurun_out/r05_pmc_final/summary.txt
python bench.py --workload benzene > gpurun_out/r05/bench_benzene.json 2> gpurun_out/r05/bench_benzene.err; head -c 400 gpurun_out/r05/bench_benzene.json
python bench.py --workload 0166-irregular-nitrogenous --steps 2 --warmup 1 --no-grid --no-cpu-baseline > gpurun_out/r05/bench_166.json 2> gpurun_out/r05/bench_166.err; head -c 400 gpurun_out/r05/bench_166.json
