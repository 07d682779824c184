#!/bin/bash
# round 3: the driver's own sequence on the final tree (no JQC_TRUST_KERNELS: builds outside the manifest are cross-checked on
# first use), then the bench lines and the round profile (rocprofv3 stats of the bench + PMC passes)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r03_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2700 python -m pytest tests -q -m gpu --timeout=1500 --durations=15 > $O/pytest.log 2>&1; tail -22 $O/pytest.log
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 600 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; tail -c 300 $O/bench_benzene.json
bash tools/final_profile.sh > $O/final_profile.log 2>&1; tail -8 $O/final_profile.log
