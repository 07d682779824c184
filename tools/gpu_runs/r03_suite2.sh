#!/bin/bash
# round 3: full GPU suite on the final kernels (tag a3a2040084) + bench + knob scan + multi-DM timings + K-only profile
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03_suite2; mkdir -p $O
export JQC_TRUST_KERNELS=1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu --timeout=1500 --durations=12 > $O/pytest.log 2>&1; tail -18 $O/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 600 python bench.py --workload benzene --steps 20 --warmup 5 --no-grid --no-cpu-baseline > $O/bench_benzene.json 2> $O/bench_benzene.err
bash tools/gpu_runs/r03_knobs.sh
