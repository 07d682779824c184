#!/bin/bash
# round 6, run A': gates after the g classes were re-routed (same kernel sources: only the new variant builds are unverified), twice-run gate
# of the builds above 256 registers, the bench line with the final parity sampler
mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
JQC_TRUST_KERNELS=1 timeout 2400 python -m pytest tests -q -m gpu --timeout=900 --durations=6 -x > $O/pytest.log 2>&1; tail -10 $O/pytest.log
timeout 1800 python tools/risky_builds_gate.py run > $O/risky.log 2>&1; tail -2 $O/risky.log
JQC_TRUST_KERNELS=1 timeout 600 python bench.py --no-grid --steps 3 > $O/bench_quick.json 2> $O/bench_quick.err; head -c 300 $O/bench_quick.json
JQC_TRUST_KERNELS=1 python bench.py --workload benzene-spdfg --no-grid --no-cpu-baseline --no-parity > $O/bench_spdfg.json 2> $O/bench_spdfg.err; head -c 300 $O/bench_spdfg.json
