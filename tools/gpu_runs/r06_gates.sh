#!/bin/bash
# round 6, run A: the gates of the new kernel sources (no verified manifest exists for them yet: the oracle gates ARE the check, so the
# first-use cross-check is switched off for this run), the twice-run gate of the builds above 256 registers, benzene old vs new table
mkdir -p gpurun_out/r06a; O=gpurun_out/r06a
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
JQC_TRUST_KERNELS=1 timeout 2400 python -m pytest tests -q -m gpu --timeout=900 --durations=10 -x > $O/pytest.log 2>&1; tail -16 $O/pytest.log
timeout 1800 python tools/risky_builds_gate.py run > $O/risky.log 2>&1; tail -3 $O/risky.log
JQC_SCHEME_JSON=$PWD/tools/r05_scheme_for_ab.json JQC_TRUST_KERNELS=1 python tools/class_profile.py benzene > $O/benzene_r05_table.txt 2>&1
JQC_TRUST_KERNELS=1 python tools/class_profile.py benzene > $O/benzene_r06_table.txt 2>&1
head -1 $O/benzene_r05_table.txt; head -1 $O/benzene_r06_table.txt
JQC_TRUST_KERNELS=1 timeout 600 python bench.py --no-grid --no-cpu-baseline --steps 3 > $O/bench_quick.json 2> $O/bench_quick.err; head -c 600 $O/bench_quick.json
