#!/bin/bash
# round 6, final pass after the KW builds: gates of the new sources (oracle gates with the first-use cross-check off, twice-run gate of the builds
# above 256 registers), manifest written ON the box from those results, then everything as the driver will run it with that manifest in place:
# the suite, the round's profiles and the bench lines.  The manifest and gate record come back through gpurun_out/r06e/.
mkdir -p gpurun_out/r06e; O=$PWD/gpurun_out/r06e
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
JQC_TRUST_KERNELS=1 timeout 2400 python -m pytest tests -q -m gpu --timeout=900 --durations=6 -x > $O/pytest_gates.log 2>&1; tail -3 $O/pytest_gates.log
grep -q " passed" $O/pytest_gates.log && ! grep -q " failed\| error" $O/pytest_gates.log || { echo "GATES FAILED"; exit 1; }
timeout 1800 python tools/risky_builds_gate.py run > $O/risky.log 2>&1; tail -1 $O/risky.log
cp gpurun_out/risky_builds_gate.json joltqc_amd/data/risky_builds_gate.json
python tools/make_manifest.py "round 6 final, gpurun 'bash tools/gpu_runs/r06_all.sh' on MI355X: pytest -m gpu green (profiles/r06_gpu_suite_gates_kw_sources.txt), tools/risky_builds_gate.py all builds above 256 registers gated twice" > $O/manifest.log 2>&1; tail -2 $O/manifest.log
cp joltqc_amd/data/verified_kernels.json joltqc_amd/data/risky_builds_gate.json $O/
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
bash tools/final_profile.sh > $O/final_profile.log 2>&1; tail -3 $O/final_profile.log
cp gpurun_out/final/pmc_traffic.json profiles/r06_pmc_traffic_112atoms_tzvpp.json
bash tools/pmc_profile.sh r06_pmc_final 0112-elongated-nitrogenous > $O/pmc_final.log 2>&1; head -4 gpurun_out/r06_pmc_final/summary.txt | cut -c1-220
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 300 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; head -c 250 $O/bench_benzene.json
JQC_BENCH_BACKEND=gloo JQC_BENCH_ONE_DEVICE=1 timeout 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; head -c 250 $O/bench_2ranks.json
timeout 900 python bench.py --workload 0166-irregular-nitrogenous --steps 2 --warmup 1 --no-grid --no-cpu-baseline > $O/bench_166.json 2> $O/bench_166.err; head -c 250 $O/bench_166.json
