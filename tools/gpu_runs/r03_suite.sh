#!/bin/bash
# round 3: full GPU suite on the adopted kernels + bench + rocprofv3 profile of the bench + grid-path bench
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03_suite; mkdir -p $O
export JQC_TRUST_KERNELS=1      # the gates ARE the verification; tools/make_manifest.py lists the builds after a green run
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu --timeout=1500 --durations=15 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 python bench.py --workload benzene --steps 20 --warmup 5 --no-grid --no-cpu-baseline > $O/bench_benzene.json 2> $O/bench_benzene.err
timeout 300 python tools/dft_bench.py 0112-elongated-nitrogenous def2-tzvpp 344064 > $O/dft_bench_tzvpp.txt 2>&1; tail -12 $O/dft_bench_tzvpp.txt
JQC_RHO_MGGA=1 timeout 300 python tools/dft_bench.py 0112-elongated-nitrogenous def2-tzvpp 344064 > $O/dft_bench_tzvpp_old_mgga.txt 2>&1; grep -i mgga $O/dft_bench_tzvpp_old_mgga.txt | head -4
timeout 1500 bash tools/final_profile.sh > $O/final_profile.log 2>&1; tail -5 $O/final_profile.log
