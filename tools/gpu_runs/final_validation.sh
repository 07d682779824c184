mkdir -p gpurun_out/r3f; O=$PWD/gpurun_out/r3f
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 -v --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log
bash tools/final_profile.sh > $O/final_profile.log 2>&1; tail -5 $O/final_profile.log
timeout 900 python bench.py > $O/bench_112.json 2> $O/bench_112.err; tail -c 400 $O/bench_112.json
timeout 600 python bench.py --workload benzene > $O/bench_benzene.json 2> $O/bench_benzene.err; tail -c 200 $O/bench_benzene.json
timeout 900 python tools/pair_bench.py 0112-elongated-nitrogenous def2-tzvpp > $O/pair_bench_tzvpp.log 2>&1; grep -v amdgpu $O/pair_bench_tzvpp.log | tail -3
timeout 600 python tools/jk_parts.py > $O/jk_parts.log 2>&1; grep -v amdgpu $O/jk_parts.log
timeout 900 python tools/grad_bench.py 0112-elongated-nitrogenous def2-svp > $O/grad_bench_svp.log 2>&1; grep -v amdgpu $O/grad_bench_svp.log
