mkdir -p gpurun_out/r3p; O=$PWD/gpurun_out/r3p
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu --timeout=900 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
