#!/bin/bash
# round 3: the driver's own GPU sequence again (no JQC_TRUST_KERNELS) after the fallback record + the larger ahead-of-time set
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r03_suite3; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu --timeout=1500 --durations=25 > $O/pytest.log 2>&1; tail -32 $O/pytest.log
ls joltqc_amd/csrc/kcache | wc -l
