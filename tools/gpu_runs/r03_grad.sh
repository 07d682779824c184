#!/bin/bash
# round 3: cooperative two-electron gradient kernels (GRAD_COOP): parity tests, then timing against the round-2 form
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03_grad.txt; : > $O
timeout 900 python -m pytest tests/test_grad_gpu.py -x -q -m gpu --timeout=800 2>&1 | tail -15 >> $O
timeout 300 python tools/grad_bench.py 0112-elongated-nitrogenous def2-svp 2>&1 | grep -v amdgpu >> $O
timeout 900 python tools/grad_bench.py 0112-elongated-nitrogenous def2-tzvpp 2>&1 | grep -v amdgpu >> $O
cat $O
