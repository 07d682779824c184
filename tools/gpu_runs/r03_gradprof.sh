#!/bin/bash
# round 3: per-class kernel times of the two-electron gradient; JQC_GRAD_COOP=0 one-quartet-per-lane form, 1 cooperative form
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout 600 python -m pytest tests/test_grad_gpu.py -x -q -m gpu --timeout=500 2>&1 | tail -3 > $R/gpurun_out/gradprof_tests.log
cd /tmp && export TMPDIR=/tmp
for c in ${GRAD_FORMS:-1}; do
  if [ $c = d ]; then unset JQC_GRAD_COOP; else export JQC_GRAD_COOP=$c; fi       # d: the per-class default policy
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gradprof_c$c -o g -- python3 $R/tools/grad_bench.py 0112-elongated-nitrogenous def2-tzvpp > $R/gpurun_out/gradprof_c$c.log 2>&1
done
cat $R/gpurun_out/gradprof_tests.log; grep "two-electron" $R/gpurun_out/gradprof_c*.log
