#!/bin/bash
# per-class timing of both tile algorithms (JIT into /tmp caches); merge with tools/make_scheme.py
for algo in tile tile1q; do
  export JQC_JK_ALGO=$algo
  export JQC_KERNEL_CACHE=/tmp/kc_$algo
  JQC_STREAMS=1 timeout 900 python -u tools/class_profile.py "$@" > gpurun_out/classprof_$algo.txt 2>&1
  cp gpurun_out/class_profile.json gpurun_out/class_profile_$algo.json
  head -3 gpurun_out/classprof_$algo.txt
done
