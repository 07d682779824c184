#!/bin/bash
# usage: tools/tune_minw.sh  -- times the per-class profile for several register budgets
for mw in 1 2 3; do
  export JQC_EXTRA_DEFS="-DMINW=$mw"
  export JQC_KERNEL_CACHE=/tmp/kc_mw$mw
  echo "=== MINW=$mw"
  JQC_STREAMS=1 timeout 600 python -u tools/class_profile.py 2>&1 | head -12
done
