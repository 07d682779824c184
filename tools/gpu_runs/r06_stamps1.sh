#!/bin/bash
# round 6: phase stamps of row-lane kernels, scheme-table builds vs h-form builds (development sources)
W=0112-elongated-nitrogenous
export JQC_KERNEL_SRC=$PWD/joltqc_amd/csrc/kernels_dev
O=gpurun_out/r06_stamps1.txt; : > $O
run() { echo "=== $1 algo=$2 defs=$3" >> $O; JQC_JK_ALGO=$2 JQC_EXTRA_DEFS="$3" python tools/stamps_profile.py $1 $W 2>&1 | grep -v amdgpu >> $O; }
run 3221 "" ""
run 3221 v4457761 "-DHB=1 -DHEJ=2"
run 3222 "" ""
run 3222 v4457761 "-DHB=1 -DHEJ=3"
run 3121 "" ""
run 2221 "" ""
cat $O
