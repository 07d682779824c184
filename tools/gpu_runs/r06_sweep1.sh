#!/bin/bash
# round 6, sweep 1: h-form variants over every row-lane class (timing only; winners are oracle-checked afterwards) + sub-stamps
W=0112-elongated-nitrogenous
export JQC_KERNEL_SRC=$PWD/joltqc_amd/csrc/kernels_dev
O=gpurun_out/r06_stamps2.txt; : > $O
run() { echo "=== $1 algo=$2 defs=$3" >> $O; JQC_JK_ALGO=$2 JQC_EXTRA_DEFS="$3" python tools/stamps_profile.py $1 $W 2>&1 | grep -v amdgpu >> $O; }
run 3221 "" ""
run 3221 v4457761 "-DHB=1 -DHEJ=2"
run 3222 v4457761 "-DHB=1 -DHEJ=3"
run 3121 "" ""
run 2221 "" ""
unset JQC_KERNEL_SRC
export JQC_AB_TAG=r06_sweep1 JQC_AB_NOCHECK=1
python tools/dev_ab.py run rowlane "base=" \
  "h1r1=@0x040521:-DHB=1 -DHEJ=1" "h1r2=@0x440521:-DHB=1 -DHEJ=1" "h1r3=@0x840521:-DHB=1 -DHEJ=1" \
  "h2r1=@0x040521:-DHB=1 -DHEJ=2" "h2r2=@0x440521:-DHB=1 -DHEJ=2" "h2r3=@0x840521:-DHB=1 -DHEJ=2" \
  "h3r1=@0x040521:-DHB=1 -DHEJ=3" "h3r2=@0x440521:-DHB=1 -DHEJ=3" "h3r3=@0x840521:-DHB=1 -DHEJ=3" \
  "h6r2=@0x440521:-DHB=1 -DHEJ=6" "h6r3=@0x840521:-DHB=1 -DHEJ=6" "h3r2p=@0x4c0521:-DHB=1 -DHEJ=3" "h1r2p=@0x4c0521:-DHB=1 -DHEJ=1" \
  > gpurun_out/r06_sweep1.log 2>&1
tail -50 gpurun_out/r06_sweep1.log
