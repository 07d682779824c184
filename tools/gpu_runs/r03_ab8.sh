#!/bin/bash
# round 3, A/B 8: ket chunk rotated by the bra pair index (XCD balance), all 65 classes, development kernels; then the knob scan
export HSA_ENABLE_IPC_MODE_LEGACY=0
export JQC_TRUST_KERNELS=1
O=gpurun_out/r03_ab8; mkdir -p $O
n=0
for cfg in "" "-DCH_ROT=1" "" "-DCH_ROT=1"; do
  n=$((n+1)); echo "### run $n defs='$cfg'" >> $O/prof_all.txt
  JQC_KERNEL_SRC=$PWD/joltqc_amd/csrc/kernels_dev JQC_EXTRA_DEFS="$cfg" JQC_STREAMS=1 timeout 600 python tools/class_profile.py 0112-elongated-nitrogenous >> $O/prof_all.txt 2>&1
  JQC_KERNEL_SRC=$PWD/joltqc_amd/csrc/kernels_dev JQC_EXTRA_DEFS="$cfg" timeout 300 python tools/step_time.py >> $O/step.txt 2>&1
done
grep -h "###\|total serial" $O/prof_all.txt; grep "J+K" $O/step.txt
