#!/bin/bash
# round 3: launch-geometry knobs, multi-density-matrix timings, K-only class profile (new kernels, AOT cache)
export HSA_ENABLE_IPC_MODE_LEGACY=0 JQC_TRUST_KERNELS=1
O=gpurun_out/r03_knobs; mkdir -p $O
timeout 900 python tools/step_time.py 0112-elongated-nitrogenous ndm > $O/ndm.txt 2>&1; cat $O/ndm.txt
JQC_NDM2=0 timeout 600 python tools/step_time.py 0112-elongated-nitrogenous ndm > $O/ndm_off.txt 2>&1; cat $O/ndm_off.txt
for kc in 8 32 64; do JQC_KCHUNK_MAX=$kc timeout 300 python tools/step_time.py >> $O/knobs.txt 2>&1; done
for st in 2 8; do JQC_STREAMS=$st timeout 300 python tools/step_time.py >> $O/knobs.txt 2>&1; done
JQC_CHUNK_ALIGN=8 timeout 300 python tools/step_time.py >> $O/knobs.txt 2>&1
JQC_TARGET_WGS=16384 timeout 300 python tools/step_time.py >> $O/knobs.txt 2>&1
grep "J+K" $O/knobs.txt
JQC_PROFILE_MODE=k timeout 600 python tools/class_profile.py 0112-elongated-nitrogenous > $O/class_profile_k_only.txt 2>&1; head -12 $O/class_profile_k_only.txt
timeout 600 python tools/class_profile.py 0112-elongated-nitrogenous > $O/class_profile_jk.txt 2>&1; head -5 $O/class_profile_jk.txt
