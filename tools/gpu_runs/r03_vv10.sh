#!/bin/bash
# round 3: packed-FP32 VV10 kernel: outer points per lane (JQC_VV10_PK) x inner-loop split (JQC_VV10_SPLIT; 0 = automatic)
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03_vv10.txt; : > $O
for cfg in "0 1 0" "2 1 0" "2 0 0" "2 2 0" "2 8 0" "4 0 0" "4 4 0" "4 16 0" "2 0 1" "4 0 1"; do set -- $cfg
  echo "== JQC_VV10_PK=$1 JQC_VV10_SPLIT=$2 JQC_VV10_RCP2=$3" >> $O
  JQC_VV10_PK=$1 JQC_VV10_SPLIT=$2 JQC_VV10_RCP2=$3 python tools/vv10_bench.py 2>&1 | grep -v "amdgpu\|fp64" >> $O; done
echo "== N = 1 048 576 (config-4 size)" >> $O
for cfg in "0 1" "2 0" "4 0" "4 1"; do set -- $cfg; echo "== JQC_VV10_PK=$1 JQC_VV10_SPLIT=$2" >> $O
  JQC_VV10_PK=$1 JQC_VV10_SPLIT=$2 python tools/vv10_bench.py 1048576 2>&1 | grep -v "amdgpu\|fp64" >> $O; done
timeout 900 python -m pytest tests -q -m gpu -k "vv10 or VV10 or nlc" --timeout=800 2>&1 | tail -3 >> $O
cat $O
