#!/bin/bash
# round 6: the small lane-per-quartet classes (the bulk of the def2-SVP workloads, configs 3 and 5) over waves per SIMD x ket pairs per iteration x
# Rys table in LDS / through L2 (+ quad builds), timed on BOTH the def2-SVP and the def2-TZVPP 112-atom molecule
CL=1000,1010,2110,2010,1110,2111,1100,0000,1111,2011,2000,1011,2100,3000,3010,3100
for W in 0112-elongated-nitrogenous@def2-svp 0112-elongated-nitrogenous; do
export JQC_AB_TAG=r06_sweep_small_${W##*@} JQC_AB_NOCHECK=1 JQC_AB_WORKLOAD=$W
python tools/dev_ab.py run $CL "base=" "m2n1=@0x22:" "m2n1l=@0x122:" "m2n2=@0x1022:" "m2n2l=@0x1122:" "m2n4=@0x2022:" "m2n4l=@0x2122:" "m3n1=@0x32:" "m3n1l=@0x132:" "m3n2=@0x1032:" "m3n2l=@0x1132:" "m3n4=@0x2032:" "m3n4l=@0x2132:" "m4n1=@0x42:" "m4n1l=@0x142:" "m4n2=@0x1042:" "m4n2l=@0x1142:" "m4n4=@0x2042:" "m4n4l=@0x2142:" "q22=@0x1001022:" "q22l=@0x1001122:" "q32=@0x1001032:" "q31l=@0x1000132:" "q31=@0x1000032:" > gpurun_out/r06_sweep_small_${W##*@}.log 2>&1
tail -3 gpurun_out/r06_sweep_small_${W##*@}.log | cut -c1-200
done
