#!/bin/bash
# round 6, sweep 4: the j-in-registers classes with the whole integral block per lane (no k chunks, one pass through phase A per step),
# one workgroup per CU (up to 512 registers), all roots at once
export JQC_AB_TAG=r06_sweep4 JQC_AB_NOCHECK=1
python tools/dev_ab.py run 3221,3122,3220,3211,2211,2220,3121,2122,3132,3311,3320 "base=" "e128r1=@0x040d11:-DECAP=128" "e128r2=@0x440d11:-DECAP=128" "e128r1p=@0x0c0d11:-DECAP=128" "e64r1=@0x040d11:" > gpurun_out/r06_sweep4.log 2>&1
tail -14 gpurun_out/r06_sweep4.log | cut -c1-200
