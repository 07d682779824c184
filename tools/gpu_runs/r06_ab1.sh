#!/bin/bash
# round 6, A/B 1: h form (bra HRR in phase A, lane = (ci, j-group)) of the row-lane kernels vs the scheme table
CL=3221,3121,3222,2221,3122,2122,3220,3232,3132,3211,3231,2211
export JQC_AB_TAG=r06_ab1
python tools/dev_ab.py run $CL "base=" "h3r2=@0x440521:-DHB=1 -DHEJ=3" "h3r3=@0x840521:-DHB=1 -DHEJ=3" "h2r2=@0x440521:-DHB=1 -DHEJ=2" "h6r3=@0x840521:-DHB=1 -DHEJ=6" "h1r2=@0x440521:-DHB=1 -DHEJ=1" > gpurun_out/r06_ab1.log 2>&1
tail -30 gpurun_out/r06_ab1.log
