#!/bin/bash
# round 6, sweep 3: owner reduction with hoisted destination decode / grouped reads / pairwise sums (ORED_HOIST) on the scheme-table
# builds and on the h-form winners; oracle check of the new code on both
export JQC_AB_TAG=r06_sweep3 JQC_AB_NOCHECK=1
python tools/dev_ab.py run rowlane "base=" "base0=-DORED_HOIST=0" "h1r1=@0x040521:-DHB=1 -DHEJ=1" "h2r1=@0x040521:-DHB=1 -DHEJ=2" "h3r1=@0x040521:-DHB=1 -DHEJ=3" "h3r2=@0x440521:-DHB=1 -DHEJ=3" "h2r2=@0x440521:-DHB=1 -DHEJ=2" > gpurun_out/r06_sweep3.log 2>&1
tail -36 gpurun_out/r06_sweep3.log | cut -c1-200
unset JQC_AB_NOCHECK
export JQC_AB_TAG=r06_check3
python tools/dev_ab.py run 3221,3121,3222,2221,3232,3231,3322,2222,3331,3233,2022,3211 "base=" "h2r1=@0x040521:-DHB=1 -DHEJ=2" "h3r1=@0x040521:-DHB=1 -DHEJ=3" > gpurun_out/r06_check3.log 2>&1
head -4 gpurun_out/r06_check3.log | cut -c1-300
