#!/bin/bash
# round 6, sweep 5: the h form on the classes the scheme table routes to the lane-per-quartet / quad kernels (100-330 integrals)
export JQC_AB_TAG=r06_sweep5 JQC_AB_NOCHECK=1
python tools/dev_ab.py run 2121,2111,3111,3120,3210,3021,2120,2210,2021,3011,3110,2110,3020,2011,3200,2200,3030 "base=" "h1r1=@0x040521:-DHB=1 -DHEJ=1" "h3r1=@0x040521:-DHB=1 -DHEJ=3" "h6r1=@0x040521:-DHB=1 -DHEJ=6" "h3r1nl=@0x040421:-DHB=1 -DHEJ=3" > gpurun_out/r06_sweep5.log 2>&1
tail -20 gpurun_out/r06_sweep5.log | cut -c1-200
