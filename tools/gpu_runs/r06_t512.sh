#!/bin/bash
# round 6: 512-thread workgroups (one set of tiles per CU, every root in one pass of phase A, jobs over 512 lanes) on the one-chunk row-lane classes
export JQC_AB_TAG=r06_t512 JQC_AB_NOCHECK=1
python tools/dev_ab.py run 3121,3220,3211,2211,2220,3311,3320,3130,3310,3230,3031,3022,2022,3032,3330 "base=" "t1=@0x40923:" "t2=@0x440923:" "t1p=@0xc0923:" "t1n=@0x40123:" > gpurun_out/r06_t512.log 2>&1
tail -17 gpurun_out/r06_t512.log | cut -c1-200
