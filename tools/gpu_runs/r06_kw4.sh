#!/bin/bash
# round 6: KW with the Rys table in LDS (one workgroup per CU leaves room for it)
export JQC_AB_TAG=r06_kw4 JQC_AB_NOCHECK=1
python tools/dev_ab.py run 3221,2122,3122,2221 "kw1u=@0x40923:-DKW=1 -DUNROLL_B=0" "kw1l=@0x40823:-DKW=1 -DUNROLL_B=0" "kw2l=@0x440823:-DKW=1 -DUNROLL_B=0" "kw1lp=@0xc0823:-DKW=1 -DUNROLL_B=0" > gpurun_out/r06_kw4.log 2>&1
tail -5 gpurun_out/r06_kw4.log | cut -c1-250
