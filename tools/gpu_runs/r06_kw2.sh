#!/bin/bash
# round 6: k-chunks on wave groups (KW) with the root loop of phase B rolled (the unrolled five-root loop spilled 540-620 B)
export JQC_AB_TAG=r06_kw2 JQC_AB_NOCHECK=1
python tools/dev_ab.py run 3221,2122 "base=" "kw1u=@0x40923:-DKW=1 -DUNROLL_B=0" "kw2u=@0x440923:-DKW=1 -DUNROLL_B=0" "kw3u=@0x840923:-DKW=1 -DUNROLL_B=0" > gpurun_out/r06_kw2.log 2>&1
tail -4 gpurun_out/r06_kw2.log | cut -c1-250
export JQC_AB_TAG=r06_kw2e
python tools/dev_ab.py run 3121,3220,2220,3320 "base=" "kw1eu=@0x50923:-DKW=1 -DUNROLL_B=0" "kw2eu=@0x450923:-DKW=1 -DUNROLL_B=0" > gpurun_out/r06_kw2e.log 2>&1
tail -6 gpurun_out/r06_kw2e.log | cut -c1-250
