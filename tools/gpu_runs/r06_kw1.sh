#!/bin/bash
# round 6: k-chunks on wave groups (KW): the chunks of a j-in-registers class run on different waves of a 512-thread workgroup and share
# the recurrence arrays, phase A once per step with every root in one pass
export JQC_AB_TAG=r06_kw1
python tools/dev_ab.py run 3221,2122,3122 "base=" "kw1=@0x40923:-DKW=1" "kw2=@0x440923:-DKW=1" "t512=@0x440923:" > gpurun_out/r06_kw1.log 2>&1
tail -8 gpurun_out/r06_kw1.log | cut -c1-250
export JQC_AB_TAG=r06_kw1e
python tools/dev_ab.py run 3121,3211,2211,3220,2220,3320 "base=" "kw1e=@0x50923:-DKW=1" "kw2e=@0x450923:-DKW=1" > gpurun_out/r06_kw1e.log 2>&1
tail -10 gpurun_out/r06_kw1e.log | cut -c1-250
