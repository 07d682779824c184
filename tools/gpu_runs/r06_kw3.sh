#!/bin/bash
# round 6: k-chunks on wave groups (KW) on the five classes with two chunks: root groups, partial unroll, more owner-reduction rows,
# per-root phase A, and the h form on top (bra HRR in phase A, all j components per lane)
export JQC_AB_TAG=r06_kw3
python tools/dev_ab.py run 3221,2122,3122,2221,3131 "base=" "kw1u=@0x40923:-DKW=1 -DUNROLL_B=0" "kw2u=@0x440923:-DKW=1 -DUNROLL_B=0" "kw1u2=@0x40923:-DKW=1 -DUNROLL_B=2" "kw1r=@0x40923:-DKW=1 -DUNROLL_B=0 -DRGMIN_ROWS=23" "kw1p=@0xc0923:-DKW=1 -DUNROLL_B=0" "hbkw2=@0x26440123:-DKW=1 -DUNROLL_B=0" "hbkw3=@0x26840123:-DKW=1 -DUNROLL_B=0" "hbkw2u=@0x26440123:-DKW=1" > gpurun_out/r06_kw3.log 2>&1
tail -16 gpurun_out/r06_kw3.log | cut -c1-250
