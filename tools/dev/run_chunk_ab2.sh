#!/bin/bash
export JQC_QUAD_MAX=600
Q2=$((0x1001122)); Q2n=$((0x1001022)); Q4=$((0x1002122))
mkdir -p gpurun_out/r05
run() { cls=$1; shift; JQC_AB_TAG=chunk2_$cls python tools/dev_ab.py run $cls "$@" > gpurun_out/r05/chunk2_$cls.log 2>&1; tail -1 gpurun_out/r05/chunk2_$cls.log; grep -c '"bad": \[\]' gpurun_out/r05/chunk2_$cls.log; }
run 2121 "base=" "c2k=@$Q2:-DQNCH=2 -DQY=0" "c3k=@$Q2:-DQNCH=3 -DQY=0" "c2n=@$Q2n:-DQNCH=2 -DQY=0" "c2q=@$Q4:-DQNCH=2 -DQY=0" "c2yk=@$Q2:-DQNCH=2 -DQY=2"
run 3111 "base=" "c2k=@$Q2:-DQNCH=2 -DQY=0" "c2n=@$Q2n:-DQNCH=2 -DQY=0" "c2q=@$Q4:-DQNCH=2 -DQY=0" "c3k=@$Q2:-DQNCH=3 -DQY=1"
