#!/bin/bash
# GPU A/B of the chunked quad builds (tools/dev_ab.py per class; results in gpurun_out/r05/chunk_<class>.log)
export JQC_QUAD_MAX=600
Q1=$((0x1000122)); Q2=$((0x1001122))
mkdir -p gpurun_out/r05
run() { cls=$1; shift; JQC_AB_TAG=chunk_$cls python tools/dev_ab.py run $cls "$@" > gpurun_out/r05/chunk_$cls.log 2>&1; tail -$(( $# + 1 )) gpurun_out/r05/chunk_$cls.log | head -$#; tail -1 gpurun_out/r05/chunk_$cls.log; }
run 2121 "base=" "c2=@$Q1:-DQNCH=2 -DQY=0" "c2k=@$Q2:-DQNCH=2 -DQY=0" "c3=@$Q1:-DQNCH=3 -DQY=0" "c2y=@$Q1:-DQNCH=2 -DQY=2"
run 3111 "base=" "c2=@$Q1:-DQNCH=2 -DQY=0" "c2k=@$Q2:-DQNCH=2 -DQY=0" "c3=@$Q1:-DQNCH=3 -DQY=1" "c5=@$Q1:-DQNCH=5 -DQY=0"
run 2211 "base=" "c2=@$Q1:-DQNCH=2 -DQY=0" "c2k=@$Q2:-DQNCH=2 -DQY=0" "c3=@$Q1:-DQNCH=3 -DQY=0"
run 3121 "base=" "c3=@$Q1:-DQNCH=3 -DQY=2" "c6=@$Q1:-DQNCH=6 -DQY=2" "c5=@$Q1:-DQNCH=5 -DQY=0"
run 3211 "base=" "c3=@$Q1:-DQNCH=3 -DQY=1" "c6=@$Q1:-DQNCH=6 -DQY=1" "c5=@$Q1:-DQNCH=5 -DQY=0"
