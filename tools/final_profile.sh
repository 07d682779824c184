#!/bin/bash
# Round profile of the headline benchmark: kernel-trace stats of `bench.py`, then PMC passes (FP64 instruction counters,
# HBM-side traffic) on the serialised class kernels.  Output: gpurun_out/final/*, summarised by tools/final_summary.py.
out=$PWD/gpurun_out/final; rm -rf $out; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
# usage: tools/final_profile.sh [workload]   (default: the bench's default workload, 112 atoms / def2-TZVPP)
WL=${1:-0112-elongated-nitrogenous}
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-grid > $out/bench_line.json 2> $out/bench.err
export JQC_STREAMS=1
rocprofv3 --kernel-trace --output-format csv -d $out/flop -o p --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 $R/tools/class_profile.py $WL > $out/flop.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/fetch -o p --pmc FETCH_SIZE -- python3 $R/tools/class_profile.py $WL > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/write -o p --pmc WRITE_SIZE -- python3 $R/tools/class_profile.py $WL > $out/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/atomic -o p --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -- python3 $R/tools/class_profile.py $WL > $out/atomic.log 2>&1
cd $R && python3 tools/final_summary.py $out > $out/summary.txt; head -30 $out/summary.txt
