#!/bin/bash
# Round profile of the headline benchmark: kernel-trace stats of `bench.py`, then PMC passes (FP64 instruction counters,
# HBM-side traffic) on the serialised class kernels.  Output: gpurun_out/final/*, summarised by tools/final_summary.py.
out=$PWD/gpurun_out/final; rm -rf $out; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_line.json 2> $out/bench.err
export JQC_STREAMS=1
rocprofv3 --kernel-trace --output-format csv -d $out/flop -o p --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- python3 $R/tools/class_profile.py "$@" > $out/flop.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/fetch -o p --pmc FETCH_SIZE -- python3 $R/tools/class_profile.py "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/write -o p --pmc WRITE_SIZE -- python3 $R/tools/class_profile.py "$@" > $out/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/atomic -o p --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -- python3 $R/tools/class_profile.py "$@" > $out/atomic.log 2>&1
cd $R && python3 tools/final_summary.py $out > $out/summary.txt; head -30 $out/summary.txt
