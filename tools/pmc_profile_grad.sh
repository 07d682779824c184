#!/bin/bash
# PMC profile of gradient class kernels (jk_grad_<class>, default form per class) -> gpurun_out/<tag>/pass{1,2,3}; summary by tools/pmc_summary.py
# usage: tools/pmc_profile_grad.sh <tag> <class> [<class> ...]      (three separate --pmc passes, kernel trace only)
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/pass1 -o p --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -- python3 $R/tools/grad_ab.py pmc "$@" > $out/pass1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pass2 -o p --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU_INT32 -- python3 $R/tools/grad_ab.py pmc "$@" > $out/pass2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/pass3 -o p --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT -- python3 $R/tools/grad_ab.py pmc "$@" > $out/pass3.log 2>&1
cd $R && python3 tools/pmc_summary.py $out > $out/summary.txt; head -30 $out/summary.txt
