#!/bin/bash
# usage: tools/run_variants.sh name:defs ...   (caches prebuilt by tools/build_variant.py)
for v in "$@"; do n=${v%%:*}; d=${v#*:}
  export JQC_EXTRA_DEFS="$d"; export JQC_KERNEL_CACHE=$PWD/joltqc_amd/csrc/kcache_$n; export JQC_JK_ALGO=${ALGO:-tile}
  JQC_STREAMS=1 timeout 600 python -u tools/class_profile.py > gpurun_out/cp_$n.txt 2>&1; cp gpurun_out/class_profile.json gpurun_out/cp_$n.json; head -2 gpurun_out/cp_$n.txt | tail -1
done
