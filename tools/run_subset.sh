#!/bin/bash
# usage: tools/run_subset.sh <workload> <classes csv> name:algo:defs ...   (caches prebuilt by tools/build_variant.py)
wl=$1; cls=$2; shift 2
for v in "$@"; do n=${v%%:*}; r=${v#*:}; a=${r%%:*}; d=${r#*:}
  export JQC_EXTRA_DEFS="$d"; export JQC_KERNEL_CACHE=$PWD/joltqc_amd/csrc/kcache_$n; export JQC_JK_ALGO=$a; export JQC_ONLY_CLASS=$cls
  JQC_STREAMS=1 timeout 900 python -u tools/class_profile.py $wl > gpurun_out/cs_$n.txt 2>&1; cp gpurun_out/class_profile.json gpurun_out/cs_$n.json; grep -v amdgpu gpurun_out/cs_$n.txt | head -30
done
