#!/bin/bash
# round 6: does the re-routed scheme table (tuned on 112 atoms / def2-TZVPP) hold on the def2-SVP workloads of BASELINE configs 3 and 5?
# per-class kernel times with the round-5 table and with the round-6 table, 112 atoms and 425 atoms / def2-SVP; config-5 size bench line
mkdir -p gpurun_out/r06f; O=gpurun_out/r06f
for W in 0112-elongated-nitrogenous@def2-svp 0425-globular-nitrogenous@def2-svp; do
  JQC_SCHEME_JSON=$PWD/tools/r05_scheme_for_ab.json JQC_TRUST_KERNELS=1 python tools/class_profile.py $W > $O/r05_table_$W.txt 2>&1
  JQC_TRUST_KERNELS=1 python tools/class_profile.py $W > $O/r06_table_$W.txt 2>&1
  grep "total serial" $O/r05_table_$W.txt $O/r06_table_$W.txt
done
timeout 900 python bench.py --workload 0425-globular-nitrogenous@def2-svp --steps 3 --warmup 1 --no-grid --no-cpu-baseline > $O/bench_425_svp.json 2> $O/bench_425_svp.err; head -c 300 $O/bench_425_svp.json
