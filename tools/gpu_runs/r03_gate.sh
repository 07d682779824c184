#!/bin/bash
# round 3: gates of the adopted kernels (owner reduction, NDM2, re-tuned table) + bench + VV10 A/B + DMA-staging A/B
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r03_gate; mkdir -p $O
export JQC_TRUST_KERNELS=1      # the gates below ARE the verification; the manifest is written from a green run
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 3300 python -m pytest tests -q -m gpu --timeout=1500 --durations=30 > $O/pytest.log 2>&1; tail -45 $O/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
for n in 1 2 4; do JQC_VV10_NOUT=$n timeout 200 python tools/vv10_bench.py >> $O/vv10.txt 2>&1; done; cat $O/vv10.txt
T1Q=1010,1000,1110,2110,2010,1100,2011,1011,1111,2111,3110,3021,0000,2000,3121,3221,2121,3111
JQC_AB_TAG=r03_dma timeout 1200 python tools/dev_ab.py run $T1Q base= dma="-DDMA_STAGE=1" > $O/ab_dma.txt 2>&1; tail -22 $O/ab_dma.txt
