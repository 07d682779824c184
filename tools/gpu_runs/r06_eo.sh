#!/bin/bash
# round 6, timing probe (WRONG results: the tables are not converted): even / odd split of the Rys Chebyshev series, four Clenshaw chains of depth 7
export JQC_AB_TAG=r06_eo_tzvpp JQC_AB_NOCHECK=1
python tools/dev_ab.py run 1000,1010,2110,2010,1110,0000,2111,1100,3221,3121,2121 "base=" "eo=-DRYS_EO=1" > gpurun_out/r06_eo.log 2>&1; tail -13 gpurun_out/r06_eo.log | cut -c1-120
export JQC_AB_TAG=r06_eo_svp JQC_AB_WORKLOAD=0112-elongated-nitrogenous@def2-svp
python tools/dev_ab.py run 1000,1010,2110,2010,1110,0000,2111,1100 "base=" "eo=-DRYS_EO=1" > gpurun_out/r06_eo_svp.log 2>&1; tail -10 gpurun_out/r06_eo_svp.log | cut -c1-120
