#!/bin/bash
# round 6: the row-lane classes on benzene / def2-TZVPP (small launches: the "fp64_small" table) -- h form x components per lane x root groups
export JQC_AB_TAG=r06_sweep_benzene JQC_AB_NOCHECK=1 JQC_AB_WORKLOAD=benzene
python tools/dev_ab.py run rowlane "base=" "h0r1=@0x20040521:" "h0r2=@0x20440521:" "h1r1=@0x22040521:" "h1r2=@0x22440521:" "h2r1=@0x24040521:" "h2r2=@0x24440521:" "h3r1=@0x26040521:" "h3r2=@0x26440521:" "c1=@0x40d21:" "c2=@0x440d21:" "o1=@0x40521:" > gpurun_out/r06_sweep_benzene.log 2>&1
tail -4 gpurun_out/r06_sweep_benzene.log | cut -c1-300
