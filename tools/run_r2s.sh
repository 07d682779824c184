mkdir -p gpurun_out/r2s
JQC_AB_TAG=sched timeout 2400 python tools/dev_ab.py run all base= "ilp=-mllvm -amdgpu-sched-strategy=max-ilp" "bias0=-mllvm -amdgpu-schedule-metric-bias=0" > gpurun_out/r2s/ab.log 2>&1
grep -v amdgpu gpurun_out/r2s/ab.log | head -80
