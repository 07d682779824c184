mkdir -p gpurun_out/r3d
timeout 2400 python tools/autotune.py run 0112-elongated-nitrogenous > gpurun_out/r3d/autotune.log 2>&1; grep -v amdgpu gpurun_out/r3d/autotune.log | tail -30
cp gpurun_out/autotune_0112-elongated-nitrogenous.json gpurun_out/r3d/ 2>/dev/null
