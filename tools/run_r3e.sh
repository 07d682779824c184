mkdir -p gpurun_out/r3e
for o in rows morton; do JQC_GRID_ORDER=$o timeout 600 python tools/dft_host_time.py 2>&1 | grep -v amdgpu | grep "GGA:" | cut -c1-200 | sed "s/^/$o: /"; done
timeout 2400 python tools/autotune.py run 0112-elongated-nitrogenous > gpurun_out/r3e/autotune.log 2>&1; grep -v amdgpu gpurun_out/r3e/autotune.log | tail -22
cp gpurun_out/autotune_0112-elongated-nitrogenous.json gpurun_out/r3e/ 2>/dev/null
