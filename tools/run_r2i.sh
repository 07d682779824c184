set -x
mkdir -p gpurun_out/r2i; O=$PWD/gpurun_out/r2i
timeout 900 python tools/grid_leg_probe.py > $O/grid_leg_probe.log 2>&1; grep -v amdgpu $O/grid_leg_probe.log | tail -8
