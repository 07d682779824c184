timeout 900 python tools/decay_density_bench.py 0.2 0.5 1.0 2.0 2>&1 | grep -v amdgpu
