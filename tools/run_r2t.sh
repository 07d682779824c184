mkdir -p gpurun_out/r2t; O=$PWD/gpurun_out/r2t
JQC_PROFILE_MODE=j timeout 900 python tools/class_profile.py 0112-elongated-nitrogenous > $O/class_profile_j.txt 2>&1; grep -v amdgpu $O/class_profile_j.txt | head -30
JQC_PROFILE_MODE=k timeout 900 python tools/class_profile.py 0112-elongated-nitrogenous > $O/class_profile_k.txt 2>&1; grep -v amdgpu $O/class_profile_k.txt | head -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
