"""Resource usage of the code objects in a kernel cache directory (vgpr / agpr / sgpr spills / scratch / LDS)."""
import glob, os, re, subprocess, sys
RE = "/opt/rocm/lib/llvm/bin/llvm-readelf"
d = sys.argv[1] if len(sys.argv) > 1 else "joltqc_amd/csrc/kcache_dev"
pat = sys.argv[2] if len(sys.argv) > 2 else ""
print("vgpr agpr sgpr_spill vgpr_spill scratch lds  file")
for f in sorted(glob.glob(os.path.join(d, "*.hsaco")), key=lambda x: (os.path.basename(x).split("_")[1], os.path.getmtime(x))):
    if pat not in f:
        continue
    out = subprocess.run([RE, "--notes", f], capture_output=True, text=True).stdout
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, out).group(1))
    print(f'{g("vgpr_count"):4d} {g("agpr_count"):4d} {g("sgpr_spill_count"):4d} {g("vgpr_spill_count"):4d} {g("private_segment_fixed_size"):5d} {g("group_segment_fixed_size"):6d}  {os.path.basename(f)}')
