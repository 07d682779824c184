"""Register/spill survey of the kernel cache: reads the code-object metadata (llvm-readelf --notes) of every
.hsaco under csrc/kcache and prints sgpr/vgpr counts and spill counts, sorted by SGPR spills."""
import os, re, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RE = "/opt/rocm/lib/llvm/bin/llvm-readelf"
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "joltqc_amd/csrc/kcache/*.hsaco"))):
    out = subprocess.run([RE, "--notes", f], capture_output=True, text=True).stdout
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, out).group(1))
    rows.append((g("sgpr_spill_count"), g("vgpr_spill_count"), g("sgpr_count"), g("vgpr_count"), g("agpr_count"),
                 g("group_segment_fixed_size"), g("private_segment_fixed_size"), os.path.basename(f)))
rows.sort(reverse=True)
print("sgpr_spill vgpr_spill sgpr vgpr agpr lds scratch file")
for r in rows[: int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print(*r)
n = len(rows)
print(f"{n} kernels; with sgpr spills: {sum(r[0] > 0 for r in rows)}; with vgpr spills: {sum(r[1] > 0 for r in rows)}; scratch>0: {sum(r[6] > 0 for r in rows)}")
