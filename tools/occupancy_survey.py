"""Registers, scratch, LDS and the resulting waves per SIMD of the kernel the scheme table selects per class (J+K, fp64),
weighted by the class's serial launch time in a PMC summary (tools/final_summary.py output).
usage: python tools/occupancy_survey.py [profiles/r02_final_pmc_flops_traffic_112atoms.csv]"""
import glob, json, os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(R, "profiles/r02_final_pmc_flops_traffic_112atoms.csv")
s = json.load(open(os.path.join(R, "joltqc_amd/data/gfx950_scheme.json")))["fp64"]
t = {}
for l in open(src).read().splitlines()[1:]:
    f = l.split(",")
    m = re.match(r"jk_tile(1q)?_(\d+)$", f[0])
    if m:
        t[m.group(2)] = (float(f[1]) / 1e3, float(f[5]), float(f[4]))
print(f"# kernels of the scheme table, times and FLOP rates from {os.path.basename(src)}; llvm-readelf --notes of the cached code objects")
print("# waves/SIMD = min(160 KB / LDS, 512 / VGPRs) workgroups of 4 waves per CU")
print("class  ms  variant  vgpr  agpr  scratch_B  lds_KB  waves/SIMD  model_TF  hw_TF")
tot, occ, scr, hi = 0.0, {}, 0.0, 0.0
for cls, (ms, mtf, htf) in sorted(t.items(), key=lambda kv: -kv[1][0]):
    v = s.get(cls.lstrip("0") or "0", s.get(cls))
    fs = glob.glob(os.path.join(R, f"joltqc_amd/csrc/kcache/jk{v}_{cls}_j1k1_lr0_f64_*.hsaco"))
    if not fs:
        continue
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", fs[0]], capture_output=True, text=True).stdout
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, out).group(1))
    lds, vg, ag, sc, wg = g("group_segment_fixed_size"), g("vgpr_count"), g("agpr_count"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")
    wpw = wg // 256
    waves = min(160 * 1024 // lds, max(1, (512 // max(vg, 1)) // wpw)) * wpw
    occ[waves] = occ.get(waves, 0) + ms
    tot += ms
    scr += ms if sc else 0
    hi += ms if vg > 256 else 0
    print(f"{cls} {ms:7.1f} {hex(v):>8s} {vg:4d} {ag:4d} {sc:5d} {lds / 1024:6.1f} {waves:3d} {mtf:6.2f} {htf:6.2f}")
print(f"# {tot:.0f} ms; share of the time by waves/SIMD: " + ", ".join(f"{k}: {v / tot * 100:.0f} %" for k, v in sorted(occ.items()))
      + f"; in kernels with scratch {scr / tot * 100:.0f} %, above 256 registers {hi / tot * 100:.0f} %")
