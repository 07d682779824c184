"""Refresh "ns_per_quartet" of joltqc_amd/data/gfx950_scheme.json (the weights of the multi-GPU split, pyscf/jk.py:_shard_assign) from a
per-class profile of the 112-atom workload (gpurun_out/class_profile.json, written by tools/class_profile.py on the GPU box).
usage: python tools/refresh_costs.py [class_profile.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "class_profile.json")
path = os.path.join(ROOT, "joltqc_amd", "data", "gfx950_scheme.json")
sch = json.load(open(path))
rows = json.load(open(src))
n = 0
for r in rows:
    key = str(1000 * r["ang"][0] + 100 * r["ang"][1] + 10 * r["ang"][2] + r["ang"][3])
    if r["quartets"] > 0:
        sch["ns_per_quartet"][key] = round(r["ms"] * 1e6 / r["quartets"], 4)
        n += 1
json.dump(sch, open(path, "w"), indent=1)
print(f"{n} classes refreshed from {src}")
