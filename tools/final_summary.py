"""Per-kernel summary of tools/final_profile.sh: duration, FP64 flops executed (hardware counters) vs the algorithmic
model, HBM-side traffic (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as read)."""
import csv, glob, json, sys, collections
root = sys.argv[1]
def load(sub):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("jk_"):
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{root}/{sub}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("jk_"):
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return vals, dur
fl, dur = load("flop")
fe, _ = load("fetch")
wr, _ = load("write")
at, _ = load("atomic")
mean = lambda v: sum(v) / len(v) if v else float("nan")
model = {}
try:
    for r in json.load(open("gpurun_out/class_profile.json")):
        model["%d%d%d%d" % tuple(r["ang"])] = (r["flop"], r["quartets"])
except Exception:
    pass
rows = []
for k in fl:
    c = fl[k]
    hw = 64 * (2 * mean(c["SQ_INSTS_VALU_FMA_F64"]) + mean(c["SQ_INSTS_VALU_MUL_F64"]) + mean(c["SQ_INSTS_VALU_ADD_F64"]))
    us = mean(dur[k])
    cls = k.split("_")[-1]
    mf, nq = model.get(cls, (float("nan"), float("nan")))
    fetch = 2 * 1024 * mean(fe[k].get("FETCH_SIZE", []))
    write = 1024 * mean(wr[k].get("WRITE_SIZE", []))
    rows.append((us, k, hw, mf, nq, fetch, write, mean(at[k].get("TCC_EA0_ATOMIC_sum", [])),
                 mean(at[k].get("TCC_HIT_sum", [])), mean(at[k].get("TCC_MISS_sum", []))))
rows.sort(reverse=True)
print("kernel, us/launch, hw FP64 flop (64 lanes x (2 FMA + MUL + ADD) wave instructions; upper bound: counts masked lanes), "
      "model flop, hw TFLOP/s, model TFLOP/s, quartets, HBM-side read MB (2 x FETCH_SIZE), write MB (WRITE_SIZE, incl. atomics), "
      "GB/s, EA atomic requests, L2 hit rate")
for us, k, hw, mf, nq, fetch, write, atom, hit, miss in rows:
    print(f"{k},{us:.1f},{hw:.4g},{mf:.4g},{hw/us/1e6:.2f},{mf/us/1e6:.2f},{nq},{fetch/1e6:.2f},{write/1e6:.2f},"
          f"{(fetch+write)/us/1e3:.1f},{atom:.4g},{hit/(hit+miss) if hit+miss else float('nan'):.3f}")
# machine-readable copy for bench.py's roofline.traffic (profiles/*pmc_traffic*.json)
json.dump({k: {"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "us_per_launch": us,
               "quartets": (nq if nq == nq else None),
               "note": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction of MI355X_MICROARCH.md) and WRITE_SIZE, separate passes"}
           for us, k, hw, mf, nq, fetch, write, atom, hit, miss in rows if fetch == fetch and write == write},
          open(f"{root}/pmc_traffic.json", "w"), indent=1)
tot_us = sum(r[0] for r in rows)
print(f"TOTAL,{tot_us:.1f},{sum(r[2] for r in rows):.4g},{sum(r[3] for r in rows if r[3]==r[3]):.4g},,,"
      f"{sum(r[4] for r in rows if r[4]==r[4])},{sum(r[5] for r in rows)/1e6:.2f},{sum(r[6] for r in rows)/1e6:.2f}")
