"""Aggregate rocprofv3 --pmc CSVs (tools/pmc_profile.sh) per kernel name: mean counter values per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not k.startswith("jk_"):
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"], r["Grid_Size"])
dur = collections.defaultdict(list)
for f in glob.glob(root + "/pass1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("jk_"):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
def m(k, c):
    v = vals[k].get(c)
    return sum(v) / len(v) if v else float("nan")
rows = []
for k in vals:
    d = sum(dur[k]) / max(len(dur[k]), 1)
    rows.append((d, k))
rows.sort(reverse=True)
print("kernel us vgpr agpr lds scratch grid | valu_busy%% (ACTIVE_VALU/WAVE_CYC) wait_any%% wait_inst%% | VALU/wave FMA64 MUL64 ADD64 INT32 SALU LDS(ld/st/at) VMEM | flop64_frac lds_active%% bankconf%%")
for d, k in rows:
    wc, w = m(k, "SQ_WAVE_CYCLES"), m(k, "SQ_WAVES")
    f = m(k, "SQ_INSTS_VALU_FMA_F64"); mu = m(k, "SQ_INSTS_VALU_MUL_F64"); ad = m(k, "SQ_INSTS_VALU_ADD_F64")
    valu = m(k, "SQ_INSTS_VALU")
    print(f"{k:28s} {d:9.1f} {meta[k][0]:>4s} {meta[k][1]:>4s} {meta[k][2]:>6s} {meta[k][3]:>5s} {meta[k][4]:>8s} | "
          f"{100*m(k,'SQ_ACTIVE_INST_VALU')/wc:5.1f} {100*m(k,'SQ_WAIT_ANY')/wc:5.1f} {100*m(k,'SQ_WAIT_INST_ANY')/wc:5.1f} | "
          f"{valu/w:9.0f} {f/w:8.0f} {mu/w:7.0f} {ad/w:7.0f} {m(k,'SQ_INSTS_VALU_INT32')/w:7.0f} {m(k,'SQ_INSTS_SALU')/w:7.0f} "
          f"{m(k,'SQ_INSTS_LDS_LOAD')/w:6.0f}/{m(k,'SQ_INSTS_LDS_STORE')/w:5.0f}/{m(k,'SQ_INSTS_LDS_ATOMIC')/w:5.0f} {m(k,'SQ_INSTS_VMEM')/w:6.0f} | "
          f"{(f+mu+ad)/valu:5.2f} {100*m(k,'SQ_ACTIVE_INST_LDS')/wc:5.1f} {100*m(k,'SQ_LDS_BANK_CONFLICT')/max(m(k,'SQ_LDS_IDX_ACTIVE'),1):5.1f}")
