"""Per-kernel table of the counter groups scripts/prof_mix.sh collected (means over dispatches), plus the kernel durations of
the same runs.   usage: python scripts/summarize_mix.py gpurun_out r03"""
import csv, glob, os, re, sys
from collections import defaultdict

G, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in sorted(glob.glob(os.path.join(G, f"{tag}_mix_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
m = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
ks = sorted((k for k in m if k.startswith("k_")), key=lambda k: -m[k].get("SQ_INSTS_VALU", 0))
cols = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
        "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH",
        "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAVES",
        "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
print("| kernel | us (pmc runs) | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |")
print("|---|---|" + "---|" * len(cols))
for k in ks[:14]:
    t = sum(dur[k]) / len(dur[k]) if dur[k] else float("nan")
    print(f"| `{k}` | {t:.1f} | " + " | ".join(f"{m[k].get(c, float('nan')):.4g}" for c in cols) + " |")
print()
for k in ks[:6]:
    v = m[k]
    iv = v.get("SQ_INSTS_VALU", 0)
    if not iv:
        continue
    f64 = sum(v.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    flops = v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_TRANS_F64", 0)
    t = sum(dur[k]) / len(dur[k]) if dur[k] else float("nan")
    lanes = v.get("SQ_THREAD_CYCLES_VALU", 0) / v["SQ_ACTIVE_INST_VALU"] if v.get("SQ_ACTIVE_INST_VALU") else float("nan")
    print(f"{k}: FP64 share of VALU instructions {f64 / iv:.3f}; int32 {v.get('SQ_INSTS_VALU_INT32', 0) / iv:.3f}; int64 {v.get('SQ_INSTS_VALU_INT64', 0) / iv:.3f}; "
          f"wave-level FP64 flop/s = 64 x {flops:.4g} / {t:.1f} us = {64 * flops / (t * 1e-6) / 1e12:.2f} TF (all lanes counted); "
          f"lanes active per VALU cycle (THREAD_CYCLES / ACTIVE_INST) {lanes:.1f}")
