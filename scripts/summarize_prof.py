"""Turn the rocprofv3 CSVs under gpurun_out/ (scripts/prof_pmc.sh + the --stats runs) into profiles/<round>_summary.md,
copy the kernel-stats CSVs and refresh profiles/pmc_traffic.json.   usage: python scripts/summarize_prof.py r01"""
import csv, json, os, re, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").strip()


def stats_table(path, top=26):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | avg us | total % |", "|---|---|---|---|"]
    for r in rows[:top]:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |")
    return "\n".join(out), {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}


def pmc_means(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        f = os.path.join(G, d, "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def pmc_table(m, counters):
    ks = sorted((k for k in m if k.startswith("k_")), key=lambda k: -m[k].get("SQ_WAVE_CYCLES", m[k].get("FETCH_SIZE", 0)))
    out = ["| kernel | " + " | ".join(counters) + " |", "|---|" + "---|" * len(counters)]
    for k in ks:
        out.append(f"| `{k}` | " + " | ".join(f"{m[k].get(c, float('nan')):.4g}" for c in counters) + " |")
    return "\n".join(out)


md = [f"# Round {tag[1:]} rocprofv3 summaries (MI355X, ROCm 7.2)", ""]
src = os.path.join(G, "keep", "bench_kernel_stats.csv")
if not os.path.exists(src):
    src = os.path.join(G, "prof_stats", "bench_kernel_stats.csv")
shutil.copy(src, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
t, avg = stats_table(src)
md += ["## `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 --cpu-rows 0`  (C384 -> 1440x720, order 2)", "",
       f"Full CSV: `{tag}_bench_kernel_stats.csv`.  Top kernels:", "", t, ""]
srcg = os.path.join(G, "keep", "gc_kernel_stats.csv")
if not os.path.exists(srcg):
    srcg = os.path.join(G, "prof_stats_gc", "gc_kernel_stats.csv")
if os.path.exists(srcg):
    shutil.copy(srcg, os.path.join(P, f"{tag}_gc_kernel_stats.csv"))
    t, avg_gc = stats_table(srcg, 12)
    md += ["## `rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/prof_step.py 5 gc`  (great-circle search, C384 -> 1440x720, + first-order sweep)", "",
           f"Full CSV: `{tag}_gc_kernel_stats.csv`.  Top kernels:", "", t, ""]
m = pmc_means(["prof_pmc1", "prof_pmc2", "prof_pmc3", "prof_pmc4"])
cs = ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"]
md += ["## PMC passes (`scripts/prof_pmc.sh`: `rocprofv3 --kernel-trace --pmc <group> -- python3 scripts/prof_step.py 3 legacy`, one run per group; FETCH_SIZE and WRITE_SIZE in separate runs)", "",
       "FETCH_SIZE / WRITE_SIZE are KB per dispatch (mean over dispatches).  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x "
       "(MI355X_MICROARCH.md, HBM section); `traffic` in bench.py applies that correction to reads: bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.", "",
       pmc_table(m, cs), ""]
mg = pmc_means(["prof_pmc_gc1", "prof_pmc_gc2", "prof_pmc_gc3"])
if mg:
    md += ["## PMC passes, great-circle search (`prof_step.py 3 gc`)", "", pmc_table(mg, cs[:5]), ""]
open(os.path.join(P, f"{tag}_summary.md"), "w").write("\n".join(md))
traffic = {}
for key, kname in (("k_clip_quad", "k_clip_quad<2>"), ("k_apply", "k_apply_il<2, 8, 4, true, 1>")):
    if kname in m and "FETCH_SIZE" in m[kname] and "WRITE_SIZE" in m[kname]:
        traffic[key] = (2 * m[kname]["FETCH_SIZE"] + m[kname]["WRITE_SIZE"]) * 1024.0
if "k_clip_quad<2>" in m and "SQ_INSTS_VALU" in m["k_clip_quad<2>"]:
    traffic["k_clip_quad_valu_insts"] = m["k_clip_quad<2>"]["SQ_INSTS_VALU"]      # wave instructions per launch
if mg.get("k_gc_clip"):
    traffic["k_gc_clip"] = (2 * mg["k_gc_clip"].get("FETCH_SIZE", 0) + mg["k_gc_clip"].get("WRITE_SIZE", 0)) * 1024.0
traffic["_note"] = ("HBM bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, KB->bytes), C384->1440x720, "
                    f"round {tag[1:]}; see profiles/{tag}_summary.md")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
