"""profiles/r02_*: turn the rocprofv3 CSVs of scripts/prof_pmc2.sh (+ the bench --stats run) under gpurun_out/ into
profiles/r02_summary.md, copy the kernel-stats CSVs, refresh profiles/pmc_traffic.json.   usage: python scripts/summarize_r02.py"""
import csv, json, os, re, shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "").strip()


def stats(path, top):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | avg us | total % |", "|---|---|---|---|"]
    for r in rows[:top]:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |")
    return "\n".join(out), {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}


def pmc(tags):
    acc = defaultdict(lambda: defaultdict(list))
    for t in tags:
        f = os.path.join(G, f"r02_pmc_{t}", "p_counter_collection.csv")
        if os.path.exists(f):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def table(m, counters, keys=None):
    ks = keys or sorted((k for k in m if k.startswith("k_")), key=lambda k: -m[k].get("SQ_WAVE_CYCLES", 0))
    out = ["| kernel | " + " | ".join(counters) + " |", "|---|" + "---|" * len(counters)]
    for k in ks:
        if k in m:
            out.append(f"| `{k}` | " + " | ".join(f"{m[k].get(c, float('nan')):.4g}" for c in counters) + " |")
    return "\n".join(out)


md = ["# Round 02 rocprofv3 summaries (MI355X, ROCm 7.2)", ""]
b = os.path.join(G, "r02_stats_bench", "s_kernel_stats.csv")
if os.path.exists(b):
    shutil.copy(b, os.path.join(P, "r02_bench_kernel_stats.csv"))
    t, _ = stats(b, 30)
    md += ["## `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py`  (default run: C384 -> 1440x720 order 2 + all legs)", "",
           "Full CSV: `r02_bench_kernel_stats.csv`.  Top kernels:", "", t, ""]
for tag, title in (("search", "python3 scripts/prof_step.py 10 legacy  (search + finalize + order-2 sweep)"),
                   ("sweep", "python3 scripts/prof_step.py 10 sweep  (order-2 level-major, order-2 interleaved, order-1 level-major sweeps)")):
    f = os.path.join(G, f"r02_stats_{tag}", "s_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, f"r02_{tag}_kernel_stats.csv"))
        t, _ = stats(f, 18)
        md += [f"## `rocprofv3 --kernel-trace --stats -- {title}`", "", f"Full CSV: `r02_{tag}_kernel_stats.csv`.", "", t, ""]
ms = pmc(["search_fetch", "search_write", "search_sq"])
md += ["## PMC passes, search (`scripts/prof_pmc2.sh`: one counter group per run, `--kernel-trace --pmc <group>`, `prof_step.py 3 legacy`)", "",
       "FETCH_SIZE / WRITE_SIZE in KB per dispatch (mean).  FETCH_SIZE counts a 128-byte fabric request as 64 bytes on gfx950 (see the sweep "
       "section: the by-size counters prove it), so read bytes = 2 x FETCH_SIZE for these kernels.", "",
       table(ms, ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"]), ""]
mw = pmc(["sweep_fetch", "sweep_write", "sweep_rd", "sweep_l2", "sweep_wr", "sweep_sq"])
keys = [k for k in mw if k.startswith(("k_apply_il", "k_interleave3", "k_merge3"))]
md += ["## PMC passes, sweep: what the fabric counters really count", "",
       "`k_interleave3<8>` is a pure streaming kernel of known size (8 levels x 884 736 doubles = 56.6 MB in, 56.6 MB out): the calibration.", "",
       table(mw, ["FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_64B_sum"], keys), "",
       table(mw, ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"], keys), ""]
lines = ["| kernel | read MB (32/64/128-byte requests) | RDREQ x 64 B = FETCH_SIZE | write MB | L2 hit rate |", "|---|---|---|---|---|"]
traffic = {}
for k in keys:
    c = mw[k]
    if "TCC_EA0_RDREQ_sum" not in c:
        continue
    rd = (32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)) / 1e6
    wr = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) / 1e6
    hit = c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
    lines.append(f"| `{k}` | {rd:.1f} | {c['TCC_EA0_RDREQ_sum'] * 64 / 1e6:.1f} | {wr:.1f} | {hit:.2f} |")
    if k in ("k_apply_il<2, 8, 4, true>", "k_apply_il<2, 8, 2, true>"):
        traffic["k_apply"] = (rd + wr) * 1e6
md += ["Bytes from the by-size request counters (exact on the calibration kernel: 56.6 MB read, 56.6 MB written):", "", "\n".join(lines), "",
       "Reading: nearly every fabric read is a 128-byte request, FETCH_SIZE books it as 64 bytes -- the guide's 2x correction holds for the "
       "sweep too.  `k_apply_il<2,8,2,MERGED>` moves 538 MB in + 66 MB out per launch against 369 MB algorithmic (1.64x): the gathered source "
       "records are fetched by more than one XCD (L2 hit rate 0.41).  Those re-fetches are served by the Infinity Cache (the 170 MB record array "
       "fits), which the fabric counters include; an XCD-banded row mapping that removes them is SLOWER (scripts/apply_ab.py: 0.110 vs 0.096 ms).", ""]
k = "k_clip_quad<2>"
if k in ms and "FETCH_SIZE" in ms[k] and "WRITE_SIZE" in ms[k]:
    traffic["k_clip_quad"] = (2 * ms[k]["FETCH_SIZE"] + ms[k]["WRITE_SIZE"]) * 1024.0
if k in ms and "SQ_INSTS_VALU" in ms[k]:
    traffic["k_clip_quad_valu_insts"] = ms[k]["SQ_INSTS_VALU"]
old = json.load(open(os.path.join(P, "pmc_traffic.json")))
if "k_gc_clip" in old:
    traffic["k_gc_clip"] = old["k_gc_clip"]
traffic["round"] = "round 2"
traffic["_note"] = ("HBM-side bytes per launch from rocprofv3 PMC, C384 -> 1440x720: k_apply from the by-size fabric request counters "
                    "(TCC_EA0_RDREQ_{32,64,128}B, TCC_EA0_WRREQ_64B); k_clip_quad = 2*FETCH_SIZE + WRITE_SIZE; k_gc_clip from round 1; "
                    "see profiles/r02_summary.md")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
bt = os.path.join(G, "r02_band_time.txt")
if os.path.exists(bt):
    shutil.copy(bt, os.path.join(P, "r02_band_time.txt"))
    md += ["## Per-rank time of the banded search (`scripts/band_time.py`, one GPU running each rank's band in turn)", "", "```", open(bt).read().strip(), "```", ""]
open(os.path.join(P, "r02_summary.md"), "w").write("\n".join(md))
print("wrote profiles/r02_summary.md;", traffic)
