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
b = os.path.join(G, "r02_stats_bench_headline", "s_kernel_stats.csv")
if os.path.exists(b):
    shutil.copy(b, os.path.join(P, "r02_bench_headline_kernel_stats.csv"))
    t, _ = stats(b, 16)
    md += ["## `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --legs none --cpu-rows 0 --gc-steps 0`  (the headline job only: "
           "every `k_clip_quad<2>` / `k_apply_il` launch here is a C384 -> 1440x720 one, so the averages are the ones the bench line's `roofline` quotes)", "",
           "Full CSV: `r02_bench_headline_kernel_stats.csv`.", "", t, ""]
for tag, title in (("search", "python3 scripts/prof_step.py 10 legacy  (search + finalize + order-2 sweep)"),
                   ("gc", "python3 scripts/prof_step.py 10 gc  (great-circle search + finalize + order-1 sweep, C384 -> 1440x720)"),
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
mg = pmc(["gc_fetch", "gc_write", "gc_sq"])
if mg:
    md += ["## PMC passes, great-circle search (`prof_step.py 3 gc`)", "",
           table(mg, ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"],
                 keys=[k for k in sorted(mg, key=lambda k: -mg[k].get("SQ_WAVE_CYCLES", 0)) if k.startswith("k_gc")]), ""]
mw = pmc(["sweep_fetch", "sweep_write", "sweep_rd", "sweep_l2", "sweep_wr", "sweep_sq"])
keys = [k for k in mw if k.startswith(("k_apply_il", "k_apply_ep8", "k_interleave3", "k_merge3"))]
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
    if k.startswith("k_apply_ep8"):
        traffic["k_apply"] = (rd + wr) * 1e6
md += ["Bytes from the by-size request counters (exact on the calibration kernel: 56.6 MB read, 56.6 MB written):", "", "\n".join(lines), "",
       "Reading: nearly every fabric read is a 128-byte request, FETCH_SIZE books it as 64 bytes -- the guide's 2x correction holds for the "
       "sweep too.  The order-2 sweep on records (`k_apply_il<2,8,4,MERGED>`) has 369 MB of algorithmic traffic per launch (133 MB CSR records, "
       "170 MB source records, 66 MB output); what it really moves depends on the tile -> XCD mapping, because neighbouring destination rows "
       "share source records and each XCD has its own L2 (separate PMC passes, `scripts/prof_xcd.sh`, two levels per lane): tiles in row order 532 MB "
       "read, L2 hit rate 0.42, 0.0966 ms; chunks of 64 tiles per XCD (the default) 384 MB, 0.56, 0.0936 ms; one band per XCD 314 MB, 0.63, "
       "0.1105 ms.  Less traffic is NOT faster in proportion: the kernel is bound by the latency of its dependent gathers at the occupancy it "
       "has (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.68; ~12 us per wave, 8 waves per SIMD), not by HBM or fabric bytes.  The tables above are the "
       "shipped configuration: `k_apply_ep8<256, 256>` (entry-parallel, chunked tiles) for the 8-level order-2 sweep on records -- 0.0833 ms "
       "back to back (`scripts/apply_ab.py`), 0.0885 ms per launch by HIP events in `bench.py`, a few us more per dispatch under rocprofv3 "
       "(which serialises dispatches with a cache write-back inside its timestamps) -- and `k_apply_il<.., 4, ..>` (four levels per lane) "
       "for the order-1 and the interleaved-array sweeps.", ""]
k = "k_clip_quad<2>"
if k in ms and "FETCH_SIZE" in ms[k] and "WRITE_SIZE" in ms[k]:
    traffic["k_clip_quad"] = (2 * ms[k]["FETCH_SIZE"] + ms[k]["WRITE_SIZE"]) * 1024.0
if k in ms and "SQ_INSTS_VALU" in ms[k]:
    traffic["k_clip_quad_valu_insts"] = ms[k]["SQ_INSTS_VALU"]
for kk in ("k_gc_walk", "k_gc_solve", "k_gc_screen"):
    if kk in mg and "FETCH_SIZE" in mg[kk] and "WRITE_SIZE" in mg[kk]:
        traffic[kk] = (2 * mg[kk]["FETCH_SIZE"] + mg[kk]["WRITE_SIZE"]) * 1024.0
traffic["k_clip_quad_pairs"] = 4589624      # candidate pairs of the launch the counters were taken on (C384 -> 1440x720, one rank)
traffic["round"] = "round 2"
traffic["_note"] = ("HBM-side bytes per launch from rocprofv3 PMC, C384 -> 1440x720: k_apply from the by-size fabric request counters "
                    "(TCC_EA0_RDREQ_{32,64,128}B, TCC_EA0_WRREQ_64B); k_clip_quad and the k_gc_* kernels = 2*FETCH_SIZE + WRITE_SIZE; "
                    "see profiles/r02_summary.md")
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
bt = os.path.join(G, "r02_band_time.txt")
if os.path.exists(bt):
    shutil.copy(bt, os.path.join(P, "r02_band_time.txt"))
    md += ["## Per-rank time of the banded search (`scripts/band_time.py`, one GPU running each rank's band in turn)", "", "```", open(bt).read().strip(), "```", ""]
open(os.path.join(P, "r02_summary.md"), "w").write("\n".join(md))
print("wrote profiles/r02_summary.md;", traffic)
