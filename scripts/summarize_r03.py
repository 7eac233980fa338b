"""profiles/r03_*: turn the rocprofv3 CSVs of scripts/prof_r03.sh under gpurun_out/ into profiles/r03_summary.md, copy the
kernel-stats CSVs and refresh profiles/pmc_traffic.json (the static PMC numbers bench.py's roofline objects quote).
usage: python scripts/summarize_r03.py"""
import csv, json, os, re, shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "").strip()
FP64_PEAK = 78.6e12          # MI355X vector FP64 (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
VALU_PEAK = 1024 * 2.4e9     # wave-instruction issue slots: a wave64 VALU instruction holds its SIMD for 4 cycles


def stats(path, top):
    rows = list(csv.DictReader(open(path)))
    out = ["| kernel | calls | avg us | total % |", "|---|---|---|---|"]
    for r in rows[:top]:
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |")
    return "\n".join(out), {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in rows}


def pmc(tags):
    acc = defaultdict(lambda: defaultdict(list))
    for t in tags:
        f = os.path.join(G, f"r03_pmc_{t}", "p_counter_collection.csv")
        if os.path.exists(f):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def table(m, counters, keys=None, n=12):
    ks = keys or sorted((k for k in m if k.startswith("k_")), key=lambda k: -m[k].get("SQ_INSTS_VALU", m[k].get("SQ_WAVE_CYCLES", m[k].get("FETCH_SIZE", 0))))[:n]
    out = ["| kernel | " + " | ".join(c.replace("SQ_", "").replace("INSTS_VALU_", "") for c in counters) + " |", "|---|" + "---|" * len(counters)]
    for k in ks:
        if k in m:
            out.append(f"| `{k}` | " + " | ".join(f"{m[k].get(c, float('nan')):.4g}" for c in counters) + " |")
    return "\n".join(out)


def mix_lines(m, avg_us, names):
    out = []
    res = {}
    for k in names:
        if k not in m or "SQ_INSTS_VALU" not in m[k]:
            continue
        v = m[k]
        iv = v["SQ_INSTS_VALU"]
        f64 = sum(v.get("SQ_INSTS_VALU_" + c, 0) for c in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64"))
        flops = v.get("SQ_INSTS_VALU_ADD_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_TRANS_F64", 0)
        lanes = v.get("SQ_THREAD_CYCLES_VALU", 0) / v["SQ_ACTIVE_INST_VALU"] if v.get("SQ_ACTIVE_INST_VALU") else float("nan")
        t = avg_us.get(k)
        res[k] = {"valu_insts": iv, "fp64_insts": f64, "fp64_wave_flops": flops, "int32_insts": v.get("SQ_INSTS_VALU_INT32", 0),
                  "int64_insts": v.get("SQ_INSTS_VALU_INT64", 0), "lanes_active_per_valu_inst": lanes}
        if t:
            busy = 4.0 * iv / (VALU_PEAK * t * 1e-6)
            tf_all = 64 * flops / (t * 1e-6)
            tf_live = lanes * flops / (t * 1e-6)
            out.append(f"* `{k}` ({t:.1f} us by `--stats`): {iv:.4g} wave VALU instructions -> pipe-busy {busy:.3f} of the issue rate; FP64 share of VALU "
                       f"instructions {f64 / iv:.3f} (int32 {v.get('SQ_INSTS_VALU_INT32', 0) / iv:.3f}, int64 {v.get('SQ_INSTS_VALU_INT64', 0) / iv:.3f}, the rest "
                       f"moves / selects / compares); lanes active per VALU instruction {lanes:.1f} of 64; FP64 flop/s = {tf_all / 1e12:.1f} TF counting all 64 lanes "
                       f"of every FP64 instruction = {tf_all / FP64_PEAK:.3f} of {FP64_PEAK / 1e12:.1f} TF, {tf_live / 1e12:.1f} TF = {tf_live / FP64_PEAK:.3f} counting the lanes that were live")
    return out, res


md = ["# Round 03 rocprofv3 summaries (MI355X, ROCm 7.2)", ""]
avg_head = {}
b = os.path.join(G, "r03_stats_bench_headline", "s_kernel_stats.csv")
if os.path.exists(b):
    shutil.copy(b, os.path.join(P, "r03_bench_headline_kernel_stats.csv"))
    t, avg_head = stats(b, 18)
    md += ["## `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --legs none --cpu-rows 0 --gc-steps 0`  (the headline job only: "
           "every search / sweep launch here is a C384 -> 1440x720 one, so these averages are the ones the bench line's `roofline` objects must agree with)", "",
           "Full CSV: `r03_bench_headline_kernel_stats.csv`.", "", t, ""]
avg = {}
for tag, title in (("search", "python3 scripts/prof_step.py 10 legacy  (search + finalize + order-2 sweep)"),
                   ("gc", "python3 scripts/prof_step.py 10 gc  (great-circle search + finalize + order-1 sweep, C384 -> 1440x720)"),
                   ("sweep", "python3 scripts/prof_step.py 10 sweep  (order-2 level-major, order-2 interleaved, order-1 level-major sweeps)")):
    f = os.path.join(G, f"r03_stats_{tag}", "s_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, f"r03_{tag}_kernel_stats.csv"))
        t, a = stats(f, 18)
        avg[tag] = a
        md += [f"## `rocprofv3 --kernel-trace --stats -- {title}`", "", f"Full CSV: `r03_{tag}_kernel_stats.csv`.", "", t, ""]
traffic = {}
ms = pmc(["search_fetch", "search_write", "search_f64", "search_int", "search_cyc"])
if ms:
    md += ["## PMC passes, legacy search (`scripts/prof_r03.sh`: one counter group per run, `--kernel-trace --pmc <group>`, `prof_step.py 3 legacy`)", "",
           "FETCH_SIZE / WRITE_SIZE in KB per dispatch (mean; FETCH_SIZE counts a 128-byte fabric request as 64 bytes on gfx950: read bytes = 2 x FETCH_SIZE).", "",
           table(ms, ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CU_CYCLES"]), "",
           "Instruction mix (wave-level instruction counts per dispatch):", "",
           table(ms, ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32",
                      "SQ_INSTS_VALU_INT64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"]), ""]
    names = [k for k in ms if k.startswith("k_clip_quad<2") or k.startswith("k_cell_struct2r") or k.startswith("k_candidates_rect") or k.startswith("k_compact<2")]
    lines, res = mix_lines(ms, avg.get("search", {}), names)
    md += ["What the VALU time of the search kernels is made of:", ""] + lines + [""]
    for k in ms:
        if k.startswith("k_clip_quad<2"):
            if "FETCH_SIZE" in ms[k] and "WRITE_SIZE" in ms[k]:
                traffic["k_clip_quad"] = (2 * ms[k]["FETCH_SIZE"] + ms[k]["WRITE_SIZE"]) * 1024.0
            if k in res:
                traffic["k_clip_quad_valu_insts"] = res[k]["valu_insts"]
                traffic["k_clip_quad_fp64_insts"] = res[k]["fp64_insts"]
                traffic["k_clip_quad_fp64_wave_flops"] = res[k]["fp64_wave_flops"]
                traffic["k_clip_quad_int32_insts"] = res[k]["int32_insts"]
                traffic["k_clip_quad_lanes_active"] = res[k]["lanes_active_per_valu_inst"]
mg = pmc(["gc_fetch", "gc_write", "gc_f64", "gc_int", "gc_cyc"])
if mg:
    keys = [k for k in sorted(mg, key=lambda k: -mg[k].get("SQ_INSTS_VALU", 0)) if k.startswith("k_gc")]
    md += ["## PMC passes, great-circle search (`prof_step.py 3 gc`)", "",
           table(mg, ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"], keys), "",
           table(mg, ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32",
                      "SQ_INSTS_VALU_INT64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"], keys), ""]
    lines, res = mix_lines(mg, avg.get("gc", {}), keys)
    md += lines + [""]
    for kk in ("k_gc_walk", "k_gc_solve", "k_gc_screen"):
        if kk in mg and "FETCH_SIZE" in mg[kk] and "WRITE_SIZE" in mg[kk]:
            traffic[kk] = (2 * mg[kk]["FETCH_SIZE"] + mg[kk]["WRITE_SIZE"]) * 1024.0
        if kk in res:
            traffic[kk + "_valu_insts"] = res[kk]["valu_insts"]
            traffic[kk + "_fp64_insts"] = res[kk]["fp64_insts"]
            traffic[kk + "_int_insts"] = res[kk]["int32_insts"] + res[kk]["int64_insts"]
            traffic[kk + "_lanes_active"] = res[kk]["lanes_active_per_valu_inst"]
mw = pmc(["sweep_rd", "sweep_l2", "sweep_wr"])
keys = [k for k in mw if k.startswith(("k_apply_il", "k_apply_ep8", "k_interleave3", "k_merge3"))]
if keys:
    lines = ["| kernel | read MB (32/64/128-byte requests) | write MB | L2 hit rate |", "|---|---|---|---|"]
    for k in keys:
        c = mw[k]
        if "TCC_EA0_RDREQ_sum" not in c:
            continue
        rd = (32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)) / 1e6
        wr = 64 * c.get("TCC_EA0_WRREQ_64B_sum", 0) / 1e6
        hit = c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
        lines.append(f"| `{k}` | {rd:.1f} | {wr:.1f} | {hit:.2f} |")
        if k.startswith("k_apply_ep8"):
            traffic["k_apply"] = (rd + wr) * 1e6
    md += ["## PMC passes, sweep (bytes from the by-size fabric request counters, as calibrated in round 2 on `k_interleave3<8>`)", "", "\n".join(lines), ""]
old = {}
pj = os.path.join(P, "pmc_traffic.json")
if os.path.exists(pj):
    old = json.load(open(pj))
for k, v in old.items():                      # keep what this round did not re-measure, labelled by its round
    if k not in traffic and not k.startswith("_") and k != "round":
        traffic[k] = v
        traffic.setdefault("_carried_over_from", {})[k] = old.get("round", "an earlier round")
traffic["k_clip_quad_pairs"] = 4589624      # candidate pairs of the launch the counters were taken on (C384 -> 1440x720, one rank)
traffic["round"] = "round 3"
traffic["_note"] = ("rocprofv3 PMC, C384 -> 1440x720, per launch: k_apply bytes from the by-size fabric request counters; k_clip_quad (= k_clip_quad<2, true>, the "
                    "rectilinear-target clip) and k_gc_* bytes = 2*FETCH_SIZE + WRITE_SIZE; *_valu_insts = SQ_INSTS_VALU, *_fp64_insts = ADD+MUL+FMA+TRANS_F64, "
                    "*_fp64_wave_flops = ADD+MUL+2*FMA+TRANS (x64 lanes = flops with every lane counted), *_lanes_active = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; "
                    "see profiles/r03_summary.md")
json.dump(traffic, open(pj, "w"), indent=1)
bt = os.path.join(P, "r03_band_time.txt")
if os.path.exists(bt):
    md += ["## Per-rank time of the banded search (`scripts/band_time.py`, one GPU running each rank's band in turn, culling on)", "", "```", open(bt).read().strip(), "```", ""]
open(os.path.join(P, "r03_summary.md"), "w").write("\n".join(md))
print("wrote profiles/r03_summary.md;", {k: v for k, v in traffic.items() if not k.startswith("_")})
