#!/bin/bash
# Kernel timeline of one band's search + finalize (rocprofv3 --kernel-trace): busy time and gaps per step.
# usage (through gpurun): bash scripts/band_trace.sh "8 0" [ni nlon nlat]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export FG_BAND="$1" FG_CULL=1 FG_NOPROF=1
shift
timeout -k 10 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/band_trace -o p -- python3 scripts/phase_time.py ${@:-384 1440 720} 2 20 > gpurun_out/band_trace.log 2>&1 || { tail -5 gpurun_out/band_trace.log; exit 1; }
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/band_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
for f in glob.glob("gpurun_out/band_trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
rows.sort()
# steps start at k_rect_tables; take the last 10 steps
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_rect_tables")]
steps = [rows[a:b] for a, b in zip(starts[-11:-1], starts[-10:])]
import collections
busy = collections.defaultdict(float); gap = collections.defaultdict(float)
tot = 0.0
for st in steps:
    tot += (st[-1][1] - st[0][0]) / 1e3
    for k, r in enumerate(st):
        busy[r[2]] += (r[1] - r[0]) / 1e3
        if k: gap[r[2]] += (r[0] - st[k - 1][1]) / 1e3
n = len(steps)
print(f"{n} steps; first kernel start -> last kernel end: {tot / n:.1f} us per step")
print(f"{'kernel':40s} {'busy us':>9s} {'gap before us':>14s}")
for k in busy: print(f"{k[:40]:40s} {busy[k] / n:9.1f} {gap[k] / n:14.1f}")
print(f"{'sum':40s} {sum(busy.values()) / n:9.1f} {sum(gap.values()) / n:14.1f}")
PY
