"""Print the headline numbers of one scripts/gpu_cycle.sh run: summarize_run.py TAG"""
import csv, glob, json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/{tag}_b.json").read().strip().split("\n")[-1])
    print("cells/s %.3e  ms/step %.4f" % (d["value"], d["ms_per_step"]))
    print({k: round(v, 4) for k, v in d["phase_ms"].items()})
except Exception as e:
    print("no bench line:", e)
for f in glob.glob(f"gpurun_out/{tag}_prof/*/*_kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
        print(f"{r['Name'][:64]:64s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us")
