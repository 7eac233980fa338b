#!/bin/bash
# rocprofv3 kernel stats of the great-circle search (C384 -> 0.25 deg): prints the top kernels.  usage (through gpurun): bash scripts/gc_prof.sh TAG
TAG=${1:-gc}; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/scripts/prof_step.py 10 gc > $R/gpurun_out/${TAG}_prof.log 2>&1
python3 - $R/gpurun_out/${TAG}_prof <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(f"{r['Name'][:36]:36s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
