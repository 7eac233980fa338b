#!/bin/bash
# fabric read traffic of the sweep under the tile -> XCD mappings (FG_XCD = 0 identity, 1 banded, 64 chunked)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
for x in 0 64 1; do
  FG_XCD=$x timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $R/xcd_rd_$x -o p -- python3 scripts/prof_step.py 3 sweep > $R/xcd_rd_$x.log 2>&1 || { echo fail $x; exit 1; }
  FG_XCD=$x timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/xcd_l2_$x -o p -- python3 scripts/prof_step.py 3 sweep > $R/xcd_l2_$x.log 2>&1 || { echo fail $x; exit 1; }
  echo done $x
done
