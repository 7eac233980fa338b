#!/bin/bash
# rocprofv3 PMC passes for profiles/ (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC has 4 counters, they cost 3 + 2) (each in its own run, kernel-trace only)
# on scripts/prof_step.py.  Run on the GPU box from the repo root; outputs under gpurun_out/prof_*.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/prof_pmc$i -o p -- python3 scripts/prof_step.py 3 legacy > $R/pmc$i.log 2>&1 || exit $((10+i))
  echo "pmc group $i done"
done
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/prof_pmc_gc$i -o p -- python3 scripts/prof_step.py 3 gc > $R/pmc_gc$i.log 2>&1 || exit $((20+i))
  echo "gc pmc group $i done"
done
echo "all done"
