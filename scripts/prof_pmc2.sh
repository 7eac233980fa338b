#!/bin/bash
# Round-2 rocprofv3 passes (each counter group in its own run with --kernel-trace only; TCC has 4 slots per pass).
#   search (prof_step.py legacy): stats, FETCH_SIZE, WRITE_SIZE, SQ group
#   gc     (prof_step.py gc):     stats, SQ group, FETCH/WRITE
#   bench.py: stats of the default run and of the headline-only run (--legs none)
#   sweep  (prof_step.py sweep):  stats, FETCH_SIZE, WRITE_SIZE, the raw TCC request counters by size, L2 hit / miss
# Run on the GPU box from the repo root; outputs under gpurun_out/r02_*.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
run() { # tag mode counters...
  local tag=$1 mode=$2; shift 2
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/r02_pmc_$tag -o p -- python3 scripts/prof_step.py 3 $mode > $R/r02_pmc_$tag.log 2>&1 || { echo "pmc $tag failed"; exit 1; }
  echo "pmc $tag done"
}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r02_stats_search -o s -- python3 scripts/prof_step.py 10 legacy > $R/r02_stats_search.log 2>&1 || exit 2
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r02_stats_sweep -o s -- python3 scripts/prof_step.py 10 sweep > $R/r02_stats_sweep.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r02_stats_gc -o s -- python3 scripts/prof_step.py 10 gc > $R/r02_stats_gc.log 2>&1 || exit 4
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r02_stats_bench_headline -o s -- python3 bench.py --legs none --cpu-rows 0 --gc-steps 0 > $R/r02_stats_bench_headline.log 2>&1 || exit 5
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r02_stats_bench -o s -- python3 bench.py > $R/r02_stats_bench.log 2>&1 || exit 6
run gc_sq gc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run gc_fetch gc FETCH_SIZE
run gc_write gc WRITE_SIZE
run search_fetch legacy FETCH_SIZE
run search_write legacy WRITE_SIZE
run search_sq legacy SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sweep_fetch sweep FETCH_SIZE
run sweep_write sweep WRITE_SIZE
run sweep_rd sweep TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run sweep_l2 sweep TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run sweep_wr sweep TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
run sweep_sq sweep SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
echo "all done"
