#!/bin/bash
# Round-3 rocprofv3 passes (each counter group in its own run with --kernel-trace only).
#   bench headline (bench.py --legs none): stats -- the kernel averages the bench line's roofline quotes
#   search (prof_step.py legacy): stats, FETCH_SIZE, WRITE_SIZE, instruction mix (f64 / int / cycles)
#   gc     (prof_step.py gc):     stats, FETCH/WRITE, instruction mix
#   sweep  (prof_step.py sweep):  stats, fabric requests by size, L2
# Run on the GPU box from the repo root; outputs under gpurun_out/r03_*.  Then: python scripts/summarize_r03.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
run() { # tag mode counters...
  local tag=$1 mode=$2; shift 2
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/r03_pmc_$tag -o p -- python3 scripts/prof_step.py 3 $mode > $R/r03_pmc_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -3 $R/r03_pmc_$tag.log; exit 1; }
  echo "pmc $tag done"
}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r03_stats_bench_headline -o s -- python3 bench.py --legs none --cpu-rows 0 --gc-steps 0 > $R/r03_stats_bench_headline.log 2>&1 || exit 5
echo "stats bench headline done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r03_stats_search -o s -- python3 scripts/prof_step.py 10 legacy > $R/r03_stats_search.log 2>&1 || exit 2
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r03_stats_sweep -o s -- python3 scripts/prof_step.py 10 sweep > $R/r03_stats_sweep.log 2>&1 || exit 3
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/r03_stats_gc -o s -- python3 scripts/prof_step.py 10 gc > $R/r03_stats_gc.log 2>&1 || exit 4
echo "stats done"
run search_fetch legacy FETCH_SIZE
run search_write legacy WRITE_SIZE
run search_f64 legacy SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
run search_int legacy SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
run search_cyc legacy SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY
run gc_fetch gc FETCH_SIZE
run gc_write gc WRITE_SIZE
run gc_f64 gc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
run gc_int gc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM
run gc_cyc gc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY
run sweep_rd sweep TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run sweep_l2 sweep TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run sweep_wr sweep TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
echo "all done"
