#!/bin/bash
# rocprofv3 kernel-trace statistics for profiles/: the bench command and the great-circle profiling target.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
mkdir -p $R/keep
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats -o bench -- python3 bench.py --steps 20 --warmup 3 --cpu-rows 0 --gc-steps 0 > $R/bench_prof.json 2> $R/bench_prof.err || exit 2
cp $R/prof_stats/bench_kernel_stats.csv $R/keep/
echo "stats done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats_gc -o gc -- python3 scripts/prof_step.py 5 gc > $R/gc_prof.log 2>&1 || exit 3
cp $R/prof_stats_gc/gc_kernel_stats.csv $R/keep/
echo "gc stats done"
