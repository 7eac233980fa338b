#!/bin/bash
# One development cycle on the GPU box: parity tests, a short bench, a rocprofv3 kernel-trace of a few steps.
# usage (through gpurun): bash scripts/gpu_cycle.sh TAG [tests|notests] [prof_mode]
TAG=${1:-x}; DO_TESTS=${2:-tests}; MODE=${3:-legacy}
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$DO_TESTS" = tests ]; then
  timeout -k 10 500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${TAG}_t.log 2>&1; tail -3 $R/gpurun_out/${TAG}_t.log
fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --gc-steps 0 --cpu-rows 0 > $R/gpurun_out/${TAG}_b.json 2> $R/gpurun_out/${TAG}_b.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/scripts/prof_step.py 10 $MODE > $R/gpurun_out/${TAG}_prof.log 2>&1
echo cycle done
