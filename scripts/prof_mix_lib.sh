#!/bin/bash
# Two counter groups (FP64 mix, lane occupancy) of the search kernels for one build of the library.
# usage (through gpurun): bash scripts/prof_mix_lib.sh TAG [path/to/lib.so]
TAG=$1; export FREGRID_HIP_LIB=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
run() { local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/${TAG}_mix_$name -o p -- python3 scripts/prof_step.py 3 legacy > $R/${TAG}_mix_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $R/${TAG}_mix_$name.log; exit 1; }
}
run f64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 &&
run cyc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_BRANCH &&
python3 scripts/summarize_mix.py $R ${TAG} > $R/${TAG}_mix_summary.txt && grep "k_clip_quad" $R/${TAG}_mix_summary.txt | cut -c1-400
