#!/bin/bash
# Instruction mix of the search kernels (round 3): what the clip's VALU time is made of.
# Each counter group in its own rocprofv3 run with --kernel-trace only (no other trace domain) on scripts/prof_step.py.
# usage (through gpurun): bash scripts/prof_mix.sh [TAG] [mode]     outputs under gpurun_out/<TAG>_mix_<group>/
TAG=${1:-r03}; MODE=${2:-legacy}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
R=gpurun_out
run() { # name counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/${TAG}_mix_$name -o p -- python3 scripts/prof_step.py 3 $MODE > $R/${TAG}_mix_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $R/${TAG}_mix_$name.log; exit 1; }
  echo "pmc $name done"
}
run f64 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 &&
run int SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM &&
run cyc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_BRANCH &&
run act SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
python3 scripts/summarize_mix.py $R ${TAG} > $R/${TAG}_mix_summary.txt && cat $R/${TAG}_mix_summary.txt
