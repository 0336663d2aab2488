#!/bin/bash
# Extra SQ busy/instruction-mix counters for the bench workload (separate passes, each under its own timeout).
# Usage (GPU box, repo root): bash scripts/profile_sq.sh <tag>
TAG=${1:-sq}; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu"
pass() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1; }
pass busy1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_CYCLES
pass busy2 SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM
pass mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VSKIPPED
python3 scripts/pmc_summary.py $OUT
