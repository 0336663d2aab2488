#!/bin/bash
# SQ counters of one library variant on S1M bounce (GPU box):  bash scripts/pmc_variant.sh <tag> <lib.so> [kernel_time.py args]
TAG=$1; LIB=$2; shift 2
OUT=gpurun_out/pmcv_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export VISTRACE_HIP_LIB=$PWD/$LIB
EXTRA="$@"
pass() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 scripts/kernel_time.py --work S1M:bounce --reps 2 $EXTRA > $OUT/pmc_$name.log 2>&1; }
pass sq SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
python3 scripts/pmc_summary.py $OUT "trace_kernel<false, false" | sed "s/^/$TAG /"
