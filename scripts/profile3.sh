#!/bin/bash
# SQ/LDS issue counters for the current default kernel. Usage: bash scripts/profile3.sh <tag> [bench args]
TAG=${1:-sq}; shift
ARGS=${@:---steps 2 --warmup 0 --no-cpu}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1; }
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pass sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_WAVES SQ_BUSY_CYCLES
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
