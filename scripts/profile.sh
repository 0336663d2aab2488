#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and separate PMC passes under gpurun_out/prof_<tag>/.
# PMC passes are collected in their own runs (never combined with sys/hip traces).
TAG=${1:-r1}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
pass() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
find $OUT -name "*.csv" | head -50
