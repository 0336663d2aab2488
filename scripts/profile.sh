#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag> [bench args...]
# rocprofv3 kernel-trace stats of the bench command + separate PMC passes (never combined with
# sys/hip traces), each under its own timeout.  Results land in gpurun_out/prof_<tag>/ ;
# scripts/collect_profile.py turns them into the summaries committed under profiles/.
TAG=${1:-r1}; shift
ARGS=${@:---steps 5 --warmup 1 --no-cpu}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
pass() { local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass l1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
grep -h '^{' $OUT/stats.log | tail -1 > $OUT/bench_line.json
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cat $OUT/stats/*/*_kernel_stats.csv | head -8
cat $OUT/pmc_summary.txt
