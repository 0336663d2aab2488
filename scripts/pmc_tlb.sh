#!/bin/bash
# Address-translation and cache counters of the trace kernel on one workload (GPU box):  bash scripts/pmc_tlb.sh S10M:bounce
W=${1:-S10M:bounce}; TAG=$(echo $W | tr ':' '_')
OUT=gpurun_out/pmctlb_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
pass() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 scripts/kernel_time.py --work $W --reps 2 > $OUT/pmc_$name.log 2>&1; }
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
python3 scripts/pmc_summary.py $OUT "trace_kernel<false, false" | sed "s/^/$TAG /"
