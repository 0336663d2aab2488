#!/bin/bash
# L2 hit / miss and fabric bytes of one (library, environment) combination (GPU box):
#   bash scripts/pmc_l2.sh <tag> <work, e.g. S1M:bounce> [lib.so] [ENV=V ...]
# Separate --pmc passes, never combined with other trace domains; summary by scripts/pmc_summary.py.
TAG=$1; WORK=$2; LIB=${3:-vistrace_amd/lib/libvistrace_hip.so}; shift 3
OUT=gpurun_out/l2_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export VISTRACE_HIP_LIB=$PWD/$LIB
for kv in "$@"; do case "$kv" in *=*) export "$kv";; esac; done
pass() { local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 scripts/kernel_time.py --work $WORK --reps 2 > $OUT/pmc_$name.log 2>&1; }
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 scripts/pmc_summary.py $OUT "trace_kernel<false, false" | sed "s/^/$TAG /"
