#!/bin/bash
# HBM-side traffic of another bench workload (FETCH_SIZE / WRITE_SIZE / L2 passes only), e.g. the 10 M-triangle scene:
#   bash scripts/profile_traffic.sh s10m --scene S10M
TAG=${1:-traffic}; shift
ARGS="$@ --steps 3 --warmup 1 --no-cpu"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
pass() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
grep -h '^{' $OUT/stats.log | tail -1 > $OUT/bench_line.json
python3 scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
head -3 $OUT/stats/*/*_kernel_stats.csv | cut -c1-200
cat $OUT/pmc_summary.txt
