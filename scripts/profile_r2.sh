#!/bin/bash
# Round-2 evidence run (GPU box, repo root): bench lines + rocprofv3 kernel stats of the same commands.
#   bash scripts/profile_r2.sh            -> gpurun_out/r2/*
# bench.py collects its own PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, instruction mix) in child processes and writes
# gpurun_out/pmc_<workload>.json; the kernel-trace stats below are separate runs (never combined with --pmc).
OUT=gpurun_out/r2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.log
timeout 600 python3 bench.py --scene S10M --steps 100 > $OUT/s10m_bench_line.json 2> $OUT/s10m_bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-pmc --alt-builder none > $OUT/stats.log 2>&1
timeout 600 python3 bench.py --builder ploc --alt-builder none --no-cpu > $OUT/ploc_bench_line.json 2> $OUT/ploc_bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s10m_stats -- python3 bench.py --scene S10M --steps 20 --warmup 2 --no-cpu --no-pmc --alt-builder none > $OUT/s10m_stats.log 2>&1
timeout 300 python3 bench.py --force-dist --no-cpu --no-pmc --alt-builder none 2> $OUT/forcedist.log | grep '^{' | tail -1 > $OUT/forcedist_1rank_bench_line.json
timeout 600 python3 bench.py --scaling strong --scene S10M --tiles 16 --steps 100 --no-cpu --no-pmc --alt-builder none 2> $OUT/strong.log | grep '^{' | tail -1 > $OUT/strong_s10m_16tiles_bench_line.json
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cp $OUT/s10m_stats/*/*_kernel_stats.csv $OUT/s10m_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc_S1M_bounce16777216_*.json gpurun_out/pmc_S10M_bounce16777216_*.json $OUT/ 2>/dev/null
rm -rf $OUT/stats $OUT/s10m_stats
head -4 $OUT/kernel_stats.csv | cut -c1-220
head -4 $OUT/s10m_kernel_stats.csv | cut -c1-220
