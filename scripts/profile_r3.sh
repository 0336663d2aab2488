#!/bin/bash
# Round-3 evidence run (GPU box, repo root): one bench line + live PMC passes per BASELINE config, rocprofv3 kernel stats.
#   bash scripts/profile_r3.sh [quick]      -> gpurun_out/r3/*
# bench.py collects its own PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, instruction mix, TCP, TCC) in child processes and writes
# gpurun_out/pmc_<workload>_<builder>.json; the kernel-trace stats are separate runs (never combined with --pmc).
OUT=gpurun_out/r3; mkdir -p $OUT; export TMPDIR=/tmp
line() { local name=$1; shift; timeout 1500 python3 bench.py "$@" 2> $OUT/$name.log | grep '^{' | tail -1 > $OUT/${name}_bench_line.json; echo "$name: $(cut -c1-160 $OUT/${name}_bench_line.json)"; }
line headline                                                     # configs[2]: 16 Mi bounce rays into S1M (the headline)
line config2 --scene S100k --kind primary --side 1024 --steps 2000 --alt-builder none           # configs[1]
line config4 --kind shadow --steps 100 --alt-builder none --pmc-timeout 600                     # configs[3]: 64 Mi any-hit rays
line primary_s1m --kind primary --alt-builder none --no-cpu
line alpha30 --alpha-frac 0.3 --steps 200 --alt-builder none
line s10m --scene S10M --steps 100 --alt-builder none
line strong_s10m_128tiles --scaling strong --scene S10M --tiles 128 --steps 20 --warmup 2 --no-cpu --alt-builder none   # configs[4], N = 1 point
line ploc --builder ploc --alt-builder none --no-cpu
line forcedist_1rank --force-dist --no-cpu --no-pmc --alt-builder none
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-pmc --alt-builder none > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4stats -- python3 bench.py --kind shadow --steps 10 --warmup 2 --no-cpu --no-pmc --alt-builder none > $OUT/c4stats.log 2>&1
cp $OUT/c4stats/*/*_kernel_stats.csv $OUT/config4_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc_*.json $OUT/ 2>/dev/null
rm -rf $OUT/stats $OUT/c4stats
head -4 $OUT/kernel_stats.csv | cut -c1-200
