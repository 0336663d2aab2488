#!/bin/bash
# Round-4 evidence run (GPU box, repo root): one bench line + live PMC passes per BASELINE config, rocprofv3 kernel stats, the
# round's side measurements.     bash scripts/profile_r4.sh      -> gpurun_out/r4/*  (copy what is to be judged into profiles/r4/)
# bench.py collects its own PMC passes (FETCH_SIZE, WRITE_SIZE, SQ, instruction mix, TCP, TCC) in child processes and writes
# gpurun_out/pmc_<workload>_<builder>.json; the kernel-trace stats are separate runs (never combined with --pmc).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4; mkdir -p $OUT; export TMPDIR=/tmp
line() { local name=$1; shift; timeout 1500 python3 bench.py "$@" 2> $OUT/$name.log | grep '^{' | tail -1 > $OUT/${name}_bench_line.json; echo "$name: $(cut -c1-160 $OUT/${name}_bench_line.json)"; }
# the S10M line first: its PMC file is what the headline line's beyond_cache leg cites
line s10m --scene S10M --steps 100 --alt-builder none
mkdir -p profiles/r4; cp gpurun_out/pmc_S10M_bounce16777216_sah.json profiles/r4/ 2>/dev/null
line headline                                                     # configs[2]: 16 Mi bounce rays into S1M (the headline), all legs
line config2 --scene S100k --kind primary --side 1024 --steps 2000 --alt-builder none           # configs[1]
line config4 --kind shadow --steps 100 --alt-builder none --pmc-timeout 600                     # configs[3]: 64 Mi any-hit rays
line primary_s1m --kind primary --alt-builder none --no-cpu
line alpha30 --alpha-frac 0.3 --steps 200 --alt-builder none
line strong_s10m_128tiles --scaling strong --scene S10M --tiles 128 --steps 20 --warmup 2 --no-cpu --alt-builder none   # configs[4], N = 1 point
line ploc --builder ploc --alt-builder none --no-cpu --legs off
line forcedist_1rank --force-dist --no-cpu --no-pmc --alt-builder none --steps 200
line forcedist_strong_16tiles --force-dist --scaling strong --scene S10M --tiles 16 --steps 40 --warmup 2 --no-cpu --no-pmc --alt-builder none
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-pmc --alt-builder none --legs off > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4stats -- python3 bench.py --kind shadow --steps 10 --warmup 2 --no-cpu --no-pmc --alt-builder none --legs off > $OUT/c4stats.log 2>&1
cp $OUT/c4stats/*/*_kernel_stats.csv $OUT/config4_kernel_stats.csv 2>/dev/null
cp gpurun_out/pmc_*.json $OUT/ 2>/dev/null
rm -rf $OUT/stats $OUT/c4stats
python3 scripts/merged_launch_rate.py > $OUT/merged_s100k_primary.txt 2>&1
python3 scripts/merged_launch_rate.py --scene S1M --side 512 > $OUT/merged_s1m_primary_512.txt 2>&1
python3 scripts/merged_launch_rate.py --scene S1M --side 1024 --kind bounce > $OUT/merged_s1m_bounce_1024.txt 2>&1
timeout 300 tests/cpp/_build/test_binding --bench > $OUT/binding_bench.txt 2>&1
timeout 300 python3 scripts/shading_frame_rate.py 2>&1 | grep -vE 'RCCL|NCCL|amdgpu.ids' > $OUT/shading_frame.txt
head -4 $OUT/kernel_stats.csv | cut -c1-200
