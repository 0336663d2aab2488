#!/bin/bash
# Round-6 evidence run (GPU box, repo root), ONCE on the final tree: one bench line + live PMC passes per BASELINE config (the new
# `busy` pass included: VALU busy from counters), the headline kernel's statistics from a kernel TRACE of the headline batch only,
# the round's own measurements (bounce loop without host round trips, group-wide upload / refit, fault-injection and mutant runs are
# separate calls).     bash scripts/profile_r6.sh      -> gpurun_out/r6/*  (copied into profiles/r6/ by hand afterwards)
cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/r6; mkdir -p $OUT; export TMPDIR=/tmp
line() { local name=$1; shift; timeout 1500 python3 bench.py "$@" 2> $OUT/$name.log | grep '^{' | tail -1 > $OUT/${name}_bench_line.json; echo "$name: $(cut -c1-160 $OUT/${name}_bench_line.json)"; }
# the S10M line first: its PMC file is what the headline line's beyond_cache leg cites
line s10m --scene S10M --steps 100 --alt-builder none --rebuild-leg on
mkdir -p profiles/r6; cp gpurun_out/pmc_S10M_bounce16777216_sah.json profiles/r6/ 2>/dev/null
line headline                                                     # configs[2]: 16 Mi bounce rays into S1M (the headline), all legs incl. rebuild
line config2 --scene S100k --kind primary --side 1024 --steps 2000 --alt-builder none           # configs[1]
line config4 --kind shadow --steps 100 --alt-builder none --pmc-timeout 600                     # configs[3]: 64 Mi any-hit rays
line primary_s1m --kind primary --alt-builder none --no-cpu
line alpha30 --alpha-frac 0.3 --steps 200 --alt-builder none
line strong_s10m_128tiles --scaling strong --scene S10M --tiles 128 --steps 20 --warmup 2 --no-cpu --alt-builder none   # configs[4], N = 1 point
line ploc --builder ploc --alt-builder none --no-cpu --legs off
line forcedist_1rank --force-dist --no-cpu --no-pmc --alt-builder none --steps 200
# per-workload kernel statistics: a kernel TRACE of a run that launches the headline batch and nothing else on that kernel
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --alt-builder none --legs off > $OUT/trace.log 2>&1
python3 scripts/kernel_stats_headline.py $OUT/trace $OUT/kernel_stats_headline.csv --warmup 5 --steps 20
# ... and rocprofv3's own roll-up of the same command, for comparison (round 5's file: averages the camera pass in)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --alt-builder none --legs off > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats $OUT/trace
( export VT_RCCL_LIB=$R/tests/cpp/_build/libfake_rccl.so VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1
  timeout 600 python3 scripts/group_update_rate.py 2>&1 | grep -vE 'amdgpu.ids|TEST HOOK' > $OUT/group_update_rate.txt )
for sc in S1M terrain; do timeout 300 python3 scripts/bounce_loop_rate.py --scene $sc 2>&1 | grep -v amdgpu.ids; done > $OUT/bounce_loop_rate.txt
cp gpurun_out/pmc_*.json $OUT/ 2>/dev/null
timeout 300 tests/cpp/_build/test_binding --bench > $OUT/binding_bench.txt 2>&1
cat $OUT/kernel_stats_headline.csv | head -3 | cut -c1-220
