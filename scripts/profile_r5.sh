#!/bin/bash
# Round-5 evidence run (GPU box, repo root), ONCE on the final tree: one bench line + live PMC passes per BASELINE config, rocprofv3
# kernel stats, the round's own measurements (Rebuild on the device, the single-process group on a simulated group, group-wide updates).
#     bash scripts/profile_r5.sh      -> gpurun_out/r5/*  (copied into profiles/r5/ by hand afterwards)
cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/r5; mkdir -p $OUT; export TMPDIR=/tmp
line() { local name=$1; shift; timeout 1500 python3 bench.py "$@" 2> $OUT/$name.log | grep '^{' | tail -1 > $OUT/${name}_bench_line.json; echo "$name: $(cut -c1-160 $OUT/${name}_bench_line.json)"; }
# the S10M line first: its PMC file is what the headline line's beyond_cache leg cites; with the Rebuild leg of the 10 M-triangle scene
line s10m --scene S10M --steps 100 --alt-builder none --rebuild-leg on
mkdir -p profiles/r5; cp gpurun_out/pmc_S10M_bounce16777216_sah.json profiles/r5/ 2>/dev/null
line headline                                                     # configs[2]: 16 Mi bounce rays into S1M (the headline), all legs incl. rebuild
line config2 --scene S100k --kind primary --side 1024 --steps 2000 --alt-builder none           # configs[1]
line config4 --kind shadow --steps 100 --alt-builder none --pmc-timeout 600                     # configs[3]: 64 Mi any-hit rays
line primary_s1m --kind primary --alt-builder none --no-cpu
line alpha30 --alpha-frac 0.3 --steps 200 --alt-builder none
line strong_s10m_128tiles --scaling strong --scene S10M --tiles 128 --steps 20 --warmup 2 --no-cpu --alt-builder none   # configs[4], N = 1 point
line ploc --builder ploc --alt-builder none --no-cpu --legs off
line forcedist_1rank --force-dist --no-cpu --no-pmc --alt-builder none --steps 200
line forcedist_strong_16tiles --force-dist --scaling strong --scene S10M --tiles 16 --steps 40 --warmup 2 --no-cpu --no-pmc --alt-builder none
# the bare N > 1 command (no launcher): bench.py starts its own ranks; gloo test mode, both ranks on this GPU
timeout 600 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu --no-pmc --alt-builder none --legs off --side 1024 2> $OUT/bare2.log | grep '^{' | tail -1 > $OUT/bare_gpus2_gloo_bench_line.json
# the single-process group (vt_engine_open_multi + vt_trace_closest_gather_dev) on a SIMULATED group: members share device 0, RCCL = test double
( export VT_RCCL_LIB=$R/tests/cpp/_build/libfake_rccl.so VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1
  for N in 2 8; do
    D=$(python3 -c "print(','.join(['0']*$N))")
    timeout 900 python3 bench.py --form group --gpus $N --group-devices $D --steps 10 --warmup 3 2> $OUT/group_form_sim_n$N.log | grep '^{' | tail -1 > $OUT/group_form_sim_n${N}_bench_line.json
  done
  timeout 900 python3 bench.py --form group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --scaling strong --scene S10M --tiles 128 --steps 3 --warmup 1 2> $OUT/group_form_sim_strong8.log | grep '^{' | tail -1 > $OUT/group_form_sim_strong8_bench_line.json
  timeout 600 python3 scripts/group_update_rate.py 2>&1 | grep -vE 'amdgpu.ids|TEST HOOK' > $OUT/group_update_rate.txt )
timeout 300 python3 bench.py --form group --gpus 1 --steps 20 --warmup 3 2> $OUT/group_form_n1.log | grep '^{' | tail -1 > $OUT/group_form_n1_real_rccl_bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 2 --no-cpu --no-pmc --alt-builder none --legs off > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/upstats -- python3 scripts/upload_tree_rate.py S1M S10M > $OUT/upload_tree_rate_under_profiler.txt 2>&1
cp $OUT/upstats/*/*_kernel_stats.csv $OUT/upload_tree_kernel_stats.csv 2>/dev/null
python3 scripts/upload_tree_rate.py S1M S10M 2>&1 | grep -v amdgpu.ids > $OUT/upload_tree_rate.txt
cp gpurun_out/pmc_*.json $OUT/ 2>/dev/null
rm -rf $OUT/stats $OUT/upstats
timeout 300 tests/cpp/_build/test_binding --bench > $OUT/binding_bench.txt 2>&1
head -4 $OUT/kernel_stats.csv | cut -c1-200
