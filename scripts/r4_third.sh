#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_third; mkdir -p $O; export TMPDIR=/tmp
line() { local name=$1; shift; timeout 900 python3 bench.py "$@" 2> $O/$name.log | grep '^{' | tail -1 > $O/${name}_bench_line.json; echo "$name: $(cut -c1-200 $O/${name}_bench_line.json)"; }
line forcedist_1rank --force-dist --no-cpu --no-pmc --alt-builder none --steps 200
line forcedist_1rank_chunks4 --force-dist --no-cpu --no-pmc --alt-builder none --steps 200 --chunks 4
line forcedist_strong_16tiles --force-dist --scaling strong --scene S10M --tiles 16 --steps 40 --warmup 2 --no-cpu --no-pmc --alt-builder none
python scripts/merged_launch_rate.py 2>&1 | tee $O/merged_s100k_primary.txt
python scripts/merged_launch_rate.py --scene S1M --side 512 2>&1 | tee $O/merged_s1m_primary_512.txt
