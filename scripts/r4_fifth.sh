#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_fifth; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python3 bench.py --scene S10M --steps 100 --alt-builder none --no-cpu 2> $O/s10m.log | grep '^{' | tail -1 > $O/s10m_bench_line.json ) 2>&1 | grep real
mkdir -p profiles/r4; cp gpurun_out/pmc_S10M_bounce16777216_sah.json profiles/r4/ 2>/dev/null
( time timeout 900 python3 bench.py 2> $O/default.log | grep '^{' | tail -1 > $O/default_bench_line.json ) 2>&1 | grep real
cp gpurun_out/pmc_*.json $O/
tail -5 $O/default.log
