#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_seventh; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "page_locked or host_buffer" 2>&1 | tail -5
( time timeout 900 python3 bench.py --no-pmc 2> $O/default.log | grep '^{' | tail -1 > $O/default_bench_line.json ) 2>&1 | grep real
python3 -c "
import json; d=json.load(open('$O/default_bench_line.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['host_inclusive'], indent=1))"
