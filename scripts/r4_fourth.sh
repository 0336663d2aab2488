#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_fourth; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batch" 2>&1 | tail -8 | tee $O/batch_tests.txt
timeout 600 python -m pytest tests/test_host_binding.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O/batch_tests.txt
for i in 1 2; do timeout 300 tests/cpp/_build/test_binding --bench 2>&1 | tee $O/binding_bench_$i.txt; done
