#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r4_sixth; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
timeout 300 tests/cpp/_build/test_binding --bench 2>&1 | tee $O/binding_bench.txt | tail -5
