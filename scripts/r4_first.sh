#!/bin/bash
# round 4, first GPU call: the new merged-launch tests, the whole GPU suite, same-box A/B of the round-3 library against the tree
cd "$(dirname "$0")/.."
O=gpurun_out/r4_first; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi_batch.py -x -q -m gpu 2>&1 | tail -15 | tee $O/multi_batch.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
export VT_ALLOW_OLDER_ABI=1
bash scripts/ab_variants.sh "S1M:bounce,S1M:primary,S100k:primary" 3 r3 base 2>&1 | tee $O/ab_r3_vs_tree.txt
