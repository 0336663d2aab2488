#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5e; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
bash scripts/alpha_threshold_sweep.sh > $O/alpha_threshold_sweep.txt 2>&1
cat $O/alpha_threshold_sweep.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "alpha" > $O/pytest_alpha.log 2>&1; tail -2 $O/pytest_alpha.log
