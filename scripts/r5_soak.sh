#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5g; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1700 python3 scripts/soak_parity.py 700000 100000 1500 > $O/soak.log 2>&1; echo "rc $?" >> $O/soak.log
timeout 700 bash scripts/fake_group_soak.sh 540 > $O/fake_group_soak.log 2>&1; echo "rc $?" >> $O/fake_group_soak.log
tail -2 $O/soak.log; tail -3 $O/fake_group_soak.log
