#!/bin/bash
# round 5, closing call: the whole GPU suite, smoke, a short soak and the evidence run on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5h; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 500 python3 scripts/soak_parity.py 900000 100000 360 > $O/soak.log 2>&1; echo "rc $?" >> $O/soak.log
bash scripts/profile_r5.sh > $O/profile.log 2>&1
grep -E "passed|failed|rc " $O/pytest.log | tail -3; tail -2 $O/smoke.log; tail -2 $O/soak.log; tail -16 $O/profile.log | cut -c1-150
