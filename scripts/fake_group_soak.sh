#!/bin/bash
# The N > 1 control flow against the RCCL test double (tests/fake_group_check.py), many rounds with other batch sizes, rays and link
# delays: races in a schedule are a matter of timing.   bash scripts/fake_group_soak.sh [seconds]   (GPU box, repo root)
cd "$(dirname "$0")/.."
make -C tests/cpp fake_rccl > /dev/null || exit 1
export VT_RCCL_LIB=$PWD/tests/cpp/_build/libfake_rccl.so VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1
budget=${1:-300}; t0=$(date +%s); round=1; bad=0
while [ $(( $(date +%s) - t0 )) -lt $budget ]; do
    for delay in 0 200 1500 6000; do
        out=$(FAKE_GROUP_ROUND=$round FAKE_RCCL_RECV_DELAY_US=$delay timeout 600 python3 tests/fake_group_check.py 2>&1 | grep -E "fake group|Error|error" | tail -3)
        case "$out" in *"fake group: ok"*) ;; *) bad=$((bad + 1)); echo "ROUND $round delay $delay: $out";; esac
        round=$((round + 1))
    done
done
echo "fake group soak: $((round - 1)) rounds, $bad failed, $(( $(date +%s) - t0 )) s"
[ $bad -eq 0 ]
