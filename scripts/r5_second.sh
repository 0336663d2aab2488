#!/bin/bash
# round 5, second GPU call: the GPU suite, rebuild leg (S1M + S10M), simulated groups, group update rate
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5b; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
timeout 900 python3 bench.py --scene S10M --steps 10 --warmup 3 --no-pmc --no-cpu --alt-builder none --legs off --rebuild-leg on > $O/bench_s10m_rebuild.json 2> $O/bench_s10m_rebuild.err
timeout 600 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu --no-pmc --alt-builder none --legs off > $O/bench_bare2.json 2> $O/bench_bare2.err; echo "rc $?" >> $O/bench_bare2.err
export VT_RCCL_LIB=$PWD/tests/cpp/_build/libfake_rccl.so VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1
for N in 1 2 8; do
  D=$(python3 -c "print(','.join(['0']*$N))")
  timeout 900 python3 bench.py --form group --gpus $N --group-devices $D --steps 10 --warmup 3 > $O/group_sim_n$N.json 2> $O/group_sim_n$N.err; echo "rc $?" >> $O/group_sim_n$N.err
done
timeout 900 python3 bench.py --form group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --scaling strong --scene S10M --tiles 128 --steps 3 --warmup 1 > $O/group_sim_strong8.json 2> $O/group_sim_strong8.err; echo "rc $?" >> $O/group_sim_strong8.err
timeout 600 python3 scripts/group_update_rate.py > $O/group_update_rate.txt 2>&1
tail -3 $O/pytest.log
