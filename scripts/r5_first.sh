#!/bin/bash
# round 5, first GPU call: the GPU suite, the driver's N = 1 command, the bare N = 2 command (gloo test mode), the group form simulated
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5a
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r5a/build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a/pytest.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a/bench_n1.json 2> gpurun_out/r5a/bench_n1.err
timeout 600 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu --no-pmc --alt-builder none --legs off > gpurun_out/r5a/bench_bare2.json 2> gpurun_out/r5a/bench_bare2.err; echo "rc $?" >> gpurun_out/r5a/bench_bare2.err
export VT_RCCL_LIB=$PWD/tests/cpp/_build/libfake_rccl.so VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1
for N in 1 2 8; do
  D=$(python3 -c "print(','.join(['0']*$N))")
  timeout 900 python3 bench.py --form group --gpus $N --group-devices $D --steps 10 --warmup 3 > gpurun_out/r5a/group_sim_n$N.json 2> gpurun_out/r5a/group_sim_n$N.err; echo "rc $?" >> gpurun_out/r5a/group_sim_n$N.err
done
timeout 900 python3 bench.py --form group --gpus 8 --group-devices 0,0,0,0,0,0,0,0 --scaling strong --scene S10M --tiles 128 --steps 3 --warmup 1 > gpurun_out/r5a/group_sim_strong8.json 2> gpurun_out/r5a/group_sim_strong8.err; echo "rc $?" >> gpurun_out/r5a/group_sim_strong8.err
tail -3 gpurun_out/r5a/pytest.log
