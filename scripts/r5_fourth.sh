#!/bin/bash
# round 5, fourth GPU call: rebuild tests with the fence-free lin_counts, upload rate + kernel stats, fake group, a 12-minute soak
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5d; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_rebuild.py tests/test_gpu_fake_group.py -m gpu -q -x > $O/pytest_rebuild.log 2>&1; echo "rc $?" >> $O/pytest_rebuild.log
python3 scripts/upload_tree_rate.py S1M S10M 2>&1 | grep -v amdgpu > $O/upload_tree_rate.txt
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_upload -- python3 $R/scripts/upload_tree_rate.py S1M S10M > $O/prof_upload.log 2>&1)
find $O/prof_upload -name "*kernel_stats.csv" -exec cp {} $O/upload_kernel_stats.csv \;
rm -rf $O/prof_upload
timeout 900 python3 scripts/soak_parity.py 500000 100000 720 > $O/soak.log 2>&1; echo "rc $?" >> $O/soak.log
tail -3 $O/pytest_rebuild.log; tail -2 $O/soak.log; cat $O/upload_tree_rate.txt | cut -c1-200
