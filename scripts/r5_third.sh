#!/bin/bash
# round 5, third GPU call: where vt_scene_upload_tree's time goes; A/B of the neighbour-fetch experiment; group update host times
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out/r5c; mkdir -p $O
python3 scripts/upload_tree_rate.py S1M S10M > $O/upload_tree_rate.txt 2>&1
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_upload -- python3 $R/scripts/upload_tree_rate.py S1M S10M > $O/prof_upload.log 2>&1)
find $O/prof_upload -name "*kernel_stats.csv" -exec cp {} $O/upload_kernel_stats.csv \;
rm -rf $O/prof_upload
bash scripts/ab_variants.sh "S1M:bounce,S10M:bounce,S1M:primary" 3 base nb > $O/ab_nb.txt 2>&1
for v in base nb; do
  if [ $v = base ]; then L=vistrace_amd/lib/libvistrace_hip.so; else L=vistrace_amd/lib/variants/libvistrace_hip_$v.so; fi
  VISTRACE_HIP_LIB=$R/$L timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu --alt-builder none --legs off --pmc-passes tcp,sq > $O/pmc_$v.json 2> $O/pmc_$v.err
done
VT_RCCL_LIB=$R/tests/cpp/_build/libfake_rccl.so VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1 timeout 600 python3 scripts/group_update_rate.py > $O/group_update_rate.txt 2>&1
cat $O/ab_nb.txt
