#!/bin/bash
# round 3, GPU batch 1: parity of the default build, kernel A/B (prefetch, DMA cache policy), record layouts, L2 counters
mkdir -p gpurun_out/r3b1; O=gpurun_out/r3b1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "== kernel variants, S1M" | tee $O/ab_kernel.txt
bash scripts/ab_env.sh "S1M:bounce,S1M:primary" 3 base pf@pf nt@nt pfnt@pfnt 2>&1 | tee -a $O/ab_kernel.txt
echo "== layouts, S1M" | tee $O/ab_layout.txt
bash scripts/ab_env.sh "S1M:bounce,S1M:primary" 2 base "big|VT_LAYOUT_BIG_FIRST=1" "bfs8|VT_LAYOUT_BFS_LEVELS=8" "bfs12|VT_LAYOUT_BFS_LEVELS=12" \
   "bfs16|VT_LAYOUT_BFS_LEVELS=16" "il|VT_LAYOUT_INTERLEAVE=1" "ilbig|VT_LAYOUT_INTERLEAVE=1 VT_LAYOUT_BIG_FIRST=1" \
   "ilbigbfs12|VT_LAYOUT_INTERLEAVE=1 VT_LAYOUT_BIG_FIRST=1 VT_LAYOUT_BFS_LEVELS=12" 2>&1 | tee -a $O/ab_layout.txt
echo "== S10M" | tee $O/ab_s10m.txt
bash scripts/ab_env.sh "S10M:bounce" 1 base pf@pf "il|VT_LAYOUT_INTERLEAVE=1" "ilbig|VT_LAYOUT_INTERLEAVE=1 VT_LAYOUT_BIG_FIRST=1" "pfilbig@pf|VT_LAYOUT_INTERLEAVE=1 VT_LAYOUT_BIG_FIRST=1" 2>&1 | tee -a $O/ab_s10m.txt
echo "== L2 counters" | tee $O/l2.txt
bash scripts/pmc_l2.sh base S1M:bounce 2>&1 | tee -a $O/l2.txt
bash scripts/pmc_l2.sh il S1M:bounce vistrace_amd/lib/libvistrace_hip.so VT_LAYOUT_INTERLEAVE=1 2>&1 | tee -a $O/l2.txt
bash scripts/pmc_l2.sh ilbig S1M:bounce vistrace_amd/lib/libvistrace_hip.so VT_LAYOUT_INTERLEAVE=1 VT_LAYOUT_BIG_FIRST=1 2>&1 | tee -a $O/l2.txt
bash scripts/pmc_l2.sh pf S1M:bounce vistrace_amd/lib/variants/libvistrace_hip_pf.so 2>&1 | tee -a $O/l2.txt
