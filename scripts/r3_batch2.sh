#!/bin/bash
# round 3, GPU batch 2: what does an extra memory request cost?  (dup = L1-hitting duplicates of the record DMA, pf0 = the
# prefetch instruction aimed at record 0, mask = no dummy fetches for idle lanes) + vector-L1 stall counters of the base kernel
mkdir -p gpurun_out/r3b2; O=gpurun_out/r3b2; export TMPDIR=/tmp
bash scripts/ab_env.sh "S1M:bounce,S1M:primary" 2 base dup@dup pf0@pf0 pf@pf mask@mask 2>&1 | tee $O/ab_requests.txt
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?.*TCP_" | head -80 > $O/tcp_counters.txt
rocprofv3 -L > $O/all_counters.txt 2>&1
pass() { local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$name -- python3 scripts/kernel_time.py --work S1M:bounce --reps 2 > $O/pmc_$name.log 2>&1; echo "pass $name rc $?"; }
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum
pass tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass grbm GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD
python3 scripts/pmc_summary.py $O "trace_kernel<false, false" | tee $O/tcp_summary.txt
