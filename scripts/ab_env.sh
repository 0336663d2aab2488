#!/bin/bash
# A/B of (library variant, environment) combinations on one box, each run in its own process, interleaved rounds:
#   bash scripts/ab_env.sh "S1M:bounce,S1M:primary" 2 "base" "big|VT_LAYOUT_BIG_FIRST=1" "pf@pf" "pfbig@pf|VT_LAYOUT_BIG_FIRST=1"
# entry = tag[@variant][|ENV=V ENV=V ...]; variant names a library under vistrace_amd/lib/variants/, default the product's.
# Extra arguments for kernel_time.py come from $KT_ARGS.
WORK=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for entry in "$@"; do
    head=${entry%%|*}; envs=""; [ "$entry" != "$head" ] && envs=${entry#*|}
    tag=${head%%@*}; var=""; [ "$head" != "$tag" ] && var=${head#*@}
    if [ -z "$var" ]; then L=vistrace_amd/lib/libvistrace_hip.so; else L=vistrace_amd/lib/variants/libvistrace_hip_$var.so; fi
    env $envs VISTRACE_HIP_LIB=$PWD/$L timeout 600 python scripts/kernel_time.py --work "$WORK" --tag $tag $KT_ARGS 2>&1 | grep -E "median|Error|error"
  done
done
