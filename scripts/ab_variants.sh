#!/bin/bash
# A/B of library variants on one box, each in its own process, interleaved rounds:
#   bash scripts/ab_variants.sh "S1M:bounce,S1M:primary" 3 base x y     (names under vistrace_amd/lib/variants/, "base" = the product lib)
WORK=$1; ROUNDS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = base ]; then L=vistrace_amd/lib/libvistrace_hip.so; else L=vistrace_amd/lib/variants/libvistrace_hip_$v.so; fi
    VISTRACE_HIP_LIB=$PWD/$L timeout 300 python scripts/kernel_time.py --work "$WORK" --tag $v 2>&1 | grep -E "median|Error|error" 
  done
done
