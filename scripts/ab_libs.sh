#!/bin/bash
# A/B two builds of the library on one box: scripts/ab_libs.sh <libA.so> <libB.so> [sweep.py args...]
# (each build in its own process via VISTRACE_HIP_LIB, alternating, 3 rounds)
A=$1; B=$2; shift 2
for r in 1 2 3; do
  for L in "$A" "$B"; do
    echo -n "$(basename $L) : "
    VISTRACE_HIP_LIB=$L timeout 300 python scripts/sweep.py --rounds 5 "$@" 2>&1 | grep median
  done
done
