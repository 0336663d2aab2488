#!/bin/bash
# Disassembly of one trace-kernel instantiation:  bash scripts/dump_isa.sh <out.s> [template args, default 0 0 1 1 0] [-D...]
# (ANY_HIT STATS PERSISTENT FETCH_DMA ALPHA); prints VGPR / SGPR use and writes the kernel's ISA to <out.s>
OUT=$1; shift
A=${1:-0}; S=${2:-0}; P=${3:-1}; D=${4:-1}; AL=${5:-0}; shift 5 2>/dev/null
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero \
  -fno-slp-vectorize -I$ROOT/include -I$ROOT/vistrace_amd/csrc "$@" --cuda-device-only -S -o $TMP/all.s \
  $ROOT/vistrace_amd/csrc/trace_kernels.hip -Rpass-analysis=kernel-resource-usage 2> $TMP/res.txt
if [ "$AL" = 1 ]; then SYM="_ZN2vt18trace_kernel_alphaILb${A}ELb${S}ELb${P}ELb${D}EEEvNS_9TraceArgsE"; else SYM="_ZN2vt12trace_kernelILb${A}ELb${S}ELb${P}ELb${D}ELb0EEEvNS_9TraceArgsE"; fi
grep -A8 "Function Name: $SYM" $TMP/res.txt | grep -E "SGPRs:|VGPRs:|Spill|Occupancy" | sed 's/.*remark: *//'
awk "/^$SYM:/,/^.Lfunc_end/" $TMP/all.s > $OUT
echo "$(grep -c '^\s*v_' $OUT) VALU, $(grep -c '^\s*s_' $OUT) SALU static instructions -> $OUT"
rm -rf $TMP
