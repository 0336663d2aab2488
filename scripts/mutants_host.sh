#!/bin/bash
# Mutation testing of the CPU tests of the single-ray host walk (vistrace_amd/csrc/host_walk.cpp: what ONE accel:Traverse runs,
# BASELINE configs[0]).  No GPU: builds each mutant here and runs the host-walk and binding tests against it.
#   bash scripts/mutants_host.sh [out.txt] [mutant numbers...]
main() {
OUT=${1:-profiles/r6/mutants_host.txt}; shift
KS=${@:-1 2 4 5 6 7 8 11 12 13 14 15}
declare -A WHAT=([1]="near/far swap on >=" [2]="hit accepted on t < tmax" [4]="leaf slots from the back" [5]="plain 1/x instead of safe_inverse"
 [6]="slab entry without the tmin term" [7]="node accepted on first < second" [8]="hit needs u > 0" [11]="hit accepted on t > tmin"
 [12]="w = 1 - (u + v)" [13]="FRONT faces culled" [14]="hit needs v > 0" [15]="hit needs w > 0")
cd "$(dirname "$0")/.."
TESTS="tests/test_host_walk.py tests/test_oracle_golden.py tests/test_host_binding.py"
{ echo "# mutation testing of the CPU tests of the host walk (host_walk.cpp), $(date -u +%Y-%m-%dT%H:%MZ)"
  echo "# a mutant library = the product's objects with host_walk.cpp compiled -DVT_MUTANT=<k>; tests: $TESTS -m 'not gpu'"; } > "$OUT"
if timeout 600 python -m pytest $TESTS -m "not gpu" -x -q -p no:cacheprovider > /tmp/hm_ctl.log 2>&1; then
  echo "control   product library                              GREEN  ($(grep -E ' passed' /tmp/hm_ctl.log | tail -1))" >> "$OUT"
else echo "control   product library                              RED" >> "$OUT"; fi
for k in $KS; do
  make -C vistrace_amd/csrc host_mutant K=$k > /tmp/hm_build_$k.log 2>&1 || { echo "mutant $k: build failed" >> "$OUT"; continue; }
  L=$PWD/vistrace_amd/lib/variants/libvistrace_hip_hostmut_$k.so
  verdict=SURVIVED; by=""
  if ! VISTRACE_HIP_LIB=$L timeout 600 python -m pytest $TESTS -m "not gpu" -x -q -p no:cacheprovider > /tmp/hm_$k.log 2>&1; then
    by=$(grep -m1 -E "^(FAILED|ERROR) " /tmp/hm_$k.log | sed -E 's/ - .*//'); verdict=KILLED
  fi
  printf "host mutant %-2s %-40s %-8s %s\n" "$k" "${WHAT[$k]}" "$verdict" "$by" >> "$OUT"
  rm -rf vistrace_amd/csrc/_build_hostmut_$k "$L"
done
# the host TraceResult class (vistrace_amd/csrc/host/TraceResult.cpp: the fields one accel:Traverse hands to Lua) against the oracle:
# tests/cpp/test_trace_result.cpp rebuilt with the class compiled -DVT_MUTANT=<k>
make -C tests/cpp trace_result > /dev/null 2>&1
echo "# host TraceResult class: tests/cpp/test_trace_result (class vs oracle, bit for bit) with TraceResult.cpp compiled -DVT_MUTANT=<k>" >> "$OUT"
echo "control   product class                                $(tests/cpp/_build/test_trace_result | tail -1)" >> "$OUT"
declare -A TWHAT=([41]="uvw.z = 1 - (u + v)" [42]="frontFacing on dot > 0" [43]="CalcTBN correction on cosTheta < 0.1" [44]="mipOverride on coneAngle < 0"
 [45]="corrected binormal = cross(normal, tangent)" [46]="texUV weights u and v swapped" [48]="lerp weights both the geometric normal's")
for k in 41 42 43 44 45 46 48; do
  g++ -O1 -g -std=c++17 -ffp-contract=off -DVT_MUTANT=$k -Ivistrace_amd/csrc/host -Iinclude -Ioracle -o /tmp/ttr_$k tests/cpp/test_trace_result.cpp \
      vistrace_amd/csrc/host/TraceResult.cpp tests/cpp/_build/vt_oracle_for_tests.o -fopenmp -lm 2> /dev/null || { echo "class mutant $k: build failed" >> "$OUT"; continue; }
  if /tmp/ttr_$k > /tmp/ttr_$k.log 2>&1; then v=SURVIVED; else v=KILLED; fi
  printf "class mutant %-2s %-48s %-8s %s\n" "$k" "${TWHAT[$k]}" "$v" "$(tail -1 /tmp/ttr_$k.log)" >> "$OUT"
done
# the shard rule of the multi-GPU path (host_api.cpp: vt_shard_capacity / vt_shard_bounds; gather_schedule.h: gather_chunk_bounds)
echo "# shard rule: tests/test_multigpu_gloo.py (properties + world_size-2 gloo runs), tests/cpp/test_gather_schedule (83: the header compiled -DVT_MUTANT=83)" >> "$OUT"
declare -A SWHAT=([81]="shard capacity not rounded to whole 64-ray blocks" [82]="the tail shard's end not clamped to n" [83]="a chunk's end not clamped to the shard" [84]="rays per shard rounded DOWN")
for k in 81 82 84; do
  make -C vistrace_amd/csrc host_mutant K=$k > /tmp/hm_build_$k.log 2>&1 || { echo "mutant $k: build failed" >> "$OUT"; continue; }
  L=$PWD/vistrace_amd/lib/variants/libvistrace_hip_hostmut_$k.so
  verdict=SURVIVED; by=""
  if ! VISTRACE_HIP_LIB=$L timeout 900 python -m pytest tests/test_multigpu_gloo.py -m "not gpu" -x -q -p no:cacheprovider > /tmp/hm_$k.log 2>&1; then
    by=$(grep -m1 -E "^(FAILED|ERROR) " /tmp/hm_$k.log | sed -E 's/ - .*//'); verdict=KILLED
  fi
  printf "shard mutant %-2s %-52s %-8s %s\n" "$k" "${SWHAT[$k]}" "$verdict" "$by" >> "$OUT"
  rm -rf vistrace_amd/csrc/_build_hostmut_$k "$L"
done
if g++ -O2 -g -std=c++17 -DVT_MUTANT=83 -Ivistrace_amd/csrc -o /tmp/tgs_83 tests/cpp/test_gather_schedule.cpp 2> /dev/null; then
  if timeout 600 /tmp/tgs_83 > /tmp/tgs_83.log 2>&1; then v=SURVIVED; else v=KILLED; fi
  printf "shard mutant %-2s %-52s %-8s %s\n" 83 "${SWHAT[83]}" "$v" "$(tail -1 /tmp/tgs_83.log | cut -c1-120)" >> "$OUT"
fi
cat "$OUT"
}
main "$@"; exit
