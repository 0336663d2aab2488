#!/bin/bash
# Mutation testing of the reference-shaped host classes' argument checks (vistrace_amd/csrc/host/AccelStruct.cpp, following
# source/objects/AccelStruct.cpp:802-806): tests/cpp/test_binding (a fake Lua state drives the thunks) rebuilt with AccelStruct.cpp
# compiled -DVT_MUTANT=<k>.     bash scripts/mutants_binding.sh build      (CPU box: tests/cpp/_build/test_binding_mut_<k>)
#                               bash scripts/mutants_binding.sh [out.txt]  (GPU box: runs them; a mutant is KILLED when a check fails)
main() {
cd "$(dirname "$0")/.."
KS="91 92 94 95 96"
declare -A WHAT=([91]="Traverse: valid cone width with coneAngle == 0 accepted" [92]="Traverse: coneWidth == 0 with a cone angle refused"
 [94]="Traverse: tMax == tMin accepted" [95]="TraverseBatch (ray tables): tMax == tMin accepted" [96]="Traverse: coneWidth == 0 with coneAngle <= 0 accepted")
H=vistrace_amd/csrc/host; B=tests/cpp/_build
if [ "$1" = build ]; then
  make -C tests/cpp > /dev/null || exit 1
  for k in $KS; do
    g++ -O2 -g -std=c++17 -Wall -Wextra -ffp-contract=off -DVT_MUTANT=$k -Iinclude -I$H -Itests/cpp -o $B/test_binding_mut_$k tests/cpp/test_binding.cpp \
        $H/AccelStruct.cpp $H/TraceResult.cpp $H/TraceResultBatch.cpp $H/Binding.cpp -Lvistrace_amd/lib -lvistrace_hip \
        -Wl,-rpath,'$ORIGIN/../../../vistrace_amd/lib' || exit 1
  done
  ls $B | grep -c test_binding_mut_; exit 0
fi
OUT=${1:-gpurun_out/mutants_binding.txt}; mkdir -p "$(dirname "$OUT")"
{ echo "# mutation testing of the binding's argument checks (host/AccelStruct.cpp), $(date -u +%Y-%m-%dT%H:%MZ): tests/cpp/test_binding with the class compiled -DVT_MUTANT=<k>"
  echo "control   product class                                              $(timeout 300 $B/test_binding 2>&1 | tail -1)"; } > "$OUT"
for k in $KS; do
  if timeout 300 $B/test_binding_mut_$k > /tmp/tb_$k.log 2>&1; then v=SURVIVED; else v=KILLED; fi
  printf "binding mutant %-2s %-62s %-8s %s\n" "$k" "${WHAT[$k]}" "$v" "$(grep -m1 -iE 'FAIL' /tmp/tb_$k.log | cut -c1-140)" >> "$OUT"
done
cat "$OUT"
}
main "$@"; exit
