#!/bin/bash
# Mutation testing of the GPU parity suite (run on the GPU box; writes profiles/r6/mutants.txt via gpurun_out/):
#   bash scripts/mutants.sh [out.txt] [mutant numbers...]
# Every library vistrace_amd/lib/variants/libvistrace_hip_mut_<k>.so (built here on the CPU box by
# `make -C vistrace_amd/csrc mutant K=<k>`, i.e. trace_kernels.hip with ONE deliberate deviation, see its VT_MUT list)
# is put in front of the `-m gpu` tests through VISTRACE_HIP_LIB.  A mutant is KILLED by the first test that goes red
# (parity files first, then everything else); a mutant that passes every test SURVIVED and names a missing test.
# The unmutated product library runs first as the control and must be green.
# (the body is one function, parsed as a whole before it runs: a child process that inherits and moves the script's file
# offset cannot make bash re-read lines -- round 6's first run listed three mutants twice)
main() {
OUT=${1:-gpurun_out/mutants.txt}; shift
KS=${@:-1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 21 22 23 24 25 26 27 28 31 32 43 44 45 51 52 53 54 55 56 57 58 61 62 63 64 65 66 71 73}        # 16 only on request: it hangs the kernel (killed by the timeout)
declare -A WHAT=(
 [1]="near/far swap on fl >= fr instead of >"
 [2]="hit accepted on t < tmax instead of <="
 [3]="fp contraction on (fused multiply-add)"
 [4]="pending leaf range drained from the back"
 [5]="plain 1/x instead of safe_inverse"
 [6]="slab entry without the tmin term"
 [7]="node accepted on first < second instead of <="
 [8]="hit needs u > 0 instead of >= 0"
 [9]="back face culled on n.d >= 0 instead of > 0"
 [10]="stack entries beyond the LDS part hold the near child"
 [11]="hit accepted on t > tmin instead of >="
 [12]="w = 1 - (u + v) instead of (1 - u) - v"
 [13]="FRONT faces culled (n.d < 0) instead of back faces"
 [14]="hit needs v > 0 instead of >= 0"
 [15]="hit needs w > 0 instead of >= 0"
 [16]="stack entry lds_entries still written to LDS"
 [21]="GetPos with w = 1 - (u + v)"
 [22]="frontFacing on dot(wo, n) > 0 instead of >= 0"
 [23]="texUV weights u and v swapped"
 [24]="refit: e1 = p1 - p0 instead of p0 - p1"
 [25]="refit: a leaf's box misses vertex 2"
 [26]="skinning: weight of bone 0 for every bone"
 [27]="CalcRayOrigin: |pos| <= 1/32 instead of <"
 [28]="bounce direction: sin and cos of phi swapped"
 [31]="device lineariser: left subtree counted for left children too"
 [32]="level lists: a level starts one pair late"
 [43]="CalcTBN correction on cosTheta < 0.1 instead of <="
 [44]="footprint off on coneAngle < 0 instead of <= 0"
 [45]="corrected binormal = cross(normal, tangent)"
 [51]="queue count: one entry behind a short queue is counted"
 [52]="queue scan: the carry between 16-B groups drops a count"
 [53]="queue scan of the partials: thread d skips its add"
 [54]="queue emit: a hit's slot counts the hit itself"
 [55]="miss fill skipped when ONE path has died"
 [56]="queue emit: rows of paths that missed are not written"
 [57]="range check: tMax < tMin instead of <="
 [61]="alpha test passes on alpha > ref instead of >="
 [62]="alpha texel: negative index mirrored, not repeated"
 [63]="alpha bilinear without the half-texel shift in x"
 [64]="alpha bilinear: right neighbour clamped, not wrapped"
 [65]="untextured alpha material passes whatever its ref"
 [66]="alpha nearest: a / 256 instead of a / 255"
 [71]="merged launch: first block of a set looked up in the set before it"
 [73]="a NaN range costs no step (per-ray counters)"
 [58]="range check: index within the chunk, not within the batch"
)
# mutant 9 is EQUIVALENT (trace_kernels.hip's VT_MUT list says why): it must survive; every other one must be killed
FIRST="tests/test_gpu_parity.py tests/test_gpu_shading_frame.py tests/test_gpu_rebuild.py tests/test_gpu_configs.py"
REST="tests/test_gpu_multi_batch.py tests/test_gpu_fake_group.py tests/test_gpu_bench_ranks.py"
mkdir -p "$(dirname "$OUT")"
{
echo "# mutation testing of the -m gpu parity suite, $(date -u +%Y-%m-%dT%H:%MZ), $(python3 -c 'import subprocess;print(subprocess.run(["git","rev-parse","--short","HEAD"],capture_output=True,text=True).stdout.strip() or "snapshot")')"
echo "# control: the product library"
} > "$OUT"
t0=$(date +%s)
if timeout 900 python -m pytest $FIRST -m gpu -x -q -p no:cacheprovider > /tmp/mut_ctl.log 2>&1; then
  echo "control   product library                                          GREEN  ($(grep -E ' passed' /tmp/mut_ctl.log | tail -1)) $(( $(date +%s) - t0 )) s" >> "$OUT"
else
  echo "control   product library                                          RED -- the run below means nothing" >> "$OUT"; tail -20 /tmp/mut_ctl.log >> "$OUT"
fi
for k in $KS; do
  L=$PWD/vistrace_amd/lib/variants/libvistrace_hip_mut_$k.so
  [ -f "$L" ] || { echo "mutant $k: $L not built" >> "$OUT"; continue; }
  t0=$(date +%s)
  verdict=SURVIVED; by=""
  for files in "$FIRST" "$REST"; do
    VISTRACE_HIP_LIB=$L timeout 600 python -m pytest $files -m gpu -x -q -p no:cacheprovider > /tmp/mut_$k.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then
      by=$(grep -m1 -E "^(FAILED|ERROR) " /tmp/mut_$k.log | sed -E 's/ - .*//')
      [ -z "$by" ] && by="rc=$rc: $(tail -1 /tmp/mut_$k.log | cut -c1-120)"
      verdict=KILLED; break
    fi
  done
  [ "$k" = 9 ] && [ "$verdict" = SURVIVED ] && verdict="SURVIVED (equivalent mutant: expected)"
  printf "mutant %-2s %-58s %-8s %s (%s s)\n" "$k" "${WHAT[$k]}" "$verdict" "$by" "$(( $(date +%s) - t0 ))" >> "$OUT"
done
cat "$OUT"
}
main "$@"; exit
