#!/usr/bin/env python3
"""Static instruction mix of one trace-kernel instantiation, per body of its loop (round 6's bookkeeping audit).

    python3 scripts/isa_audit.py [ANY_HIT STATS PERSISTENT FETCH_DMA ALPHA]  [-D...]      (default 0 0 1 1 0 = the headline kernel)

Compiles trace_kernels.hip as the Makefile does plus -gline-tables-only (line tables do not change the generated code) and
attributes every instruction of the kernel to the body whose SOURCE LINES it came from (.loc directives): the VT_ISA_MARK anchors
in trace_body, the start_ray / finish_ray / enter_batch lambdas and the small helpers above them.  Bodies: prologue (set-up, CU
reservation), refill (idle-lane ballot, ray-block cursor, coherence probe), start_ray (ray load, safe_inverse, scaled origin),
select (which record does a lane need), fetch (DPP broadcast, 4 LDS-DMA loads, 4 ds_read_b128), tri (Primitives.h:168-215), node
(slab tests, near/far, stack), finish (store of a hit record), exit."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "vistrace_amd", "csrc", "trace_kernels.hip")
targs = [a for a in sys.argv[1:] if not a.startswith("-")]
defs = [a for a in sys.argv[1:] if a.startswith("-")]
A, S, P, D, AL = (targs + ["0", "0", "1", "1", "0"][len(targs):])[:5]
sym = (f"_ZN2vt18trace_kernel_alphaILb{A}ELb{S}ELb{P}ELb{D}EEEvNS_9TraceArgsE" if AL == "1"
       else f"_ZN2vt12trace_kernelILb{A}ELb{S}ELb{P}ELb{D}ELb0EEEvNS_9TraceArgsE")

# ---- source line -> body -------------------------------------------------------------------------------------------------
lines = open(SRC).read().splitlines()
def line_of(pattern, start=0):
    for i in range(start, len(lines)):
        if pattern in lines[i]:
            return i + 1
    raise SystemExit(f"isa_audit: anchor not found: {pattern}")
body_start = line_of("__device__ __forceinline__ void trace_body(")
anchors = [(body_start, "prologue"),
           (line_of("auto enter_batch = [&]"), "refill"),
           (line_of("auto start_ray = [&]"), "start_ray"),
           (line_of("auto finish_ray = [&]"), "finish"),
           (line_of("if constexpr (!PERSISTENT) {", line_of("auto finish_ray = [&]")), "prologue")]
for name in ("refill", "select", "fetch", "tri", "node", "finish", "exit"):
    anchors.append((line_of(f'VT_ISA_MARK("{name}")'), name))
anchors.append((line_of("// The two kernels around the body."), None))
anchors.sort()
helpers = {"float safe_inverse(float x)": "start_ray", "float slab_first(float e0": "node", "float slab_second(float x0": "node",
           "float sel(uint64_t mask": "node", "uint32_t quad_broadcast(uint32_t v)": "fetch", "uint32_t lane_id()": "prologue",
           "uint32_t prefix_count(uint64_t mask)": "refill", "void leave_grid(const TraceArgs& a": "exit"}
helper_ranges = []
for pat, body in helpers.items():
    a = line_of("__device__ __forceinline__" if False else pat)
    b = a
    while b < len(lines) and lines[b - 1].strip() != "}":
        b += 1
    helper_ranges.append((a - 2, b, body))
def body_of(line):
    for a, b, body in helper_ranges:
        if a <= line <= b:
            return body
    cur = "other"
    for a, name in anchors:
        if line >= a:
            cur = name or "other"
    return cur if line >= body_start else "other"

# ---- compile with line tables ----------------------------------------------------------------------------------------------
tmp = tempfile.mkdtemp()
asm = os.path.join(tmp, "all.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                       "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-gline-tables-only", f"-I{ROOT}/include",
                       f"-I{ROOT}/vistrace_amd/csrc", *defs, "--cuda-device-only", "-S", "-o", asm, SRC], stderr=subprocess.DEVNULL)
CLASSES = [
    ("dpp", lambda i, t: "quad_perm" in t or "row_" in t or "_dpp" in i),
    ("cndmask", lambda i, t: i.startswith("v_cndmask")),
    ("cmp", lambda i, t: i.startswith("v_cmp")),
    ("mov", lambda i, t: i.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr"))),
    ("minmax", lambda i, t: i.startswith(("v_max", "v_min"))),
    ("f32 mul/add", lambda i, t: re.match(r"v_(mul|add|sub|subrev|fma|fmac|mac|pk_mul|pk_add|pk_fma)_f32", i) is not None),
    ("rcp/cvt", lambda i, t: i.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_div", "v_floor", "v_cvt", "v_frexp", "v_ldexp"))),
    ("int valu", lambda i, t: i.startswith("v_")),
    ("lds", lambda i, t: i.startswith("ds_")),
    ("vmem", lambda i, t: i.startswith(("global_", "buffer_", "flat_", "scratch_"))),
    ("branch", lambda i, t: i.startswith(("s_cbranch", "s_branch"))),
    ("wait/nop", lambda i, t: i.startswith(("s_waitcnt", "s_nop", "s_setprio", "s_barrier"))),
    ("salu", lambda i, t: i.startswith("s_")),
]
names = [c for c, _ in CLASSES]
ORDER = ["prologue", "refill", "start_ray", "select", "fetch", "tri", "node", "finish", "exit", "other"]
bodies = {b: {c: 0 for c in names} for b in ORDER}
inside, cur_line, main_files = False, 0, set()
for raw in open(asm):
    t = raw.strip()
    fm = re.match(r'\.file\s+(\d+)\s+.*trace_kernels\.hip"', t)
    if fm:
        main_files.add(int(fm.group(1)))
    if t.startswith(sym + ":"):
        inside = True
        continue
    if inside and t.startswith(".Lfunc_end"):
        break
    if not inside:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        if int(m.group(1)) in main_files:      # code inlined from the HIP headers (fabsf, __ballot ...) stays with its call site
            cur_line = int(m.group(2))
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    ins = t.split()[0]
    if not re.match(r"[vs]_|global_|buffer_|flat_|scratch_|ds_", ins):
        continue
    b = body_of(cur_line)
    for c, f in CLASSES:
        if f(ins, t):
            bodies[b][c] += 1
            break
print(f"trace_kernel<{A},{S},{P},{D},{AL}>{' ' + ' '.join(defs) if defs else ''}: static instructions by the body their source lines belong to")
print("| body | " + " | ".join(names) + " | VALU | all |")
print("|---|" + "---|" * (len(names) + 2))
tot = {c: 0 for c in names}
for b in ORDER:
    r = bodies[b]
    if not sum(r.values()):
        continue
    print(f"| {b} | " + " | ".join(str(r[c]) for c in names) + f" | {sum(r[c] for c in names[:8])} | {sum(r.values())} |")
    for c in names:
        tot[c] += r[c]
print("| **sum** | " + " | ".join(str(tot[c]) for c in names) + f" | {sum(tot[c] for c in names[:8])} | {sum(tot.values())} |")
