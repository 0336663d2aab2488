#!/usr/bin/env python3
"""Build-contract check on the gfx950 code of the traversal kernels, run by the Makefile on the object it just built.

    python3 check_isa.py _build/trace_kernels.o        exit code 0 = the object may ship

The ALPHA kernel variants (trace_kernels.hip, "ALPHA 1" / "ALPHA 2") issue their texel loads asynchronously into
v76..v79 and read them one loop iteration later.  That is only sound while the compiler itself never touches those four
registers: the kernels ask for at most 76 compiler-allocated VGPRs (amdgpu_num_vgpr), which is a request.  This script
reads the code the compiler actually produced -- the device code object inside the host object, disassembled -- and fails
the build unless, in every trace_kernel_alpha variant that keeps loads in flight,

  * the only instructions that mention v76..v79 are the five hand-written texel loads (`global_load_ubyte v76..v79`)
    and the four read-outs (`v_mov_b32 vN, v76..v79`),
  * the kernel needs at most 80 VGPRs (six blocks per CU is what the LDS budget admits) and nothing spills,

and unless every other trace kernel stays within 72 VGPRs without spills.  A toolchain or flag change that breaks the
reservation therefore cannot produce a library; tests/test_kernel_asm.py runs the same check on the shipped object.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("VT_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ALPHA_VGPR_CAP = 80      # 76 compiler-allocated + v76..v79
PLAIN_VGPR_CAP = 72

PLAIN_RE = re.compile(r"^_ZN2vt12trace_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)ELb0EEEvNS_9TraceArgsE$")
ALPHA_RE = re.compile(r"^_ZN2vt18trace_kernel_alphaILb(\d)ELb(\d)ELb(\d)ELb(\d)EEEvNS_9TraceArgsE$")
DEVN_RE = re.compile(r"^_ZN2vt17trace_kernel_devnILb(\d)ELb(\d)EEEvNS_9TraceArgsE$")    # ray count in device memory (bounce loop)


def device_code_object(obj, tmp):
    """the gfx950 code object inside a hipcc host object (llvm-objdump --offloading writes it beside its input)"""
    local = os.path.join(tmp, "in.o")
    shutil.copy(obj, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "in.o"], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
    found = [f for f in os.listdir(tmp) if "hipv4-amdgcn-amd-amdhsa--gfx950" in f]
    if len(found) != 1:
        raise SystemExit(f"check_isa: no gfx950 code object in {obj}")
    return os.path.join(tmp, found[0])


def kernel_metadata(co):
    """{kernel name: {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size}} from the code object's notes"""
    text = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in text.splitlines():
        if re.match(r"\s*- \.\w+:", line):          # a new list entry: kernel or argument
            cur = {} if line.strip().startswith("- .agpr_count") else cur
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s*(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
            if m.group(1) == "vgpr_spill_count" or ("name" in cur and "vgpr_count" in cur):
                kernels[cur["name"]] = cur
    return kernels


def functions(co):
    """{symbol: [instruction text, ...]} from the disassembly"""
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    out, name = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if name and line.strip():
            out[name].append(line.split("//")[0].strip())
    return out


def touches_texel_regs(ins):
    if re.search(r"\bv7[6-9]\b", ins):
        return True
    return any(int(a) <= 79 and int(b) >= 76 for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", ins))


def check(obj):
    errors = []
    with tempfile.TemporaryDirectory() as tmp:
        co = device_code_object(obj, tmp)
        meta = kernel_metadata(co)
        code = functions(co)
    seen = alpha_checked = 0
    for name, m in meta.items():
        plain, alpha = PLAIN_RE.match(name), ALPHA_RE.match(name)
        if DEVN_RE.match(name):              # same budget as the plain variants; not counted among the 20
            if int(m["vgpr_count"]) > PLAIN_VGPR_CAP or int(m.get("vgpr_spill_count", 0)) or int(m.get("sgpr_spill_count", 0)) or int(m.get("private_segment_fixed_size", 0)):
                errors.append(f"{name}: over budget or spilling ({m})")
            continue
        if not plain and not alpha:
            continue
        seen += 1
        stats = int((plain or alpha).group(2))
        vgprs = int(m["vgpr_count"])
        if int(m.get("vgpr_spill_count", 0)) or int(m.get("sgpr_spill_count", 0)) or int(m.get("private_segment_fixed_size", 0)):
            errors.append(f"{name}: spills or scratch ({m})")
        cap = ALPHA_VGPR_CAP if alpha else PLAIN_VGPR_CAP
        if vgprs > cap:
            errors.append(f"{name}: {vgprs} VGPRs, budget {cap}")
        if not alpha:
            continue
        body = code.get(name)
        if body is None:
            errors.append(f"{name}: not found in the disassembly")
            continue
        loads = reads = 0
        for ins in body:
            if not touches_texel_regs(ins):
                continue
            if stats:
                continue                     # the STATS variants read their texels in place: the registers are ordinary there
            if re.match(r"global_load_ubyte v7[6-9], ", ins):
                loads += 1
            elif re.match(r"v_mov_b32(_e32)? v\d+, v7[6-9]$", ins):
                reads += 1
            else:
                errors.append(f"{name}: the compiler touches a reserved texel register: {ins}")
        if not stats:
            alpha_checked += 1
            if (loads, reads) != (5, 4):    # bilinear 4 + nearest 1; one read-out block
                errors.append(f"{name}: expected 5 texel loads and 4 read-outs, found {loads} and {reads}")
    if seen < 20:
        errors.append(f"only {seen} trace kernel variants found (expected >= 20)")
    if alpha_checked < 6:
        errors.append(f"only {alpha_checked} ALPHA variants with loads in flight found (expected >= 6)")
    return errors, seen, alpha_checked


def is_mutant(obj):
    """a trace_kernels object compiled with -DVT_MUTANT=<k> (mutation testing, scripts/mutants.sh) defines vt_mutant"""
    syms = subprocess.run([f"{LLVM}/llvm-readelf", "--symbols", "--wide", obj], check=True, capture_output=True, text=True).stdout
    return any(f[-1] == "vt_mutant" and f[-2] != "UND" for f in (line.split() for line in syms.splitlines()) if len(f) >= 8)


def main():
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    errors, seen, alpha_checked = check(sys.argv[1])
    # a deliberately wrong kernel may only ever be written to a variant directory, never to the product's _build/
    if is_mutant(sys.argv[1]) and os.path.basename(os.path.dirname(os.path.abspath(sys.argv[1]))) == "_build":
        errors.append("this object was compiled with -DVT_MUTANT: a mutant must not be built into the product library")
    for e in errors:
        print("check_isa: " + e, file=sys.stderr)
    if errors:
        raise SystemExit(1)
    print(f"check_isa: {seen} trace kernel variants within budget, texel registers private in {alpha_checked} ALPHA variants")


if __name__ == "__main__":
    main()
