"""Build-contract checks on the compiled gfx950 code of the traversal kernels (hipcc cross-compiles without a GPU).

* Register budgets: the headline kernel must stay within 72 VGPRs and the ALPHA variants within 80 (six resident blocks
  per CU is what the LDS budget admits; one register more than 80 would make it five), nothing may spill.
* The ALPHA variants issue their texel loads asynchronously into v76..v79 and read them one loop iteration later
  (trace_kernels.hip, "ALPHA 1" / "ALPHA 2").  That is only sound while the compiler itself never touches those four
  registers: the kernel is told to allocate at most 76 VGPRs (amdgpu_num_vgpr), which the compiler may exceed for a variant that needs more, and this test reads the generated assembly to make
  sure that every mention of v76..v79 sits inside one of the hand-written asm statements."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernel_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = tmp_path_factory.mktemp("isa") / "trace_kernels.s"
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-fno-gpu-flush-denormals-to-zero", "-fno-slp-vectorize", f"-I{ROOT}/include", f"-I{ROOT}/vistrace_amd/csrc",
           "--cuda-device-only", "-S", "-o", str(out), f"{ROOT}/vistrace_amd/csrc/trace_kernels.hip",
           "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    usage = {}
    name = None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {}
        m = re.search(r"remark:\s+(VGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and name:
            usage[name][m.group(1)] = int(m.group(2))
    return out.read_text(), usage


def variants(usage):
    """(mangled name, ANY_HIT, STATS, PERSISTENT, FETCH_DMA, ALPHA) of every trace_kernel / trace_kernel_alpha instantiation."""
    for name in usage:
        m = re.match(r"_ZN2vt12trace_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)ELb0EEEvNS_9TraceArgsE$", name)
        if m:
            yield (name, *[int(x) for x in m.groups()], 0)
        m = re.match(r"_ZN2vt18trace_kernel_alphaILb(\d)ELb(\d)ELb(\d)ELb(\d)EEEvNS_9TraceArgsE$", name)
        if m:
            yield (name, *[int(x) for x in m.groups()], 1)


def test_register_budgets(kernel_asm):
    _, usage = kernel_asm
    seen = 0
    for name, any_hit, stats, persistent, dma, alpha in variants(usage):
        u = usage[name]
        seen += 1
        assert u["VGPRs Spill"] == 0 and u["SGPRs Spill"] == 0 and u["ScratchSize [bytes/lane]"] == 0, (name, u)
        assert u["VGPRs"] <= (80 if alpha else 72), (name, u)
    assert seen >= 20
    headline = usage["_ZN2vt12trace_kernelILb0ELb0ELb1ELb1ELb0EEEvNS_9TraceArgsE"]
    assert headline["VGPRs"] <= 72


def test_texel_registers_belong_to_the_hand_written_asm(kernel_asm):
    text, usage = kernel_asm
    checked = 0
    for name, any_hit, stats, persistent, dma, alpha in variants(usage):
        body = text[text.index(f"\n{name}:"):]
        body = body[: body.index("\n.Lfunc_end")]      # the whole function: a kernel may hold more than one s_endpgm
        in_asm = False
        loads = reads = 0
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0]
            if re.search(r"\bv7[6-9]\b", code) or any(int(a) <= 79 and int(b) >= 76 for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code)):
                if alpha and stats and not in_asm:
                    continue                                            # no loads in flight there: the registers are ordinary
                assert alpha and in_asm, f"{name}: the compiler touches a reserved texel register: {line.strip()}"
                loads += "global_load_ubyte" in code
                reads += "v_mov_b32" in code
        if alpha and not stats:                                         # the STATS variants read their texels in place
            assert loads == 5 and reads == 4, (name, loads, reads)     # bilinear 4 + nearest 1; one read-out block
            checked += 1
        if alpha and stats:
            assert loads == 0 and reads == 0, (name, loads, reads)
    assert checked >= 6
