"""Build-contract checks on the compiled gfx950 code of the traversal kernels (hipcc cross-compiles without a GPU).

The check itself lives beside the kernels (`vistrace_amd/csrc/check_isa.py`) and is part of the build: the Makefile
disassembles the object it has just produced and deletes it when the contract does not hold, so a library cannot be
linked from a kernel that breaks it.  This test runs the same check on the SHIPPED object (the one `libvistrace_hip.so`
was linked from, not a recompilation with flags of its own) and makes sure the checker still notices the two things it
exists for:

* register budgets: every plain variant within 72 VGPRs, the ALPHA variants within 80 (six resident blocks per CU is
  what the LDS budget admits), nothing spills;
* the ALPHA variants issue their texel loads asynchronously into v76..v79 and read them one loop iteration later
  (trace_kernels.hip, "ALPHA 1" / "ALPHA 2"): sound only while every mention of those registers is one of the nine
  hand-written instructions."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vistrace_amd", "csrc")
OBJ = os.path.join(CSRC, "_build", "trace_kernels.o")


def load_checker():
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(CSRC, "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def checker():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    res = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True, timeout=1200)   # a no-op after build()
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return load_checker()


def test_shipped_object_holds_the_contract(checker):
    errors, seen, alpha_checked = checker.check(OBJ)
    assert errors == []
    assert seen >= 20 and alpha_checked >= 6


def test_library_is_not_older_than_the_checked_object():
    lib = os.path.join(ROOT, "vistrace_amd", "lib", "libvistrace_hip.so")
    if not (os.path.exists(lib) and os.path.exists(OBJ)):
        pytest.skip("library not built here")
    assert os.path.getmtime(lib) >= os.path.getmtime(OBJ)


def test_checker_recognises_foreign_uses_of_the_texel_registers(checker):
    # the instruction classifier on hand-made disassembly lines
    assert checker.touches_texel_regs("v_fma_f32 v76, -v71, v75, v74")
    assert checker.touches_texel_regs("global_load_dwordx4 v[74:77], v1, s[2:3]")
    assert checker.touches_texel_regs("ds_read_b128 v[76:79], v3")
    assert not checker.touches_texel_regs("v_fma_f32 v75, -v71, v70, v74")
    assert not checker.touches_texel_regs("global_load_dwordx4 v[72:75], v1, s[2:3]")
    assert not checker.touches_texel_regs("s_mov_b32 s76, s77")


def test_no_auxiliary_kernel_spills_or_uses_scratch(checker):
    """Every other kernel of the library (TraceResult materialisation, shading frame, refit, skinning, ray generation, the bounce
    loop's queue, the batch range check): no VGPR / SGPR spills, no private segment, and room for full occupancy (<= 64 VGPRs)."""
    import tempfile
    seen = 0
    for name in ("trace_kernels.o", "shading.o", "batch.o"):
        obj = os.path.join(CSRC, "_build", name)
        with tempfile.TemporaryDirectory() as tmp:
            meta = checker.kernel_metadata(checker.device_code_object(obj, tmp))
        for kernel, m in meta.items():
            if "trace_kernel" in kernel:
                continue                                  # the traversal kernels have their own contract (above)
            seen += 1
            assert int(m.get("vgpr_spill_count", 0)) == 0 and int(m.get("sgpr_spill_count", 0)) == 0, kernel
            assert int(m.get("private_segment_fixed_size", 0)) == 0, kernel
            assert int(m["vgpr_count"]) <= 64, (kernel, m["vgpr_count"])
    assert seen >= 15
