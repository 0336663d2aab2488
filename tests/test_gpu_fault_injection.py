"""Failure injection on the HIP error paths (SURVEY.md section 5 "Failure detection"; the reference's convention is
delete-before-throw: source/VisTrace.cpp:782-785, source/objects/AccelStruct.cpp:186-203, :780).

Behind VT_ENABLE_TEST_HOOKS=1 every device / pinned-host allocation of the library goes through one counted wrapper;
vt_test_fail_alloc(k) makes the k-th one fail.  tests/fault_injection_check.py sweeps k over engine open, Rebuild's upload (single
engine, 3-member group, host-linearised), the alpha tables, skinning, refit, host-pointer traces, batch objects and sets, launch
scratch, the bounce loop and the group's gather: a status + message every time (or a designed retry with correct results), the
engine still reproduces the golden fixture, device memory back at its level.  tests/cpp/test_binding --fail-alloc does the same
through the Lua surface: a Lua error, never an abort."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "cpp", "_build", "libfake_rccl.so")


def _env():
    return dict(os.environ, VT_ENABLE_TEST_HOOKS="1", VT_TEST_ALLOW_DEVICE_ALIASES="1", VT_RCCL_LIB=FAKE)


@pytest.mark.gpu
def test_every_allocation_failure_is_reported_and_leaves_the_engine_usable():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "fake_rccl"], stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_injection_check.py")], capture_output=True, text=True,
                         timeout=1500, env=_env())
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "fault injection: ok" in out.stdout
    lines = {m.group(1): (int(m.group(2)), int(m.group(3)), int(m.group(4)))
             for m in re.finditer(r"^(.+?): (\d+) allocations, (\d+) injected failures reported as errors, (\d+) absorbed", out.stdout, re.M)}
    assert len(lines) >= 16, out.stdout
    # scene-building and scene-rewriting calls have no retry on their path: every injected failure is an error
    for name, (count, failed, absorbed) in lines.items():
        if name.startswith(("vt_engine_open", "vt_scene_")):
            assert count >= 1 and failed == count and absorbed == 0, (name, count, failed, absorbed)
    assert lines["vt_engine_open"][0] >= 3 and lines["vt_scene_upload_tree (3 members)"][0] > lines["vt_scene_upload_tree"][0]


@pytest.mark.gpu
def test_allocation_failures_surface_as_lua_errors():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_binding")
    out = subprocess.run([exe, "--fail-alloc"], capture_output=True, text=True, timeout=600, env=dict(os.environ, VT_ENABLE_TEST_HOOKS="1"))
    assert out.returncode == 0 and " 0 failed" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "surfaced as Lua errors" in out.stdout


@pytest.mark.gpu
def test_every_checked_hip_call_failure_is_reported_and_leaves_the_engine_usable():
    """The second hook (vt_test_fail_hip): every OTHER HIP call the library checks -- copies, event and stream calls, launch checks,
    synchronisations -- reports a failure in turn instead of being made, over the same operations.  A status + message (or a second
    way with correct results: the group's replication falls back to per-member uploads), nothing leaked, the engine still
    reproduces the golden fixture; an object that was being updated in place is healed by repeating the call."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "fake_rccl"], stdout=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_injection_check.py"), "hip"], capture_output=True, text=True,
                         timeout=2400, env=_env())
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "fault injection: ok" in out.stdout
    lines = {m.group(1): (int(m.group(2)), int(m.group(3)), int(m.group(4)))
             for m in re.finditer(r"^(.+?): (\d+) checked HIP calls, (\d+) injected failures reported as errors, (\d+) absorbed", out.stdout, re.M)}
    assert len(lines) >= 16, out.stdout
    assert sum(c for c, _, _ in lines.values()) >= 300                  # the sweep is over hundreds of early returns
    for name, (count, failed, absorbed) in lines.items():
        # no second way on these paths: every failure is an error.  (The refits capture their level-by-level launches into a
        # hipGraph on first use; a capture that fails falls back to plain launches -- those failures are absorbed by design.)
        # (and a group's device-to-device replication falls back to per-member uploads)
        if "refit" not in name and "members" not in name and name.startswith(("vt_engine_open", "vt_scene_", "vt_trace_closest (", "vt_batch_")):
            assert count >= 1 and failed == count and absorbed == 0, (name, count, failed, absorbed)
        if "refit" in name:
            assert failed >= 10 and absorbed >= 1, (name, count, failed, absorbed)


@pytest.mark.gpu
def test_checked_hip_call_failures_surface_as_lua_errors():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_binding")
    out = subprocess.run([exe, "--fail-hip"], capture_output=True, text=True, timeout=600, env=dict(os.environ, VT_ENABLE_TEST_HOOKS="1"))
    assert out.returncode == 0 and " 0 failed" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "surfaced as Lua errors" in out.stdout


def test_the_hook_is_dead_without_the_switch(va):
    """vt_test_fail_alloc refuses to arm unless VT_ENABLE_TEST_HOOKS=1 (this process: unset)."""
    if os.environ.get("VT_ENABLE_TEST_HOOKS") == "1":
        pytest.skip("hooks are on in this environment")
    L = va._lib
    assert L.lib.vt_test_fail_alloc(1) == L.VT_ERR_UNSUPPORTED and b"test hooks are off" in L.lib.vt_last_error()
    assert L.lib.vt_test_alloc_count() == 0


def test_no_allocation_bypasses_the_counted_wrappers():
    """Every hipMalloc / hipHostMalloc of the product goes through dev_malloc / pinned_malloc (engine_internal.h)."""
    csrc = os.path.join(ROOT, "vistrace_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".cpp")):
            continue
        text = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, fn)).read())
        assert not re.search(r"\bhip(Host)?Malloc\s*\(", text), f"{fn} allocates behind the fault-injection wrappers"
        assert not re.search(r"\bhipMalloc(Async|Managed|Pitch)\b", text), fn
