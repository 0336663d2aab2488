"""Runs tests/cpp/test_binding (fake Lua state driving the Tracing API thunks of
vistrace_amd/csrc/host).  CPU part: error messages, type checks, TraceResult math.
GPU part: CreateAccel / Rebuild / Traverse / TraverseBatch / getters end to end."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_binding")


def _build():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)


def test_binding_cpu_side():
    _build()
    out = subprocess.run([EXE, "--cpu"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


@pytest.mark.gpu
def test_binding_end_to_end():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


@pytest.mark.gpu
def test_binding_single_call_config1():
    """BASELINE config 1 through the binding: 10 k single-ray accel:Traverse calls on a 10 k-triangle world."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    out = subprocess.run([EXE, "--bench"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "us per call" in out.stdout and "0 failed" in out.stdout, out.stdout + out.stderr


def test_host_code_under_asan_ubsan():
    """Triangle set-up, PLOC build, leaf collapse and linearise under ASan + UBSan (CPU only)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "sanitize"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_host_sanitize")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="3")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failed" in out.stdout


def test_host_trace_result_class_equals_the_oracle():
    """host/TraceResult.cpp (what one accel:Traverse call hands to Lua) against the oracle on the CPU, under ASan + UBSan: ctor
    fields, GetPos, texUV / blend, Normal / Tangent / Binormal (CalcTBN without a normal map, grazing branch included) and the
    cone footprint on 20 000 random triangles, frames and hit points -- bit for bit."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "trace_result"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_build", "test_trace_result")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and " 0 failed" in out.stdout, out.stdout + out.stderr


def test_plain_c_example_builds_and_fails_loudly_without_gpu():
    """examples/trace_batch.c compiles as C11 against the header; without a device it reports
    VT_ERR_HIP (exit 2) instead of falling back to anything."""
    import torch
    _build()
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "trace_batch")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by test_plain_c_example_on_gpu")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "no HIP device" in out.stderr + out.stdout or "hipGetDeviceCount" in out.stderr


@pytest.mark.gpu
def test_plain_c_example_on_gpu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_build", "trace_batch")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_plain_c_batch_sets_on_gpu():
    """examples/trace_sets.c: three host ray sets through vt_batch_set_begin / _add / _trace (one merged launch), from plain C."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_build", "trace_sets")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_plain_c_shading_frame_on_gpu():
    """examples/shading_frame.c: vertex frames, vt_batch_tbn and frames that follow a bone, from plain C."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_build", "shading_frame")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
