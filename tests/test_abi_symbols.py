"""The C-ABI library loads without a GPU and exports every symbol include/vistrace_hip.h
declares; POD sizes match the header; device entry points fail loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vistrace_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(va):
    syms = declared_symbols()
    assert len(syms) >= 30
    lib = C.CDLL(va._lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert set(va._lib.SYMBOLS) == set(syms), "python binding table and header disagree"


def test_product_library_is_no_mutant():
    """Mutation testing (scripts/mutants.sh) builds deliberately wrong kernels from trace_kernels.hip's VT_MUT sites; those
    libraries export vt_mutant().  The product library must not: nothing of a mutant is compiled into it, and the build never
    passes -DVT_MUTANT (the Makefile's `mutant` / `variant` targets write to lib/variants/ only)."""
    lib = C.CDLL(os.path.join(ROOT, "vistrace_amd", "lib", "libvistrace_hip.so"))
    assert not hasattr(lib, "vt_mutant")
    mk = open(os.path.join(ROOT, "vistrace_amd", "csrc", "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith(("CXXFLAGS", "HIPFLAGS", " ", "\t")) and "VT_MUTANT" in ln]
    assert all("variants" in ln or "_build_mut_" in ln or "-DVT_MUTANT=$(K)" in ln for ln in flags), flags
    src = open(os.path.join(ROOT, "vistrace_amd", "csrc", "trace_kernels.hip")).read()
    assert "#define VT_MUT(k, wrong, right) (right)" in src         # without -DVT_MUTANT every site is its shipped token


def test_core_block_matches_integration_md():
    """The header's CORE block = the twelve calls INTEGRATION.md's patch uses, one to one and in the same order."""
    text = open(os.path.join(ROOT, "include", "vistrace_hip.h")).read()
    core = text[text.index("/* ==== CORE:"):text.index("/* ==== EXTENSIONS")]
    core = re.sub(r"/\*.*?\*/", "", core, flags=re.S)
    in_header = re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", core)
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    para = md[md.index("**The twelve core calls**"):md.index("## 1. Members")]
    in_md = re.findall(r"`(vt_[a-z0-9_]+)`", para)
    assert len(in_header) == 12 and in_header == in_md, (in_header, in_md)
    patch = md[md.index("## 1. Members"):md.index("## 4.")] if "## 4." in md else md
    for name in in_md:                                   # ... and each of them really appears in sections 1-3
        assert name in patch or name == "vt_engine_close", name


def test_pod_sizes(va):
    L = va._lib
    assert (L.RAY.itemsize, L.HIT.itemsize, L.BVH_NODE.itemsize, L.NODE_PAIR.itemsize, L.TRI64.itemsize,
            L.RAY_STATS.itemsize, L.HIT_ATTRS.itemsize) == (32, 16, 32, 64, 64, 8, 64)
    assert L.lib.vt_abi_version() == 5


def test_no_product_import_of_oracle():
    """The shipped package must not import, load or link anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vistrace_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "vt_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_device_calls_fail_loudly_without_gpu(va):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(va._lib.VisTraceError) as e:
        va.Engine(0)
    assert e.value.code == va._lib.VT_ERR_HIP


def test_multi_gpu_entry_points_validate_arguments(va):
    """vt_engine_open_multi rejects an empty or duplicated device list before touching a device; without a GPU it fails
    loudly like vt_engine_open; the per-rank communicator calls refuse an engine-less / uninitialised use."""
    import ctypes as C
    import torch
    L = va._lib
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert L.lib.vt_engine_open_multi(devs, 2, C.byref(h)) == L.VT_ERR_INVALID_ARG and b"twice" in L.lib.vt_last_error()
    assert L.lib.vt_engine_open_multi(devs, 0, C.byref(h)) == L.VT_ERR_INVALID_ARG
    assert L.lib.vt_engine_open_multi(None, 1, C.byref(h)) == L.VT_ERR_INVALID_ARG
    assert L.lib.vt_engine_device_count(None) == 0 and L.lib.vt_engine_device(None, 0) == -1
    assert L.lib.vt_gather_hits_dev(None, None, 1, None, 0, None) == L.VT_ERR_INVALID_ARG
    assert L.lib.vt_engine_comm_init_rank(None, 1, 0, None) == L.VT_ERR_INVALID_ARG
    assert L.lib.vt_gather_wait(None, 0, None) == L.VT_ERR_INVALID_ARG
    assert L.lib.vt_host_scene_sync(None, None) == L.VT_ERR_INVALID_ARG
    if not torch.cuda.is_available():
        one = (C.c_int * 1)(0)
        assert L.lib.vt_engine_open_multi(one, 1, C.byref(h)) == L.VT_ERR_HIP


def test_round4_entry_points_validate_arguments(va):
    """The entry points added in round 4 (merged launches, batch options and sets, pieces of a gather, host registration) refuse
    NULL handles and malformed arguments before they touch a device, with a message -- no GPU needed."""
    L = va._lib
    lib = L.lib
    h = C.c_void_p()
    bad = C.c_uint64(0)
    badb = C.c_uint32(0)
    desc = np.zeros(1, L.BATCH_DESC)
    assert lib.vt_trace_closest_multi_dev(None, L.ptr(desc), 1, None) == L.VT_ERR_INVALID_ARG and b"scene is NULL" in lib.vt_last_error()
    assert lib.vt_trace_any_multi_dev(None, L.ptr(desc), 1, None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_trace_closest_ex(None, None, 0, 0, 0, C.byref(bad), C.byref(h)) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_trace_closest_ex(None, None, 0, 0, 0, C.byref(bad), None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_set_begin(None, 0, C.byref(h)) == L.VT_ERR_INVALID_ARG and h.value is None
    assert lib.vt_batch_set_begin(None, 0, None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_set_add(None, None, 0, 0, C.byref(bad)) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_set_trace(None, None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_set_count(None) == 0
    lib.vt_batch_set_abort(None)                                            # a no-op
    outs = (C.c_void_p * 1)()
    assert lib.vt_batch_trace_closest_set(None, None, None, None, 0, 0, C.byref(badb), C.byref(bad), outs) == L.VT_ERR_INVALID_ARG
    assert lib.vt_gather_hits_part_dev(None, None, 1, 0, 1, None, 0, None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_host_register(None, 16) == L.VT_ERR_INVALID_ARG and lib.vt_host_unregister(None) == L.VT_ERR_INVALID_ARG
    buf = np.zeros(16, np.uint8)
    assert lib.vt_host_register(L.ptr(buf), 0) == L.VT_ERR_INVALID_ARG
    # vertex frames / shading frame of a hit
    fr = np.zeros(1, L.TRI_FRAME)
    assert lib.vt_scene_set_tri_frames(None, L.ptr(fr), 1) == L.VT_ERR_INVALID_ARG and b"scene is NULL" in lib.vt_last_error()
    assert lib.vt_scene_read_tri_frames(None, L.ptr(fr)) == L.VT_ERR_INVALID_ARG
    assert lib.vt_hit_tbn_dev(None, None, None, 1, -1.0, -1.0, None, None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_batch_tbn(None, C.byref(h)) == L.VT_ERR_INVALID_ARG
    # the pieces of a shard are pure arithmetic
    lo, hi = C.c_uint64(7), C.c_uint64(7)
    lib.vt_gather_chunk_bounds(1000, 4, 1, C.byref(lo), C.byref(hi))
    assert (lo.value, hi.value) == (256, 512)
    lib.vt_gather_chunk_bounds(1000, 4, 9, C.byref(lo), C.byref(hi))
    assert (lo.value, hi.value) == (0, 0)
    lib.vt_gather_chunk_bounds(1000, 4, 1, None, None)                      # NULL outputs are tolerated


def test_round5_entry_points_validate_arguments(va):
    """vt_scene_upload_tree / vt_scene_upload_stats / vt_host_scene_download / vt_engine_member refuse NULL handles with a message."""
    L = va._lib
    lib = L.lib
    h = C.c_void_p()
    assert lib.vt_scene_upload_tree(None, None, None, 0, C.byref(h)) == L.VT_ERR_INVALID_ARG and b"NULL argument" in lib.vt_last_error()
    st = np.zeros(1, L.UPLOAD_STATS)
    assert lib.vt_scene_upload_stats(None, L.ptr(st)) == L.VT_ERR_INVALID_ARG
    assert lib.vt_host_scene_download(None, C.byref(h)) == L.VT_ERR_INVALID_ARG and h.value is None
    assert lib.vt_host_scene_download(None, None) == L.VT_ERR_INVALID_ARG
    assert lib.vt_engine_member(None, 0) is None
    assert L.UPLOAD_STATS.itemsize == 32
