"""ctypes binding of include/vistrace_hip.h (the C ABI of libvistrace_hip.so).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C vistrace_amd/csrc``).
There is no Python or CPU fallback: if the shared object is missing, importing this
module raises, and every tracing call needs a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VISTRACE_HIP_LIB") or os.path.join(_HERE, "lib", "libvistrace_hip.so")   # override: A/B of builds

VT_OK = 0
VT_ERR_INVALID_ARG = 1
VT_ERR_HIP = 2
VT_ERR_UNSUPPORTED = 3
VT_ERR_NOMEM = 4
VT_ERR_STACK = 5
VT_MISS = 0xFFFFFFFF
VT_TRI_CULL_BACKFACE = 1
VT_TRI_ALPHATEST = 2

# POD layouts of include/vistrace_hip.h
RAY = np.dtype([("org", "<f4", 3), ("dir", "<f4", 3), ("tmin", "<f4"), ("tmax", "<f4")])
HIT = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
BVH_NODE = np.dtype([("bounds", "<f4", 6), ("prim_count", "<u4"), ("first", "<u4")])
NODE_PAIR = np.dtype([("child", BVH_NODE, 2)])
TRI64 = np.dtype([("p0", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("n", "<f4", 3),
                  ("prim", "<u4"), ("flags", "<u4"), ("pad", "<u4", 2)])
RAY_STATS = np.dtype([("steps", "<u4"), ("tests", "<u4")])
HIT_ATTRS = np.dtype([("pos", "<f4", 3), ("t", "<f4"), ("ngeo", "<f4", 3), ("prim", "<u4"),
                      ("uvw", "<f4", 3), ("front", "<u4"), ("wo", "<f4", 3), ("hit", "<u4")])
TRI_ATTRIBS = np.dtype([("uv", "<f4", (3, 2)), ("alpha", "<f4", 3), ("ent_id", "<u4"), ("material", "<u4"), ("pad", "<u4")])
TRI_FRAME = np.dtype([("normal", "<f4", (3, 3)), ("tangent", "<f4", (3, 3))])
HIT_TBN = np.dtype([("normal", "<f4", 3), ("tangent", "<f4", 3), ("binormal", "<f4", 3), ("lod_info", "<f4", 2), ("lod_set", "<u4")])
HIT_SHADE = np.dtype([("tex_uv", "<f4", 2), ("blend", "<f4"), ("ent_id", "<u4"), ("material", "<u4"), ("pad", "<u4", 3)])
CAMERA = np.dtype([("pos", "<f4", 3), ("forward", "<f4", 3), ("up", "<f4", 3), ("vfov_deg", "<f4"), ("width", "<u4"), ("height", "<u4")])
ALPHA_MATERIAL = np.dtype([("tex_mat", "<f4", (2, 4)), ("tex_scale", "<f4"), ("alpha_ref", "<f4"), ("width", "<u4"),
                           ("height", "<u4"), ("filter", "<u4"), ("pad", "<u4"), ("offset", "<u8")])
SKIN_VERTEX = np.dtype([("weight", "<f4", 3), ("bone", "i1", 3), ("num_bones", "u1")])
UPLOAD_STATS = np.dtype([("alloc_ms", "<f4"), ("copy_ms", "<f4"), ("device_ms", "<f4"), ("total_ms", "<f4"), ("bytes_h2d", "<u8"),
                         ("linearised_on_device", "<u4"), ("pad", "<u4")])
BATCH_DESC = np.dtype([("d_rays", "<u8"), ("d_out", "<u8"), ("n", "<u8"), ("ray_image_width", "<u4"), ("reserved", "<u4")])
assert BATCH_DESC.itemsize == 32
assert TRI_FRAME.itemsize == 72 and HIT_TBN.itemsize == 48
assert TRI_ATTRIBS.itemsize == 48 and HIT_SHADE.itemsize == 32 and SKIN_VERTEX.itemsize == 16 and ALPHA_MATERIAL.itemsize == 64
assert RAY.itemsize == 32 and HIT.itemsize == 16 and BVH_NODE.itemsize == 32
assert NODE_PAIR.itemsize == 64 and TRI64.itemsize == 64 and HIT_ATTRS.itemsize == 64

# every symbol include/vistrace_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
_u32 = C.c_uint32
_u64 = C.c_uint64
_pp = C.POINTER(C.c_void_p)
SYMBOLS = {
    "vt_last_error": (C.c_char_p, []),
    "vt_abi_version": (C.c_int, []),
    "vt_tris_setup": (C.c_int, [_vp, _vp, _u32, _vp]),
    "vt_bvh_build": (C.c_int, [_vp, _u32, C.c_int, _pp]),
    "vt_bvh_build_ex": (C.c_int, [_vp, _u32, C.c_int, C.c_int, _pp]),
    "vt_bvh_refit": (C.c_int, [_vp, _vp]),
    "vt_bvh_free": (None, [_vp]),
    "vt_bvh_node_count": (_u32, [_vp]),
    "vt_bvh_prim_count": (_u32, [_vp]),
    "vt_bvh_nodes": (_vp, [_vp]),
    "vt_bvh_prim_indices": (_vp, [_vp]),
    "vt_scene_linearise": (C.c_int, [_vp, _vp, _pp]),
    "vt_host_scene_free": (None, [_vp]),
    "vt_host_scene_pair_count": (_u32, [_vp]),
    "vt_host_scene_tri_count": (_u32, [_vp]),
    "vt_host_scene_max_depth": (_u32, [_vp]),
    "vt_host_scene_root_leaf_count": (_u32, [_vp]),
    "vt_host_scene_pairs": (_vp, [_vp]),
    "vt_host_scene_tris": (_vp, [_vp]),
    "vt_host_scene_trace_closest": (C.c_int, [_vp, _vp, _u64, _vp]),
    "vt_host_scene_trace_any": (C.c_int, [_vp, _vp, _u64, _vp]),
    "vt_host_scene_set_alpha": (C.c_int, [_vp, _vp, _u32, _vp, _u32, _vp, _u64]),
    "vt_host_scene_sync": (C.c_int, [_vp, _vp]),
    "vt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vt_engine_open": (C.c_int, [C.c_int, _pp]),
    "vt_engine_close": (None, [_vp]),
    "vt_engine_open_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, _pp]),
    "vt_engine_device_count": (C.c_int, [_vp]),
    "vt_engine_device": (C.c_int, [_vp, C.c_int]),
    "vt_engine_member": (_vp, [_vp, C.c_int]),
    "vt_shard_capacity": (_u64, [_u64, C.c_int]),
    "vt_shard_bounds": (None, [_u64, C.c_int, C.c_int, C.POINTER(_u64), C.POINTER(_u64)]),
    "vt_trace_closest_gather_dev": (C.c_int, [_vp, C.POINTER(C.c_void_p), _u64, _vp]),
    "vt_comm_unique_id": (C.c_int, [_vp]),
    "vt_engine_comm_init_rank": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "vt_gather_hits_dev": (C.c_int, [_vp, _vp, _u64, _vp, C.c_int, _vp]),
    "vt_gather_wait": (C.c_int, [_vp, C.c_int, _vp]),
    "vt_gather_chunk_bounds": (None, [_u64, C.c_int, C.c_int, C.POINTER(_u64), C.POINTER(_u64)]),
    "vt_gather_hits_part_dev": (C.c_int, [_vp, _vp, _u64, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "vt_engine_last_gather_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "vt_scene_upload": (C.c_int, [_vp, _vp, _pp]),
    "vt_scene_upload_tree": (C.c_int, [_vp, _vp, _vp, _u32, _pp]),
    "vt_scene_upload_stats": (C.c_int, [_vp, _vp]),
    "vt_host_scene_download": (C.c_int, [_vp, _pp]),
    "vt_scene_free": (None, [_vp]),
    "vt_scene_device_bytes": (_u64, [_vp]),
    "vt_host_register": (C.c_int, [_vp, C.c_size_t]),
    "vt_host_unregister": (C.c_int, [_vp]),
    "vt_trace_closest": (C.c_int, [_vp, _vp, _u64, _vp]),
    "vt_trace_any": (C.c_int, [_vp, _vp, _u64, _vp]),
    "vt_trace_closest_dev": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "vt_trace_any_dev": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "vt_trace_closest_multi_dev": (C.c_int, [_vp, _vp, _u32, _vp]),
    "vt_trace_any_multi_dev": (C.c_int, [_vp, _vp, _u32, _vp]),
    "vt_trace_stats_dev": (C.c_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "vt_trace_any_stats_dev": (C.c_int, [_vp, _vp, _u64, _vp, _vp, _vp]),
    "vt_batch_trace_closest": (C.c_int, [_vp, _vp, _u64, C.POINTER(C.c_void_p)]),
    "vt_batch_trace_closest_ex": (C.c_int, [_vp, _vp, _u64, _u32, _u32, C.POINTER(_u64), C.POINTER(C.c_void_p)]),
    "vt_batch_trace_closest_set": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(_u64), C.POINTER(_u32), _u32, _u32, C.POINTER(_u32),
                                             C.POINTER(_u64), C.POINTER(C.c_void_p)]),
    "vt_batch_set_begin": (C.c_int, [_vp, _u32, C.POINTER(C.c_void_p)]),
    "vt_batch_set_add": (C.c_int, [_vp, _vp, _u64, _u32, C.POINTER(_u64)]),
    "vt_batch_set_count": (_u32, [_vp]),
    "vt_batch_set_trace": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vt_batch_set_abort": (None, [_vp]),
    "vt_batch_count": (_u64, [_vp]),
    "vt_batch_rays": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vt_batch_hits": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vt_batch_attrs": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vt_batch_shade": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vt_batch_tbn": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vt_batch_free": (None, [_vp]),
    "vt_hit_attrs_dev": (C.c_int, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "vt_scene_refit": (C.c_int, [_vp, _vp, _vp, _u32]),
    "vt_scene_set_alpha": (C.c_int, [_vp, _vp, _u32, _vp, _u64]),
    "vt_scene_set_skin": (C.c_int, [_vp, _vp, _vp, _vp, _u32]),
    "vt_scene_skin_refit": (C.c_int, [_vp, _vp, _vp, _u32]),
    "vt_scene_read_records": (C.c_int, [_vp, _vp, _vp]),
    "vt_scene_set_tri_attribs": (C.c_int, [_vp, _vp, _u32]),
    "vt_hit_shade_dev": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "vt_scene_set_tri_frames": (C.c_int, [_vp, _vp, _u32]),
    "vt_scene_read_tri_frames": (C.c_int, [_vp, _vp]),
    "vt_hit_tbn_dev": (C.c_int, [_vp, _vp, _vp, _u64, C.c_float, C.c_float, _vp, _vp]),
    "vt_gen_primary_dev": (C.c_int, [_vp, _vp, _vp, _vp]),
    "vt_gen_bounce_dev": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "vt_bounce_loop_dev": (C.c_int, [_vp, _vp, _u64, _u32, _u64, _vp, _vp, _vp]),
    "vt_engine_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "vt_engine_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "vt_engine_stream": (_vp, [_vp]),
    "vt_engine_synchronize": (C.c_int, [_vp]),
    "vt_engine_set_timing": (C.c_int, [_vp, C.c_int]),
    "vt_engine_last_kernel_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "vt_engine_launch_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), C.POINTER(_u32)]),
    "vt_test_fail_alloc": (C.c_int, [_u64]),
    "vt_test_alloc_count": (_u64, []),
    "vt_test_fail_hip": (C.c_int, [_u64]),
    "vt_test_hip_count": (_u64, []),
}


class VisTraceError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vistrace_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


def _preload_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 with the
    same sonames as /opt/rocm's.  Whichever copy is mapped first serves the whole process,
    and torch does not initialise on the system copy ("No HIP GPUs are available").  If torch
    is installed, map ITS runtime first (without importing torch) so that the order in which
    callers import torch and vistrace_amd does not matter."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def _load() -> C.CDLL:
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C vistrace_amd/csrc). vistrace_amd has no pure-Python or CPU tracing path.")
    lib = C.CDLL(LIB_PATH)
    older = bool(os.environ.get("VISTRACE_HIP_LIB")) and os.environ.get("VT_ALLOW_OLDER_ABI") == "1"   # dev: A/B against an earlier build
    for name, (res, args) in SYMBOLS.items():
        if older and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if hasattr(lib, "vt_mutant"):       # a deliberately wrong kernel (mutation testing, scripts/mutants.sh): never silently
        import sys
        print(f"[vistrace_amd] {LIB_PATH} is MUTANT {lib.vt_mutant()} of the traversal kernel: results are wrong on purpose",
              file=sys.stderr)
    return lib


lib = _load()


def hip_stream_synchronize(stream: int) -> None:
    """hipStreamSynchronize on a raw stream handle (0 = the null stream): for the few `_dev` calls whose host-side outputs arrive in
    stream order (Scene.bounce_loop_dev's live counts).  The symbol is looked up THROUGH libvistrace_hip.so (dlsym on its handle
    searches its dependencies), so it is the HIP runtime the library itself is bound to -- never a second copy of libamdhip64."""
    fn = lib.hipStreamSynchronize
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    rc = fn(stream or None)
    if rc != 0:
        raise VisTraceError(VT_ERR_HIP, f"hipStreamSynchronize failed ({rc})")


def check(rc: int) -> None:
    if rc != VT_OK:
        raise VisTraceError(rc, lib.vt_last_error().decode("utf-8", "replace"))


def ptr(a: np.ndarray) -> int:
    return a.ctypes.data
