"""ctypes binding of oracle/_build/libvt_oracle.so (see oracle/vt_oracle.h).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, from __graft_entry__.smoke() and
from bench.py's cpu_baseline leg -- never from the vistrace_amd package.
Parity status: "parity unpinned" (the reference holds no golden vectors for this path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libvt_oracle.so")
FAST_LIB_PATH = os.path.join(_HERE, "_build", "libvt_oracle_fast.so")   # -O3 -march=x86-64-v3: bench.py's timed leg only

MISS = 0xFFFFFFFF
RAY = np.dtype([("org", "<f4", 3), ("dir", "<f4", 3), ("tmin", "<f4"), ("tmax", "<f4")])
TRI = np.dtype([("p0", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("n", "<f4", 3), ("flags", "<u4")])
NODE = np.dtype([("bounds", "<f4", 6), ("prim_count", "<u4"), ("first", "<u4")])
HIT = np.dtype([("prim", "<u4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])
ALPHA_MATERIAL = np.dtype([("tex_mat", "<f4", (2, 4)), ("tex_scale", "<f4"), ("alpha_ref", "<f4"), ("width", "<u4"),
                           ("height", "<u4"), ("filter", "<u4"), ("pad", "<u4"), ("offset", "<u8")])
SKIN_VERTEX = np.dtype([("weight", "<f4", 3), ("bone", "i1", 3), ("num_bones", "u1")])
TBN = np.dtype([("normal", "<f4", 3), ("tangent", "<f4", 3), ("binormal", "<f4", 3), ("lod_info", "<f4", 2), ("lod_set", "<u4")])
ATTRS = np.dtype([("pos", "<f4", 3), ("uvw", "<f4", 3), ("ngeo", "<f4", 3), ("wo", "<f4", 3), ("front", "<u4")])


class _Stats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("tests", C.c_uint64)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(FAST_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None
_fast = None


def _bind_ctx(L):
    vp, u64 = C.c_void_p, C.c_uint64
    L.vto_batch_ctx_create.argtypes = [vp, u64, vp, u64, vp, u64, C.c_int]
    L.vto_batch_ctx_create.restype = vp
    L.vto_batch_ctx_replicas.argtypes = [vp]
    L.vto_batch_ctx_replicas.restype = C.c_int
    L.vto_batch_ctx_destroy.argtypes = [vp]
    L.vto_traverse_batch_ctx.argtypes = [vp, vp, u64, C.c_int, vp, vp, vp, C.c_int]
    L.vto_traverse_batch_ctx.restype = C.c_int


def _cpu_has_avx2() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = line.split(":", 1)[1].split()
                    return "avx2" in fl and "bmi2" in fl and "fma" in fl
    except OSError:
        pass
    return False


def fast_lib():
    """The -O3 -march=x86-64-v3 build of the same source (timed cpu_baseline leg); None if the host CPU lacks AVX2."""
    global _fast
    if _fast is None:
        build()
        if not _cpu_has_avx2():
            return None
        L = C.CDLL(FAST_LIB_PATH)
        _bind_ctx(L)
        _fast = L
    return _fast


class BatchContext:
    """NUMA-aware batch driver of the timed CPU baseline: one replica of the tree per NUMA node (first-touched there),
    every worker walks its node's replica.  `fast` picks the -O3 x86-64-v3 build.  Results equal traverse_batch's."""

    def __init__(self, nodes: np.ndarray, prim_indices: np.ndarray, tris: np.ndarray, nthreads: int = 0, fast: bool = True):
        assert nodes.dtype == NODE and tris.dtype == TRI
        self._L = (fast_lib() if fast else None) or lib()
        self.fast = self._L is not lib()
        nodes = np.ascontiguousarray(nodes)
        prim_indices = np.ascontiguousarray(prim_indices, np.uint32)
        tris = np.ascontiguousarray(tris)
        self._h = self._L.vto_batch_ctx_create(nodes.ctypes.data, len(nodes), prim_indices.ctypes.data, len(prim_indices),
                                               tris.ctypes.data, len(tris), nthreads)
        if not self._h:
            raise MemoryError("vto_batch_ctx_create failed")
        self.replicas = int(self._L.vto_batch_ctx_replicas(self._h))

    def traverse(self, rays: np.ndarray, any_hit: bool = False, want_stats: bool = False, nthreads: int = 0):
        rays = np.ascontiguousarray(rays)
        hits = np.zeros(len(rays), HIT)
        st = np.zeros((len(rays), 2), np.uint32) if want_stats else None
        tot = _Stats()
        used = self._L.vto_traverse_batch_ctx(self._h, rays.ctypes.data, len(rays), int(any_hit), hits.ctypes.data,
                                              st.ctypes.data if st is not None else None, C.addressof(tot), nthreads)
        return hits, st, int(tot.steps), int(tot.tests), used

    def close(self):
        if getattr(self, "_h", None):
            self._L.vto_batch_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.vto_tri_setup.argtypes = [vp, vp, vp, u32, vp]
        L.vto_tri_intersect.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
        L.vto_tri_intersect.restype = C.c_int
        L.vto_trace_brute.argtypes = [vp, u32, vp, u64, C.c_int, vp, C.c_int]
        L.vto_min_t_set.argtypes = [vp, u32, vp, vp, vp, u32]
        L.vto_min_t_set.restype = u32
        L.vto_traverse.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
        L.vto_traverse.restype = C.c_int
        L.vto_traverse_batch.argtypes = [vp, vp, vp, vp, u64, C.c_int, vp, vp, vp, C.c_int]
        L.vto_traverse_batch.restype = C.c_int
        _bind_ctx(L)
        L.vto_hit_attrs.argtypes = [vp, vp, C.c_float, C.c_float, vp]
        L.vto_hit_shade.argtypes = [C.c_float, C.c_float, vp, vp, vp, vp]
        L.vto_calc_ray_origin.argtypes = [vp, vp, vp]
        L.vto_hemisphere_cos.argtypes = [C.c_float, C.c_float, vp]
        L.vto_set_alpha.argtypes = [vp]
        L.vto_alpha_sample.argtypes = [vp, vp, C.c_float, C.c_float]
        L.vto_alpha_sample.restype = C.c_float
        L.vto_alpha_pass.argtypes = [vp, u32, C.c_float, C.c_float]
        L.vto_alpha_pass.restype = C.c_int
        L.vto_skin_matrices.argtypes = [vp, vp, u32, vp]
        L.vto_skin_verts.argtypes = [vp, vp, vp, u32, vp, vp]
        L.vto_hit_tbn.argtypes = [vp, vp, C.c_float, C.c_float, C.c_float, vp, vp, vp, C.c_float, C.c_float, vp]
        L.vto_skin_frames.argtypes = [vp, vp, vp, u32, vp, vp]
        L.vto_alt_mask.restype = C.c_uint32
        if L.vto_alt_mask() != 0:
            raise RuntimeError("oracle/_build/libvt_oracle.so was built with a VTO_ALT switch: not the oracle proper")
        _lib = L
    return _lib


def tris_setup(verts: np.ndarray, flags=None) -> np.ndarray:
    """(n,3,3) float32 -> oracle triangle records (Primitives.h:75-102)."""
    verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3, 3)
    out = np.zeros(len(verts), TRI)
    L = lib()
    for i in range(len(verts)):
        f = int(flags[i]) if flags is not None else 0
        L.vto_tri_setup(verts[i, 0].ctypes.data, verts[i, 1].ctypes.data, verts[i, 2].ctypes.data, f,
                        out[i:i + 1].ctypes.data)
    return out


def tris_from_tri64(tri64: np.ndarray) -> np.ndarray:
    """Re-use records set up by the product (fields are compared separately in tests)."""
    out = np.zeros(len(tri64), TRI)
    for k in ("p0", "e1", "e2", "n", "flags"):
        out[k] = tri64[k]
    return out


def trace_brute(tris: np.ndarray, rays: np.ndarray, any_hit: bool = False, nthreads: int = 0) -> np.ndarray:
    assert tris.dtype == TRI
    rays = np.ascontiguousarray(rays).view(RAY) if rays.dtype != RAY else np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT)
    lib().vto_trace_brute(tris.ctypes.data, len(tris), rays.ctypes.data, len(rays), int(any_hit), hits.ctypes.data,
                          nthreads)
    return hits


def min_t_set(tris: np.ndarray, ray: np.ndarray, max_ids: int = 16):
    ids = np.zeros(max_ids, np.uint32)
    t = C.c_float(0)
    ray = np.ascontiguousarray(ray)
    n = lib().vto_min_t_set(tris.ctypes.data, len(tris), ray.ctypes.data, C.addressof(t), ids.ctypes.data, max_ids)
    return t.value, ids[:min(n, max_ids)].copy(), n


ALT_NAMES = ("PLAIN_INVERSE", "SWAP_GE", "FMA", "RETEST_RIGHT", "LEAF_DESC", "ACCEPT_LT", "PUSH_NODE_CULL", "FMINMAX")
_alts = {}


def alt_lib(name: str) -> C.CDLL:
    """A recall-sensitivity VARIANT of the oracle (vt_oracle.c built with -DVTO_ALT_<name>, `make -C oracle alts`): one
    recalled bvh-v1 detail read the other way.  For scripts/recall_sensitivity.py and its test only -- never the checker."""
    if name not in ALT_NAMES:
        raise ValueError(name)
    if name not in _alts:
        path = os.path.join(_HERE, "_build", "alt", f"libvt_oracle_{name}.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "alts"], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        vp, u64 = C.c_void_p, C.c_uint64
        L.vto_traverse_batch.argtypes = [vp, vp, vp, vp, u64, C.c_int, vp, vp, vp, C.c_int]
        L.vto_traverse_batch.restype = C.c_int
        L.vto_alt_mask.restype = C.c_uint32
        assert L.vto_alt_mask() == 1 << ALT_NAMES.index(name), name
        _alts[name] = L
    return _alts[name]


def traverse_batch(nodes: np.ndarray, prim_indices: np.ndarray, tris: np.ndarray, rays: np.ndarray,
                   any_hit: bool = False, want_stats: bool = False, nthreads: int = 0, L=None):
    """Returns (hits, per_ray_stats or None, total_steps, total_tests, threads_used).  `L`: an alt_lib() variant
    (recall-sensitivity counts only); the default is the oracle proper."""
    assert nodes.dtype == NODE and tris.dtype == TRI
    nodes = np.ascontiguousarray(nodes)
    prim_indices = np.ascontiguousarray(prim_indices, np.uint32)
    rays = np.ascontiguousarray(rays)
    hits = np.zeros(len(rays), HIT)
    st = np.zeros((len(rays), 2), np.uint32) if want_stats else None
    tot = _Stats()
    used = (L or lib()).vto_traverse_batch(nodes.ctypes.data, prim_indices.ctypes.data, tris.ctypes.data, rays.ctypes.data,
                                    len(rays), int(any_hit), hits.ctypes.data,
                                    st.ctypes.data if st is not None else None, C.addressof(tot), nthreads)
    return hits, st, int(tot.steps), int(tot.tests), used


def hit_attrs(tris: np.ndarray, rays: np.ndarray, hits: np.ndarray) -> np.ndarray:
    out = np.zeros(len(hits), ATTRS)
    L = lib()
    for i in range(len(hits)):
        if hits["prim"][i] == MISS:
            continue
        L.vto_hit_attrs(tris[int(hits["prim"][i]):].ctypes.data, rays["dir"][i].ctypes.data, float(hits["u"][i]),
                        float(hits["v"][i]), out[i:i + 1].ctypes.data)
    return out


def hit_shade(u: float, v: float, uvs, alphas):
    """(texUV[2], blendFactor) per TraceResult.cpp:70,73-74."""
    uvs = np.ascontiguousarray(uvs, np.float32).reshape(6)
    alphas = np.ascontiguousarray(alphas, np.float32).reshape(3)
    tex = np.zeros(2, np.float32)
    blend = C.c_float(0)
    lib().vto_hit_shade(float(u), float(v), uvs.ctypes.data, alphas.ctypes.data, tex.ctypes.data, C.addressof(blend))
    return tex, np.float32(blend.value)


def calc_ray_origin(pos, normal) -> np.ndarray:
    pos = np.ascontiguousarray(pos, np.float32)
    normal = np.ascontiguousarray(normal, np.float32)
    out = np.zeros(3, np.float32)
    lib().vto_calc_ray_origin(pos.ctypes.data, normal.ctypes.data, out.ctypes.data)
    return out


def hemisphere_cos(r1: float, r2: float) -> np.ndarray:
    out = np.zeros(3, np.float32)
    lib().vto_hemisphere_cos(float(r1), float(r2), out.ctypes.data)
    return out


def skin_matrices(bones: np.ndarray, binds: np.ndarray) -> np.ndarray:
    """bones[i] * binds[i] (glm mat4, column-major 16 floats each), AccelStruct.cpp:44."""
    bones = np.ascontiguousarray(bones, np.float32).reshape(-1, 16)
    binds = np.ascontiguousarray(binds, np.float32).reshape(-1, 16)
    assert bones.shape == binds.shape
    out = np.zeros_like(bones)
    lib().vto_skin_matrices(bones.ctypes.data, binds.ctypes.data, bones.shape[0], out.ctypes.data)
    return out


def skin_verts(bind_verts: np.ndarray, skin: np.ndarray, matrix_base: np.ndarray, mats: np.ndarray) -> np.ndarray:
    """SkinTriangle (AccelStruct.cpp:66-102) positions for every triangle: n x 9 floats."""
    bind_verts = np.ascontiguousarray(bind_verts, np.float32).reshape(-1, 9)
    n = bind_verts.shape[0]
    skin = np.ascontiguousarray(skin, SKIN_VERTEX).reshape(n * 3)
    matrix_base = np.ascontiguousarray(matrix_base, np.uint32).reshape(n)
    mats = np.ascontiguousarray(mats, np.float32).reshape(-1, 16)
    out = np.zeros((n, 9), np.float32)
    lib().vto_skin_verts(bind_verts.ctypes.data, skin.ctypes.data, matrix_base.ctypes.data, n, mats.ctypes.data,
                         out.ctypes.data)
    return out


def hit_tbn(tris: np.ndarray, rays: np.ndarray, hits: np.ndarray, frames: np.ndarray, uvs: np.ndarray,
            cone_width: float = -1.0, cone_angle: float = -1.0) -> np.ndarray:
    """TraceResult::CalcTBN (no normal map) + CalcFootprint per hit; frames: ntris x 18 floats (normals[3][3],
    tangents[3][3]), uvs: ntris x 6.  Misses stay zero."""
    frames = np.ascontiguousarray(frames, np.float32).reshape(-1, 18)
    uvs = np.ascontiguousarray(uvs, np.float32).reshape(-1, 6)
    out = np.zeros(len(hits), TBN)
    L = lib()
    for i in range(len(hits)):
        p = int(hits["prim"][i])
        if p == MISS:
            continue
        L.vto_hit_tbn(tris[p:].ctypes.data, rays["dir"][i].ctypes.data, float(hits["t"][i]), float(hits["u"][i]),
                      float(hits["v"][i]), frames[p].ctypes.data, frames[p, 9:].ctypes.data, uvs[p].ctypes.data,
                      float(cone_width), float(cone_angle), out[i:i + 1].ctypes.data)
    return out


def skin_frames(bind_frames: np.ndarray, skin: np.ndarray, matrix_base: np.ndarray, mats: np.ndarray) -> np.ndarray:
    """SkinTriangle's normals / tangents (AccelStruct.cpp:82-92, angleOnly): n x 18 floats."""
    bind_frames = np.ascontiguousarray(bind_frames, np.float32).reshape(-1, 18)
    n = bind_frames.shape[0]
    skin = np.ascontiguousarray(skin, SKIN_VERTEX).reshape(n * 3)
    matrix_base = np.ascontiguousarray(matrix_base, np.uint32).reshape(n)
    mats = np.ascontiguousarray(mats, np.float32).reshape(-1, 16)
    out = np.zeros((n, 18), np.float32)
    lib().vto_skin_frames(bind_frames.ctypes.data, skin.ctypes.data, matrix_base.ctypes.data, n, mats.ctypes.data,
                          out.ctypes.data)
    return out


class _AlphaCtx(C.Structure):
    _fields_ = [("tris_base", C.c_void_p), ("tri_uv", C.c_void_p), ("tri_material", C.c_void_p), ("mats", C.c_void_p),
                ("nmats", C.c_uint32), ("texels", C.c_void_p)]


_alpha_keep = None


def set_alpha(tris=None, tri_uv=None, tri_material=None, mats=None, texels=None):
    """Install (or with no arguments clear) the alpha-test side data used for triangles flagged TRI_ALPHATEST.
    `tris` must be the very array later passed to traverse_batch / trace_brute (its address names the triangles)."""
    global _alpha_keep
    if tris is None:
        lib().vto_set_alpha(None)
        _alpha_keep = None
        return None
    assert tris.dtype == TRI and tris.flags.c_contiguous
    tri_uv = np.ascontiguousarray(tri_uv, np.float32).reshape(len(tris), 6)
    tri_material = np.ascontiguousarray(tri_material, np.uint32).reshape(len(tris))
    mats = np.ascontiguousarray(mats, ALPHA_MATERIAL)
    texels = np.ascontiguousarray(texels, np.uint8)
    ctx = _AlphaCtx(tris.ctypes.data, tri_uv.ctypes.data, tri_material.ctypes.data, mats.ctypes.data, len(mats), texels.ctypes.data)
    _alpha_keep = (ctx, tris, tri_uv, tri_material, mats, texels)
    lib().vto_set_alpha(C.addressof(ctx))
    return ctx


def alpha_sample(mat: np.ndarray, texels: np.ndarray, s: float, t: float) -> float:
    mat = np.ascontiguousarray(mat, ALPHA_MATERIAL).reshape(1)
    texels = np.ascontiguousarray(texels, np.uint8)
    return float(lib().vto_alpha_sample(mat.ctypes.data, texels.ctypes.data, float(s), float(t)))
