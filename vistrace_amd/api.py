"""Thin Python objects over the C ABI (include/vistrace_hip.h).

Mirrors the shape of the reference's path: ``tris_setup`` (Primitives.h:75-102) ->
``build_bvh`` (AccelStruct.cpp:763-770) -> ``linearise`` + ``Scene`` upload
(AccelStruct.cpp:772-775) -> ``Scene.trace_closest`` (AccelStruct.cpp:818).
numpy arrays use the POD dtypes of ``_lib``; device-pointer calls take raw addresses
(e.g. ``torch.Tensor.data_ptr()``) so that torch stays plumbing only.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from ._lib import (ALPHA_MATERIAL, BVH_NODE, HIT, HIT_ATTRS, HIT_SHADE, HIT_TBN, TRI_FRAME, NODE_PAIR, RAY, RAY_STATS, SKIN_VERTEX, TRI64, TRI_ATTRIBS, check, lib,
                   ptr)

FLT_MAX = float(np.finfo(np.float32).max)


def _copy_from(addr: Optional[int], count: int, dtype: np.dtype) -> np.ndarray:
    if not addr or count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def make_rays(org, dir, tmin=0.0, tmax=FLT_MAX) -> np.ndarray:
    """Pack origins/directions (n,3) into vt_ray records; defaults = AccelStruct.cpp:790-794."""
    org = np.asarray(org, dtype=np.float32).reshape(-1, 3)
    dir = np.asarray(dir, dtype=np.float32).reshape(-1, 3)
    n = max(org.shape[0], dir.shape[0])
    rays = np.zeros(n, dtype=RAY)
    rays["org"] = org
    rays["dir"] = dir
    rays["tmin"] = np.float32(tmin) if np.isscalar(tmin) else np.asarray(tmin, np.float32)
    rays["tmax"] = np.float32(tmax) if np.isscalar(tmax) else np.asarray(tmax, np.float32)
    return rays


def tris_setup(verts: np.ndarray, flags: Optional[np.ndarray] = None) -> np.ndarray:
    """(n,3,3) float32 vertices -> vt_tri64 records in original order."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    n = verts.shape[0]
    out = np.zeros(n, dtype=TRI64)
    fl = None
    if flags is not None:
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        assert fl.shape == (n,)
    check(lib.vt_tris_setup(ptr(verts), ptr(fl) if fl is not None else None, n, ptr(out)))
    return out


class HostBvh:
    """v1-layout tree (bvh::Bvh<float>): nodes[0] = root, siblings adjacent."""

    BUILDERS = {"ploc": 0, "sah": 1, "sah_refined": 2}

    def __init__(self, tris: np.ndarray, nthreads: int = 0, builder: str = "sah"):
        """builder: "sah" = binned SAH (the default of vt_bvh_build), "ploc" = the reference's algorithm (PLOC + leaf collapse),
        "sah_refined" = binned SAH + insertion-based optimisation (opt-in: slower build, fewer steps per ray)."""
        assert tris.dtype == TRI64
        self._tris = np.ascontiguousarray(tris)
        h = C.c_void_p()
        check(lib.vt_bvh_build_ex(ptr(self._tris) if len(self._tris) else None, len(self._tris), nthreads,
                                  self.BUILDERS[builder], C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.vt_bvh_free(self._h)
            self._h = None

    @property
    def tris(self) -> np.ndarray:
        return self._tris

    def refit(self, tris: np.ndarray) -> None:
        """Keep the topology, recompute all bounds from moved triangles (same count and order)."""
        assert tris.dtype == TRI64 and len(tris) == len(self._tris)
        self._tris = np.ascontiguousarray(tris)
        check(lib.vt_bvh_refit(self._h, ptr(self._tris) if len(self._tris) else None))

    def nodes(self) -> np.ndarray:
        return _copy_from(lib.vt_bvh_nodes(self._h), lib.vt_bvh_node_count(self._h), BVH_NODE)

    def prim_indices(self) -> np.ndarray:
        return _copy_from(lib.vt_bvh_prim_indices(self._h), lib.vt_bvh_prim_count(self._h), np.dtype("<u4"))


class HostScene:
    """Linearised pairs + leaf-ordered triangle records, ready for upload."""

    def __init__(self, bvh: HostBvh):
        self.bvh = bvh
        h = C.c_void_p()
        check(lib.vt_scene_linearise(bvh._h, ptr(bvh.tris) if len(bvh.tris) else None, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.vt_host_scene_free(self._h)
            self._h = None

    @property
    def pair_count(self) -> int:
        return lib.vt_host_scene_pair_count(self._h)

    @property
    def tri_count(self) -> int:
        return lib.vt_host_scene_tri_count(self._h)

    @property
    def max_depth(self) -> int:
        return lib.vt_host_scene_max_depth(self._h)

    @property
    def root_leaf_count(self) -> int:
        return lib.vt_host_scene_root_leaf_count(self._h)

    def pairs(self) -> np.ndarray:
        return _copy_from(lib.vt_host_scene_pairs(self._h), self.pair_count, NODE_PAIR)

    # the single-ray latency path (what one accel:Traverse call does): scalar walk on the host copy -----------
    def trace_closest_host(self, rays: np.ndarray) -> np.ndarray:
        assert rays.dtype == RAY
        rays = np.ascontiguousarray(rays)
        hits = np.zeros(len(rays), dtype=HIT)
        check(lib.vt_host_scene_trace_closest(self._h, ptr(rays), len(rays), ptr(hits)))
        return hits

    def trace_any_host(self, rays: np.ndarray) -> np.ndarray:
        assert rays.dtype == RAY
        rays = np.ascontiguousarray(rays)
        occ = np.zeros(len(rays), dtype=np.uint8)
        check(lib.vt_host_scene_trace_any(self._h, ptr(rays), len(rays), ptr(occ)))
        return occ

    def set_alpha_host(self, attribs: np.ndarray, mats: np.ndarray, texels: np.ndarray) -> None:
        attribs = np.ascontiguousarray(attribs, TRI_ATTRIBS)
        mats = np.ascontiguousarray(mats, ALPHA_MATERIAL)
        texels = np.ascontiguousarray(texels, np.uint8)
        check(lib.vt_host_scene_set_alpha(self._h, ptr(attribs) if len(attribs) else None, len(attribs),
                                          ptr(mats) if len(mats) else None, len(mats),
                                          ptr(texels) if len(texels) else None, len(texels)))

    def tris(self) -> np.ndarray:
        return _copy_from(lib.vt_host_scene_tris(self._h), self.tri_count, TRI64)


def host_register(arr: np.ndarray) -> None:
    """Page-lock a host array that is reused across calls (vt_host_register): trace_closest / trace_any then copy without staging."""
    check(lib.vt_host_register(ptr(arr), arr.nbytes))


def host_unregister(arr: np.ndarray) -> None:
    check(lib.vt_host_unregister(ptr(arr)))


def device_count() -> int:
    n = C.c_int(0)
    check(lib.vt_device_count(C.byref(n)))
    return n.value


def shard_capacity(n: int, ndev: int) -> int:
    return int(lib.vt_shard_capacity(n, ndev))


def gather_chunk_bounds(count: int, nchunks: int, chunk: int):
    """Records [lo, hi) of a count-record shard that form piece `chunk` of `nchunks` (vt_gather_chunk_bounds)."""
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    lib.vt_gather_chunk_bounds(count, nchunks, chunk, C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


def shard_bounds(n: int, ndev: int, g: int):
    """Contiguous shard [lo, hi) of an n-ray batch for device g of ndev (vt_shard_bounds)."""
    lo, hi = C.c_uint64(0), C.c_uint64(0)
    lib.vt_shard_bounds(n, ndev, g, C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


class Batch:
    """One traced batch resident on the device (vt_batch): arrays are downloaded once, on first use."""

    def __init__(self, handle, scene):
        self._h, self._scene = handle, scene      # the scene (and its engine) must outlive the device arrays

    def __len__(self) -> int:
        return int(lib.vt_batch_count(self._h))

    def _fetch(self, fn, dtype) -> np.ndarray:
        p = C.c_void_p()
        check(fn(self._h, C.byref(p)))
        return _copy_from(p.value, len(self), dtype)

    def rays(self) -> np.ndarray:
        return self._fetch(lib.vt_batch_rays, RAY)

    def hits(self) -> np.ndarray:
        return self._fetch(lib.vt_batch_hits, HIT)

    def attrs(self) -> np.ndarray:
        return self._fetch(lib.vt_batch_attrs, HIT_ATTRS)

    def shade(self) -> np.ndarray:
        return self._fetch(lib.vt_batch_shade, HIT_SHADE)

    def tbn(self) -> np.ndarray:
        """Shading frame per hit (TraceResult::GetNormal / GetTangent / GetBinormal without a normal map), cone off."""
        return self._fetch(lib.vt_batch_tbn, HIT_TBN)

    def free(self) -> None:
        if self._h:
            lib.vt_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def comm_unique_id() -> bytes:
    """128-byte RCCL id for vt_engine_comm_init_rank (rank 0 creates it, the launcher distributes it)."""
    buf = (C.c_char * 128)()
    check(lib.vt_comm_unique_id(buf))
    return bytes(buf)


class Engine:
    def __init__(self, device=0):
        """device: a HIP device index, or a list of them = one single-process multi-GPU group (root = first)."""
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            arr = (C.c_int * len(device))(*device)
            check(lib.vt_engine_open_multi(arr, len(device), C.byref(h)))
            self.devices = list(device)
            device = device[0]
        else:
            check(lib.vt_engine_open(device, C.byref(h)))
            self.devices = [device]
        self._h = h
        self.device = device

    @property
    def device_count(self) -> int:
        return int(lib.vt_engine_device_count(self._h))

    def member_kernel_ms(self, g: int) -> float:
        """Duration of the latest timed trace launch of group member g (vt_engine_member + vt_engine_last_kernel_ms)."""
        m = lib.vt_engine_member(self._h, g)
        if not m:
            raise IndexError(f"group member {g} out of range")
        ms = C.c_float(0)
        check(lib.vt_engine_last_kernel_ms(C.c_void_p(m), C.byref(ms)))
        return float(ms.value)

    # one process per GPU: native RCCL gather of hit records (vt_gather_hits_dev)
    def comm_init_rank(self, nranks: int, rank: int, unique_id: bytes) -> None:
        assert len(unique_id) == 128
        check(lib.vt_engine_comm_init_rank(self._h, nranks, rank, unique_id))

    def gather_hits_dev(self, d_send: int, count: int, d_recv_root: int, root: int = 0, stream: int = 0) -> None:
        check(lib.vt_gather_hits_dev(self._h, d_send, count, d_recv_root or None, root, stream or None))

    def gather_hits_part_dev(self, d_send: int, count: int, chunk: int, nchunks: int, d_recv_root: int, root: int = 0, stream: int = 0) -> None:
        """Piece `chunk` of `nchunks` of every rank's count-record shard to the root (vt_gather_hits_part_dev)."""
        check(lib.vt_gather_hits_part_dev(self._h, d_send, count, chunk, nchunks, d_recv_root or None, root, stream or None))

    def gather_wait(self, batches_in_flight: int = 0, stream: int = 0) -> None:
        check(lib.vt_gather_wait(self._h, batches_in_flight, stream or None))

    def last_gather_ms(self) -> float:
        """Duration of the latest vt_gather_hits_dev on the communication stream (needs set_timing(True))."""
        ms = C.c_float(0)
        check(lib.vt_engine_last_gather_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            lib.vt_engine_close(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_option(self, key: str, value: int) -> None:
        check(lib.vt_engine_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64(0)
        check(lib.vt_engine_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    @property
    def stream(self) -> int:
        """hipStream_t of the engine's private stream (host-buffer calls run there)."""
        return int(lib.vt_engine_stream(self._h) or 0)

    def synchronize(self) -> None:
        check(lib.vt_engine_synchronize(self._h))

    def set_timing(self, on: bool) -> None:
        check(lib.vt_engine_set_timing(self._h, 1 if on else 0))

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        check(lib.vt_engine_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def gen_primary_dev(self, width: int, height: int, d_rays: int, pos=(0.0, 0.0, 0.0), forward=(1.0, 0.0, 0.0),
                        up=(0.0, 0.0, 1.0), vfov_deg: float = 60.0, stream: int = 0) -> None:
        """width*height pinhole rays written to device memory (same camera model as workloads.primary_rays)."""
        cam = np.zeros(1, _lib.CAMERA)
        cam["pos"], cam["forward"], cam["up"], cam["vfov_deg"] = pos, forward, up, vfov_deg
        cam["width"], cam["height"] = width, height
        check(lib.vt_gen_primary_dev(self._h, ptr(cam), d_rays, stream or None))

    def gen_bounce_dev(self, d_attrs: int, n: int, seed: int, d_rays: int, stream: int = 0) -> None:
        """One cosine-hemisphere bounce ray per vt_hit_attrs record (same construction as workloads.bounce_rays)."""
        check(lib.vt_gen_bounce_dev(self._h, d_attrs, n, seed & 0xFFFFFFFFFFFFFFFF, d_rays, stream or None))

    def launch_info(self) -> dict:
        b, t, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(lib.vt_engine_launch_info(self._h, C.byref(b), C.byref(t), C.byref(l)))
        return {"blocks": b.value, "threads": t.value, "lds_bytes": l.value}


class Scene:
    """Device-resident scene: upload once per Rebuild, trace many batches."""

    def __init__(self, engine: Engine, host_scene: HostScene):
        self.engine = engine
        self.host_scene = host_scene
        h = C.c_void_p()
        check(lib.vt_scene_upload(engine._h, host_scene._h, C.byref(h)))
        self._h = h

    @classmethod
    def from_tree(cls, engine: Engine, bvh: HostBvh) -> "Scene":
        """vt_scene_upload_tree: the tree and the triangle records go up as they are, the device numbers the pairs, shuffles the
        triangles into leaf order and derives the index tables (no host walk).  `host_scene` is fetched on first use."""
        self = cls.__new__(cls)
        self.engine = engine
        self.host_scene = None
        h = C.c_void_p()
        check(lib.vt_scene_upload_tree(engine._h, bvh._h, ptr(bvh.tris) if len(bvh.tris) else None, len(bvh.tris), C.byref(h)))
        self._h = h
        self._bvh = bvh
        return self

    def download_host_scene(self) -> HostScene:
        """vt_host_scene_download: the host-side copy of this scene (single-ray path, accessors)."""
        h = C.c_void_p()
        check(lib.vt_host_scene_download(self._h, C.byref(h)))
        hs = HostScene.__new__(HostScene)
        hs.bvh = getattr(self, "_bvh", None)
        hs._h = h
        self.host_scene = hs
        return hs

    def upload_stats(self) -> dict:
        st = np.zeros(1, dtype=_lib.UPLOAD_STATS)
        check(lib.vt_scene_upload_stats(self._h, ptr(st)))
        return {k: (float(st[0][k]) if st.dtype[k].kind == "f" else int(st[0][k])) for k in st.dtype.names if k != "pad"}

    def free(self):
        if getattr(self, "_h", None):
            lib.vt_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.free()

    @property
    def device_bytes(self) -> int:
        return lib.vt_scene_device_bytes(self._h)

    # host-buffer entry points ------------------------------------------------------------
    def trace_closest(self, rays: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        assert rays.dtype == RAY
        rays = np.ascontiguousarray(rays)
        hits = np.zeros(len(rays), dtype=HIT) if out is None else out
        assert hits.dtype == HIT and len(hits) == len(rays) and hits.flags.c_contiguous
        check(lib.vt_trace_closest(self._h, ptr(rays), len(rays), ptr(hits)))
        return hits

    def trace_any(self, rays: np.ndarray) -> np.ndarray:
        assert rays.dtype == RAY
        rays = np.ascontiguousarray(rays)
        occ = np.zeros(len(rays), dtype=np.uint8)
        check(lib.vt_trace_any(self._h, ptr(rays), len(rays), ptr(occ)))
        return occ

    # device-pointer entry points (addresses of buffers on this engine's device) -----------
    def trace_closest_dev(self, d_rays: int, n: int, d_hits: int, stream: int = 0) -> None:
        check(lib.vt_trace_closest_dev(self._h, d_rays, n, d_hits, stream or None))

    def trace_multi_dev(self, batches, stream: int = 0, any_hit: bool = False) -> None:
        """ONE launch over several batches (vt_trace_closest_multi_dev / vt_trace_any_multi_dev): batches = sequence of
        (d_rays, d_out, n[, ray_image_width]) with device addresses; results equal those of separate calls."""
        desc = np.zeros(len(batches), dtype=_lib.BATCH_DESC)
        for k, b in enumerate(batches):
            desc[k]["d_rays"], desc[k]["d_out"], desc[k]["n"] = b[0] or 0, b[1] or 0, b[2]
            desc[k]["ray_image_width"] = b[3] if len(b) > 3 else 0
        fn = lib.vt_trace_any_multi_dev if any_hit else lib.vt_trace_closest_multi_dev
        check(fn(self._h, ptr(desc) if len(desc) else None, len(desc), stream or None))

    def trace_closest_gather_dev(self, d_rays_per_device, n: int, d_hits_root: int) -> None:
        """Multi-GPU group: d_rays_per_device[g] = address (on device g) of shard g's rays; the hit records of all
        shards are gathered to d_hits_root on the root device (ndev * shard_capacity records, ray i at record i).
        Asynchronous; engine.synchronize() waits for traces and gather."""
        arr = (C.c_void_p * len(d_rays_per_device))(*[C.c_void_p(p or None) for p in d_rays_per_device])
        check(lib.vt_trace_closest_gather_dev(self._h, arr, n, d_hits_root))

    def bounce_loop_dev(self, d_rays: int, n: int, depth: int, seed: int, d_hits: int, stream: int = 0) -> list:
        """Device-resident bounce loop: depth x n hit records (row d = hits of bounce d, indexed by path).
        Returns the number of live paths traced at each depth."""
        live = (C.c_uint64 * max(depth, 1))()
        check(lib.vt_bounce_loop_dev(self._h, d_rays, n, depth, seed & 0xFFFFFFFFFFFFFFFF, d_hits, live, stream or None))
        # the call only enqueues; the counts arrive in stream order (a host function behind the last depth)
        _lib.hip_stream_synchronize(stream)
        return [int(x) for x in live[:depth]]

    def trace_any_dev(self, d_rays: int, n: int, d_occ: int, stream: int = 0) -> None:
        check(lib.vt_trace_any_dev(self._h, d_rays, n, d_occ, stream or None))

    def trace_stats_dev(self, d_rays: int, n: int, d_hits: int, d_stats: int, stream: int = 0) -> None:
        check(lib.vt_trace_stats_dev(self._h, d_rays, n, d_hits, d_stats, stream or None))

    def trace_batch(self, rays: np.ndarray, image_width: Optional[int] = None, check_ranges: bool = False,
                    fetch_hits: bool = False) -> "Batch":
        """vt_batch_trace_closest[_ex]: the batch stays on the device; hits / attrs / shade come back when first asked for.
        check_ranges: the range checks of AccelStruct::Traverse on every ray while it is staged (ValueError naming the first
        offender); fetch_hits: the hit records come back behind the trace; image_width: camera rays, rays per image row."""
        rays = np.ascontiguousarray(rays, RAY)
        h = C.c_void_p()
        if image_width is None and not check_ranges and not fetch_hits:
            check(lib.vt_batch_trace_closest(self._h, ptr(rays) if len(rays) else None, len(rays), C.byref(h)))
            return Batch(h, self)
        bad = C.c_uint64(len(rays))
        rc = lib.vt_batch_trace_closest_ex(self._h, ptr(rays) if len(rays) else None, len(rays), image_width or 0,
                                           (1 if check_ranges else 0) | (2 if fetch_hits else 0), C.byref(bad), C.byref(h))
        if rc != _lib.VT_OK and bad.value < len(rays):
            raise ValueError(f"ray {bad.value}: tMin < 0 or tMax <= tMin")
        check(rc)
        return Batch(h, self)

    def trace_batch_set(self, ray_sets, image_widths=None, check_ranges: bool = False, fetch_hits: bool = False) -> list:
        """vt_batch_trace_closest_set: several host ray arrays -> one merged launch -> one Batch per array."""
        arrs = [np.ascontiguousarray(r, RAY) for r in ray_sets]
        nb = len(arrs)
        ptrs = (C.c_void_p * max(nb, 1))(*[C.c_void_p(ptr(a) if len(a) else None) for a in arrs])
        ns = (C.c_uint64 * max(nb, 1))(*[len(a) for a in arrs])
        ws = (C.c_uint32 * max(nb, 1))(*[int(w) for w in image_widths]) if image_widths is not None else None
        outs = (C.c_void_p * max(nb, 1))()
        bad_b, bad_r = C.c_uint32(nb), C.c_uint64(0)
        rc = lib.vt_batch_trace_closest_set(self._h, ptrs, ns, ws, nb, (1 if check_ranges else 0) | (2 if fetch_hits else 0),
                                            C.byref(bad_b), C.byref(bad_r), outs)
        if rc != _lib.VT_OK and bad_b.value < nb:
            raise ValueError(f"batch {bad_b.value}, ray {bad_r.value}: tMin < 0 or tMax <= tMin")
        check(rc)
        return [Batch(C.c_void_p(outs[k]), self) for k in range(nb)]

    def trace_any_stats_dev(self, d_rays: int, n: int, d_occ: int, d_stats: int, stream: int = 0) -> None:
        check(lib.vt_trace_any_stats_dev(self._h, d_rays, n, d_occ, d_stats, stream or None))

    def hit_attrs_dev(self, d_rays: int, d_hits: int, n: int, d_attrs: int, stream: int = 0) -> None:
        check(lib.vt_hit_attrs_dev(self._h, d_rays, d_hits, n, d_attrs, stream or None))

    def refit(self, verts: np.ndarray, flags: Optional[np.ndarray] = None) -> None:
        """Device-side refit in place: (n,3,3) vertices in original triangle order, same topology."""
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        fl = np.ascontiguousarray(flags, dtype=np.uint8) if flags is not None else None
        check(lib.vt_scene_refit(self._h, ptr(verts) if len(verts) else None, ptr(fl) if fl is not None else None, len(verts)))

    def set_alpha(self, mats: np.ndarray, texels: np.ndarray) -> None:
        """Alpha-test side data: ALPHA_MATERIAL records (indexed by TRI_ATTRIBS.material) and their 8-bit alpha planes."""
        mats = np.ascontiguousarray(mats, ALPHA_MATERIAL)
        texels = np.ascontiguousarray(texels, np.uint8)
        check(lib.vt_scene_set_alpha(self._h, ptr(mats) if len(mats) else None, len(mats), ptr(texels) if len(texels) else None,
                                     len(texels)))

    def set_skin(self, bind_verts: np.ndarray, skin: np.ndarray, matrix_base: np.ndarray) -> None:
        """Bind-pose triangles (n,3,3), their (n,3) SKIN_VERTEX records and each triangle's first-matrix index."""
        bind_verts = np.ascontiguousarray(bind_verts, np.float32).reshape(-1, 9)
        n = len(bind_verts)
        skin = np.ascontiguousarray(skin, SKIN_VERTEX).reshape(n * 3)
        matrix_base = np.ascontiguousarray(matrix_base, np.uint32).reshape(n)
        check(lib.vt_scene_set_skin(self._h, ptr(bind_verts) if n else None, ptr(skin) if n else None,
                                    ptr(matrix_base) if n else None, n))

    def skin_refit(self, bones: np.ndarray, binds: np.ndarray) -> None:
        """Per frame: bone and bind matrices (nmat, 16) column-major -> skin, rebuild records, refit, on the device."""
        bones = np.ascontiguousarray(bones, np.float32).reshape(-1, 16)
        binds = np.ascontiguousarray(binds, np.float32).reshape(-1, 16)
        if bones.shape != binds.shape:
            raise ValueError("bones and binds must have the same shape")
        check(lib.vt_scene_skin_refit(self._h, ptr(bones) if len(bones) else None, ptr(binds) if len(binds) else None, len(bones)))

    def sync_host_scene(self) -> None:
        """After refit / skin_refit: read the device records back into the host scene the single-ray path walks."""
        if self.host_scene is None:                 # from_tree: no host copy yet -- fetching one IS the sync
            self.download_host_scene()
            return
        check(lib.vt_host_scene_sync(self.host_scene._h, self._h))

    def read_records(self):
        """(pairs, tris) as they currently are on the device."""
        if self.host_scene is None:
            self.download_host_scene()
        pairs = np.zeros(self.host_scene.pair_count, dtype=NODE_PAIR)
        tris = np.zeros(self.host_scene.tri_count, dtype=TRI64)
        check(lib.vt_scene_read_records(self._h, ptr(pairs) if len(pairs) else None, ptr(tris) if len(tris) else None))
        return pairs, tris

    def set_tri_attribs(self, attribs: np.ndarray) -> None:
        """Per-triangle uvs / alphas / entity id / material (original order) for hit_shade_dev."""
        assert attribs.dtype == TRI_ATTRIBS
        attribs = np.ascontiguousarray(attribs)
        check(lib.vt_scene_set_tri_attribs(self._h, ptr(attribs) if len(attribs) else None, len(attribs)))

    def hit_shade_dev(self, d_hits: int, n: int, d_out: int, stream: int = 0) -> None:
        check(lib.vt_hit_shade_dev(self._h, d_hits, n, d_out, stream or None))

    def set_tri_frames(self, frames: np.ndarray) -> None:
        """Per-vertex normals / tangents (original order; the bind pose when the scene is skinned) for hit_tbn_dev."""
        assert frames.dtype == TRI_FRAME
        frames = np.ascontiguousarray(frames)
        check(lib.vt_scene_set_tri_frames(self._h, ptr(frames) if len(frames) else None, len(frames)))

    def read_tri_frames(self) -> np.ndarray:
        if self.host_scene is None:
            self.download_host_scene()
        frames = np.zeros(self.host_scene.tri_count, dtype=TRI_FRAME)
        check(lib.vt_scene_read_tri_frames(self._h, ptr(frames) if len(frames) else None))
        return frames

    def hit_tbn_dev(self, d_rays: int, d_hits: int, n: int, d_out: int, cone_width: float = -1.0, cone_angle: float = -1.0,
                    stream: int = 0) -> None:
        """n x HIT_TBN: TraceResult::CalcTBN (no normal map) + CalcFootprint (TraceResult.cpp:89-103, 132-186)."""
        check(lib.vt_hit_tbn_dev(self._h, d_rays, d_hits, n, cone_width, cone_angle, d_out, stream or None))


def build_scene(engine: Engine, verts: np.ndarray, flags: Optional[np.ndarray] = None, nthreads: int = 0) -> Scene:
    """verts (n,3,3) -> setup -> build (default builder) -> linearise -> upload."""
    tris = tris_setup(verts, flags)
    return Scene(engine, HostScene(HostBvh(tris, nthreads)))


__all__ = ["Engine", "Scene", "HostBvh", "HostScene", "tris_setup", "build_scene", "make_rays", "device_count",
           "shard_capacity", "shard_bounds", "gather_chunk_bounds", "comm_unique_id", "host_register", "host_unregister",
           "RAY", "HIT", "TRI64", "BVH_NODE", "NODE_PAIR", "RAY_STATS", "HIT_ATTRS", "TRI_ATTRIBS", "HIT_SHADE", "HIT_TBN", "TRI_FRAME", "SKIN_VERTEX", "ALPHA_MATERIAL",
           "FLT_MAX", "_lib"]
