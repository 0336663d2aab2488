"""Deterministic synthetic scenes and ray sets for the BASELINE.json configs.

Scenes: a closed axis-aligned room [-1000,1000]^3 (Source-unit scale) whose six walls are
tessellated k x k x 2 triangles, filled with M icospheres.  Rays: pinhole primary rays,
uniform-sphere rays, cosine-hemisphere bounce rays (origin offset = vistrace.CalcRayOrigin,
source/VisTrace.cpp:1495-1517; direction mapping = hemisphere_cos,
source/libraries/BSDF.cpp:69-77) and point-light shadow rays.  RNG: splitmix64, counter
based, seed 0x5EED + config id.  Everything here is numpy on the host; nothing is traced.
"""
from __future__ import annotations

import numpy as np

from ._lib import HIT_ATTRS, RAY

SEED = 0x5EED
ROOM = 1000.0
FLT_MAX = np.float32(np.finfo(np.float32).max)

#            spheres, subdivisions, wall k
SCENES = {
    "S1k": (3, 2, 4),          # 3*320 + 192      = 1 152     (fixtures / fast tests)
    "S10k": (7, 3, 9),         # 8 960 + 972      = 9 932
    "S100k": (77, 3, 11),      # 98 560 + 1 452   = 100 012
    "S1M": (770, 3, 35),       # 985 600 + 14 700 = 1 000 300
    "S10M": (1950, 4, 52),     # 9 984 000 + 32 448 = 10 016 448
}
SCENE_IDS = {"S1k": 0, "S10k": 1, "S100k": 2, "S1M": 3, "S10M": 5, "HALL100k": 7, "HALL1M": 8}
# "brush hall" scenes (make_hall): Source-map-like geometry -- few huge one-piece brush faces (hall shell, partition walls,
# pillars: 2 triangles per face, source/objects/AccelStruct.cpp:408-410) crossing thousands of small prop triangles
#            props, subdivisions of a prop, partitions, pillars
HALLS = {
    "HALL100k": (306, 2, 10, 24),    # 306*320 + 12 + 10*12 + 24*12   =  98 340
    "HALL1M": (780, 3, 14, 40),      # 780*1280 + 12 + 14*12 + 40*12  = 999 060
}


# ---- RNG ------------------------------------------------------------------------------------
def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Outputs number start .. start+count-1 (0-based) of the splitmix64 stream `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, start: int, count: int) -> np.ndarray:
    """float32 in [0,1): top 24 bits of the splitmix64 outputs."""
    return ((splitmix64(seed, start, count) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


# ---- geometry ---------------------------------------------------------------------------------
def icosphere(subdiv: int) -> np.ndarray:
    """Unit icosphere as (20*4^subdiv, 3, 3) float64 triangles, outward winding."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    tris = v[f]
    for _ in range(subdiv):
        a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
        ab, bc, ca = a + b, b + c, c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tris = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                               np.stack([ab, bc, ca], 1)], 0)
    return tris


def room_walls(k: int, half: float = ROOM) -> np.ndarray:
    """Six walls of the cube [-half,half]^3, each k x k quads x 2 triangles: (12k^2,3,3)."""
    g = np.linspace(-half, half, k + 1)
    a0, b0 = np.meshgrid(g[:-1], g[:-1], indexing="ij")
    a1, b1 = np.meshgrid(g[1:], g[1:], indexing="ij")
    a0, b0, a1, b1 = a0.ravel(), b0.ravel(), a1.ravel(), b1.ravel()
    out = []
    for axis in range(3):
        for side in (-half, half):
            def P(a, b):
                p = np.empty((a.size, 3))
                p[:, axis] = side
                p[:, (axis + 1) % 3] = a
                p[:, (axis + 2) % 3] = b
                return p
            q00, q10, q11, q01 = P(a0, b0), P(a1, b0), P(a1, b1), P(a0, b1)
            out.append(np.stack([q00, q10, q11], 1))
            out.append(np.stack([q00, q11, q01], 1))
    return np.concatenate(out, 0)


def camera_positions(name: str, count: int = 128) -> np.ndarray:
    """Camera 0 is the room centre; the others are seeded positions in [-800,800]^3."""
    u = uniform01(SEED + 100 + SCENE_IDS[name], 0, 3 * count).reshape(count, 3)
    pos = (u * 1600.0 - 800.0).astype(np.float32)
    pos[0] = 0.0
    return pos


def camera_pose(name: str, t: int, count: int = 128):
    """(position, forward) of seeded camera pose t: BASELINE config 5 renders one 1024x1024 tile per pose.
    Pose 0 looks along +x from the room centre; pose t looks from camera t towards camera t+1."""
    cams = camera_positions(name, count)
    pos = cams[t % count]
    fwd = cams[(t + 1) % count] - pos if t % count else np.array([1.0, 0.0, 0.0], np.float32)
    if np.linalg.norm(fwd) == 0:
        fwd = np.array([1.0, 0.0, 0.0], np.float32)
    return pos, fwd.astype(np.float32)


def light_positions(name: str, count: int = 16) -> np.ndarray:
    u = uniform01(SEED + 200 + SCENE_IDS[name], 0, 3 * count).reshape(count, 3)
    return (u * 1600.0 - 800.0).astype(np.float32)


def box_tris(lo, hi) -> np.ndarray:
    """The 12 triangles of the axis-aligned box [lo, hi]: (12,3,3)."""
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    out = []
    for axis in range(3):
        u, v = (axis + 1) % 3, (axis + 2) % 3
        for side in (lo[axis], hi[axis]):
            def P(a, b):
                p = np.empty(3); p[axis] = side; p[u] = a; p[v] = b
                return p
            q00, q10, q11, q01 = P(lo[u], lo[v]), P(hi[u], lo[v]), P(hi[u], hi[v]), P(lo[u], hi[v])
            out += [np.stack([q00, q10, q11]), np.stack([q00, q11, q01])]
    return np.stack(out)


def make_hall(name: str) -> np.ndarray:
    """A brush hall: the shell of the room [-1000,1000]^2 x [-300,300] as 12 huge triangles, thin partition walls
    (boxes spanning most of the hall) and square pillars from floor to ceiling -- all of them one quad per face, as a map
    compiler leaves brush faces -- and `props` icospheres of radius 6-30 resting on the floor or stacked in clusters
    around seeded spots.  Large triangles whose boxes cover thousands of small ones are what a real Source map gives
    the builder (the seeded S* scenes are uniform small triangles); nothing is enclosed around camera 0 (0,0,0)."""
    props, subdiv, nparts, npillars = HALLS[name]
    seed = SEED + SCENE_IDS[name]
    zlo, zhi = -300.0, 300.0
    parts = [box_tris((-ROOM, -ROOM, zlo), (ROOM, ROOM, zhi))]
    u = uniform01(seed, 0, 8 * (nparts + npillars + props) + 64).astype(np.float64)
    q = 0
    for i in range(nparts):                                   # thin walls along x or y with a doorway-sized gap at one end
        a, b, c, d = u[q:q + 4]; q += 4
        pos = (a * 2.0 - 1.0) * 850.0
        if abs(pos) < 120.0:
            pos = 120.0 if pos >= 0 else -120.0                # keep the camera spot clear
        span_lo, span_hi = -ROOM + 150.0 * b, ROOM - 400.0 * c - 100.0
        height = zlo + 250.0 + 350.0 * d
        if i % 2 == 0:
            parts.append(box_tris((pos - 4.0, span_lo, zlo), (pos + 4.0, span_hi, height)))
        else:
            parts.append(box_tris((span_lo, pos - 4.0, zlo), (span_hi, pos + 4.0, height)))
    for i in range(npillars):
        a, b, c = u[q:q + 3]; q += 3
        x, y, w = (a * 2.0 - 1.0) * 900.0, (b * 2.0 - 1.0) * 900.0, 12.0 + 20.0 * c
        if x * x + y * y < 150.0 ** 2:
            x += 300.0
        parts.append(box_tris((x - w, y - w, zlo), (x + w, y + w, zhi)))
    unit = icosphere(subdiv)
    nspots = max(8, props // 12)                               # props cluster around spots (furniture, debris piles)
    spots = (u[q:q + 2 * nspots].reshape(nspots, 2) * 2.0 - 1.0) * 880.0; q += 2 * nspots
    pu = uniform01(seed + 77, 0, 5 * props).reshape(props, 5).astype(np.float64)
    spot = (pu[:, 0] * nspots).astype(np.int64) % nspots
    radii = 6.0 + 24.0 * pu[:, 1]
    cx = spots[spot, 0] + (pu[:, 2] - 0.5) * 160.0
    cy = spots[spot, 1] + (pu[:, 3] - 0.5) * 160.0
    cz = zlo + radii + 120.0 * pu[:, 4] ** 3                   # most rest on the floor, a few are stacked
    near = cx * cx + cy * cy < 150.0 ** 2
    cx = np.where(near, cx + 320.0, cx)
    centres = np.stack([np.clip(cx, -ROOM + 40, ROOM - 40), np.clip(cy, -ROOM + 40, ROOM - 40), cz], 1)
    spheres = (unit[None] * radii[:, None, None, None] + centres[:, None, None, :]).reshape(-1, 3, 3)
    return np.concatenate(parts + [spheres], 0).astype(np.float32)


def make_scene(name: str) -> np.ndarray:
    """(n,3,3) float32 triangles of scene `name` (all two-sided, one material, flags 0)."""
    if name in HALLS:
        return make_hall(name)
    m, subdiv, k = SCENES[name]
    seed = SEED + SCENE_IDS[name]
    keep_out = np.concatenate([camera_positions(name), light_positions(name)], 0).astype(np.float64)
    unit = icosphere(subdiv)
    centres = np.empty((m, 3))
    radii = np.empty(m)
    draw = 0
    for i in range(m):
        while True:  # re-draw until no camera/light is within 150 units of the centre
            u = uniform01(seed, 4 * draw, 4).astype(np.float64)
            draw += 1
            r = 20.0 + 60.0 * u[3]
            c = (u[:3] * 2.0 - 1.0) * (ROOM - r)
            if np.min(np.linalg.norm(keep_out - c, axis=1)) >= 150.0:
                break
        centres[i], radii[i] = c, r
    spheres = (unit[None] * radii[:, None, None, None] + centres[:, None, None, :]).reshape(-1, 3, 3)
    return np.concatenate([spheres, room_walls(k)], 0).astype(np.float32)


def make_terrain(k: int = 24, seed: int = SEED + 50, half: float = 100.0):
    """2.5-D heightfield, every triangle oneSided (VT_TRI_CULL_BACKFACE): (2k^2,3,3), flags."""
    h = (uniform01(seed, 0, (k + 1) * (k + 1)).reshape(k + 1, k + 1) * 20.0).astype(np.float64)
    g = np.linspace(-half, half, k + 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    P = np.stack([X, Y, h], -1)
    q00, q10, q11, q01 = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t0 = np.stack([q00, q10, q11], -2).reshape(-1, 3, 3)
    t1 = np.stack([q00, q11, q01], -2).reshape(-1, 3, 3)
    verts = np.concatenate([t0, t1], 0).astype(np.float32)
    return verts, np.ones(len(verts), dtype=np.uint8)


# ---- rays -------------------------------------------------------------------------------------
def _pack(org, dirs, tmin=0.0, tmax=FLT_MAX) -> np.ndarray:
    rays = np.zeros(len(dirs), dtype=RAY)
    rays["org"] = org
    rays["dir"] = dirs
    rays["tmin"] = tmin
    rays["tmax"] = tmax
    return rays


def primary_rays(width: int, height: int, pos=(0.0, 0.0, 0.0), forward=(1.0, 0.0, 0.0), up=(0.0, 0.0, 1.0),
                 vfov_deg: float = 60.0) -> np.ndarray:
    """Pinhole camera, pixel-centre rays, row-major, normalised directions, [0, FLT_MAX]."""
    f = np.asarray(forward, np.float64); f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64)); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    th = np.tan(np.radians(vfov_deg) / 2.0)
    px = ((np.arange(width) + 0.5) / width * 2.0 - 1.0) * th * (width / height)
    py = (1.0 - (np.arange(height) + 0.5) / height * 2.0) * th
    d = f[None, None, :] + px[None, :, None] * r[None, None, :] + py[:, None, None] * u[None, None, :]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return _pack(np.asarray(pos, np.float32), d.reshape(-1, 3).astype(np.float32))


def sphere_rays(n: int, seed: int, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    """Uniform directions on the sphere from one origin."""
    u = uniform01(seed, 0, 2 * n).astype(np.float64).reshape(n, 2)
    z = 1.0 - 2.0 * u[:, 0]
    s = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    phi = 2.0 * np.pi * u[:, 1]
    d = np.stack([s * np.cos(phi), s * np.sin(phi), z], 1)
    return _pack(np.asarray(origin, np.float32), d.astype(np.float32))


def calc_ray_origin(pos: np.ndarray, normal: np.ndarray) -> np.ndarray:
    """vistrace.CalcRayOrigin, source/VisTrace.cpp:1495-1517, vectorised (float32 in/out)."""
    pos = np.ascontiguousarray(pos, np.float32)
    normal = np.ascontiguousarray(normal, np.float32)
    origin, f_scale, i_scale = np.float32(1 / 32), np.float32(1 / 65536), np.float32(256)
    i_off = (normal * i_scale).astype(np.int32)            # ivec3(normal * iScale): truncation
    bits = pos.view(np.int32) + np.where(pos < 0, -i_off, i_off)
    i_pos = bits.view(np.float32)
    f_off = normal * f_scale
    return np.where(np.abs(pos) < origin, pos + f_off, i_pos).astype(np.float32)


def hemisphere_cos(r1: np.ndarray, r2: np.ndarray) -> np.ndarray:
    """hemisphere_cos, source/libraries/BSDF.cpp:69-77 (local frame, z = normal)."""
    r1 = r1.astype(np.float32); r2 = r2.astype(np.float32)
    z = np.sqrt(r1)
    sin_t = np.sqrt(np.float32(1) - r1)
    phi = np.float32(2) * np.float32(np.pi) * r2
    return np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), z], -1).astype(np.float32)


def _onb(n: np.ndarray):
    """Branch-light orthonormal basis around unit normals (Duff et al. 2017)."""
    n = n.astype(np.float32)
    sign = np.where(n[:, 2] >= 0, np.float32(1), np.float32(-1))
    a = np.float32(-1) / (sign + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    b1 = np.stack([np.float32(1) + sign * n[:, 0] * n[:, 0] * a, sign * b, -sign * n[:, 0]], 1)
    b2 = np.stack([b, sign + n[:, 1] * n[:, 1] * a, -n[:, 1]], 1)
    return b1.astype(np.float32), b2.astype(np.float32)


def _facing_normals(attrs: np.ndarray) -> np.ndarray:
    """Geometric normal flipped towards wo (the side the ray arrived from)."""
    ng = attrs["ngeo"].astype(np.float32)
    return np.where((attrs["front"] != 0)[:, None], ng, -ng).astype(np.float32)


def fill_misses(attrs: np.ndarray) -> np.ndarray:
    """Replace missed records by the previous hit in ray order (keeps N exact)."""
    assert attrs.dtype == HIT_ATTRS
    hit = attrs["hit"] != 0
    if hit.all():
        return attrs
    idx = np.where(hit, np.arange(len(attrs)), -1)
    idx = np.maximum.accumulate(idx)
    first = int(np.argmax(hit))
    idx[idx < 0] = first
    return attrs[idx]


def bounce_rays(attrs: np.ndarray, seed: int) -> np.ndarray:
    """One cosine-hemisphere bounce ray per primary hit record, order left as generated."""
    attrs = fill_misses(attrs)
    n = len(attrs)
    nrm = _facing_normals(attrs)
    org = calc_ray_origin(attrs["pos"], nrm)
    u = uniform01(seed, 0, 2 * n).reshape(n, 2)
    loc = hemisphere_cos(u[:, 0], u[:, 1])
    b1, b2 = _onb(nrm)
    d = b1 * loc[:, 0:1] + b2 * loc[:, 1:2] + nrm * loc[:, 2:3]
    return _pack(org, d.astype(np.float32))


def shadow_rays(attrs: np.ndarray, lights: np.ndarray, seed: int, per_hit: int = 4) -> np.ndarray:
    """per_hit shadow rays per hit record towards seeded point lights; tmax = dist*(1-1e-4)."""
    attrs = fill_misses(attrs)
    n = len(attrs)
    nrm = _facing_normals(attrs)
    org = np.repeat(calc_ray_origin(attrs["pos"], nrm), per_hit, axis=0)
    pick = (splitmix64(seed, 0, n * per_hit) % np.uint64(len(lights))).astype(np.int64)
    to = lights[pick].astype(np.float32) - org
    dist = np.sqrt((to * to).sum(1, dtype=np.float32)).astype(np.float32)
    d = (to / dist[:, None]).astype(np.float32)
    return _pack(org, d, 0.0, (dist * np.float32(1.0 - 1e-4)).astype(np.float32))


def skinned_rig(ntris: int, nents: int = 8, bones_per_ent: int = 12, seed: int = SEED + 9):
    """Synthetic skinning inputs for ntris triangles (seeded): triangles are dealt to `nents` entities in
    contiguous runs, every vertex gets 1-3 bones of its entity with weights summing to ~1.
    Returns (skin (ntris,3) SKIN_VERTEX-compatible structured array, matrix_base (ntris,) u32, nmat)."""
    rng = np.random.default_rng(seed)
    skin = np.zeros((ntris, 3), np.dtype([("weight", "<f4", 3), ("bone", "i1", 3), ("num_bones", "u1")]))
    ent = (np.arange(ntris, dtype=np.uint64) * nents // max(ntris, 1)).astype(np.uint32)
    nb = rng.integers(1, 4, size=(ntris, 3))
    w = rng.random((ntris, 3, 3)).astype(np.float32) + np.float32(0.05)
    w *= (np.arange(3)[None, None, :] < nb[:, :, None])
    w = (w / w.sum(axis=2, keepdims=True)).astype(np.float32)
    skin["weight"] = w
    skin["bone"] = rng.integers(0, bones_per_ent, size=(ntris, 3, 3)).astype(np.int8)
    skin["num_bones"] = nb.astype(np.uint8)
    return skin, (ent * np.uint32(bones_per_ent)).astype(np.uint32), nents * bones_per_ent


def rig_pose(nmat: int, frame: int, seed: int = SEED + 10):
    """(bones, binds) as (nmat,16) column-major float32: binds are fixed seeded affine matrices, bones are
    small frame-dependent rotations + translations composed with the inverse bind (so the pose stays near
    the bind pose, as an animated model does)."""
    rng = np.random.default_rng(seed)
    def affine(r, ang, shift):
        axis = r.normal(size=(nmat, 3)); axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        a = r.normal(scale=ang, size=nmat)
        K = np.zeros((nmat, 3, 3)); K[:, 0, 1] = -axis[:, 2]; K[:, 0, 2] = axis[:, 1]; K[:, 1, 0] = axis[:, 2]
        K[:, 1, 2] = -axis[:, 0]; K[:, 2, 0] = -axis[:, 1]; K[:, 2, 1] = axis[:, 0]
        R = np.eye(3)[None] + np.sin(a)[:, None, None] * K + (1 - np.cos(a))[:, None, None] * (K @ K)
        M = np.tile(np.eye(4), (nmat, 1, 1)); M[:, :3, :3] = R; M[:, :3, 3] = r.normal(scale=shift, size=(nmat, 3))
        return M
    bind = affine(rng, 0.6, 40.0)
    move = affine(np.random.default_rng(seed + 1 + frame), 0.08, 6.0)
    bone = move @ np.linalg.inv(bind)
    cm = lambda M: np.ascontiguousarray(np.transpose(M, (0, 2, 1)).reshape(nmat, 16), np.float32)   # column-major
    return cm(bone), cm(bind)


def vertex_frames(verts: np.ndarray, seed: int = SEED + 12, bend: float = 0.35):
    """Seeded per-vertex shading frames for triangles (n,3,3): normals = the face normal bent by up to ~`bend` rad per vertex
    (as a smoothed mesh has them), tangents = a seeded direction made perpendicular to the vertex normal; unit length in fp32
    up to rounding, a few triangles with un-normalised vectors (the reference takes the mesh's as they come).
    Returns an (n,) structured array compatible with TRI_FRAME."""
    rng = np.random.default_rng(seed)
    n = len(verts)
    v = np.asarray(verts, np.float64).reshape(n, 3, 3)
    face = np.cross(v[:, 0] - v[:, 1], v[:, 2] - v[:, 0])                       # Primitives.h:95 cross(e1, e2)
    ln = np.linalg.norm(face, axis=1, keepdims=True)
    face = np.where(ln > 0, face / np.where(ln > 0, ln, 1), np.array([0.0, 0.0, 1.0]))
    nrm = face[:, None, :] + rng.normal(scale=bend, size=(n, 3, 3))
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
    tan = rng.normal(size=(n, 3, 3))
    tan -= nrm * (tan * nrm).sum(axis=2, keepdims=True)
    tan /= np.linalg.norm(tan, axis=2, keepdims=True)
    scale = np.where(rng.random((n, 1, 1)) < 0.02, rng.uniform(0.25, 4.0, (n, 1, 1)), 1.0)
    out = np.zeros(n, np.dtype([("normal", "<f4", (3, 3)), ("tangent", "<f4", (3, 3))]))
    out["normal"] = (nrm * scale).astype(np.float32)
    out["tangent"] = (tan * scale).astype(np.float32)
    return out


def alpha_test_rig(ntris: int, nmats: int = 6, alpha_fraction: float = 0.4, seed: int = SEED + 11):
    """Seeded alpha-test inputs: (flags u8 with bit 1 = VT_TRI_ALPHATEST on ~alpha_fraction of the triangles,
    TRI_ATTRIBS-compatible array with uvs + material, ALPHA_MATERIAL-compatible array, texels u8).
    Materials mix nearest / bilinear, odd plane sizes, a 1x1 plane and one material without a texture."""
    rng = np.random.default_rng(seed)
    flags = (rng.random(ntris) < alpha_fraction).astype(np.uint8) * np.uint8(2)
    attribs = np.zeros(ntris, np.dtype([("uv", "<f4", (3, 2)), ("alpha", "<f4", 3), ("ent_id", "<u4"), ("material", "<u4"), ("pad", "<u4")]))
    attribs["uv"] = rng.uniform(-2.0, 3.0, (ntris, 3, 2)).astype(np.float32)
    attribs["alpha"] = 1.0
    attribs["material"] = rng.integers(0, nmats, ntris).astype(np.uint32)
    mats = np.zeros(nmats, np.dtype([("tex_mat", "<f4", (2, 4)), ("tex_scale", "<f4"), ("alpha_ref", "<f4"), ("width", "<u4"),
                                     ("height", "<u4"), ("filter", "<u4"), ("pad", "<u4"), ("offset", "<u8")]))
    sizes = [(8, 8), (16, 5), (1, 1), (0, 0), (7, 13), (32, 32), (3, 2), (64, 16)]
    planes, off = [], 0
    for i in range(nmats):
        w, h = sizes[i % len(sizes)]
        mats["tex_mat"][i] = [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]]
        mats["tex_mat"][i] += rng.normal(scale=0.2, size=(2, 4)).astype(np.float32)
        mats["tex_scale"][i] = np.float32(rng.choice([1.0, 0.5, 2.0, 4.0]))
        mats["alpha_ref"][i] = np.float32(rng.choice([0.5, 0.5, 0.25, 0.9]))
        mats["width"][i], mats["height"][i] = w, h
        mats["filter"][i] = i % 2
        mats["offset"][i] = off
        plane = np.where(rng.random(w * h) < 0.5, 255, rng.integers(0, 256, w * h)).astype(np.uint8)
        planes.append(plane)
        off += w * h
    texels = np.concatenate(planes) if off else np.zeros(0, np.uint8)
    return flags, attribs, mats, texels


def random_soup(seed: int):
    """Seeded random triangle soup + rays (tests/test_gpu_parity.py::test_random_triangle_soups, scripts/recall_sensitivity.py):
    mixed scales, degenerate (zero-area / collinear) and exactly duplicated triangles, random cull flags, rays from inside and
    outside with random [tmin, tmax] windows and some zero direction components.
    Returns (verts (n,3,3) f32, flags (n,) u8, org (m,3), dir (m,3), tmin (m,), tmax (m,))."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 3000))
    centre = rng.uniform(-50, 50, (n, 1, 3))
    scale = 10.0 ** rng.uniform(-3, 1.5, (n, 1, 1))
    verts = (centre + rng.normal(size=(n, 3, 3)) * scale).astype(np.float32)
    verts[::17, 1] = verts[::17, 0]                                  # zero-area: two equal vertices
    verts[5::23, 2] = (verts[5::23, 0] + verts[5::23, 1]) / 2        # collinear
    verts[3::29] = verts[2::29][: len(verts[3::29])]                 # exact duplicates (ties)
    flags = (rng.random(n) < 0.4).astype(np.uint8)
    m = 6000
    org = rng.uniform(-80, 80, (m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d[: m // 4] = (centre[rng.integers(0, n, m // 4), 0] - org[: m // 4]).astype(np.float32)   # aimed at geometry
    d[::50, rng.integers(0, 3)] = 0.0
    tmin = np.where(rng.random(m) < 0.3, rng.uniform(0, 20, m), 0.0).astype(np.float32)
    tmax = np.where(rng.random(m) < 0.3, tmin + rng.uniform(0.1, 100, m), np.finfo(np.float32).max).astype(np.float32)
    return verts, flags, org, d, tmin, tmax
