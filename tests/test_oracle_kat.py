"""Known-answer tests that pin the CPU oracle (oracle/vt_oracle.c).

The reference holds no tests or vectors for this path, so the oracle is
pinned analytically: every expected value below is derived by hand from the arithmetic of
source/objects/Primitives.h:168-215 (triangle test), SURVEY.md 3.2 (bvh v1 walk),
source/VisTrace.cpp:1495-1517 (CalcRayOrigin) and source/libraries/BSDF.cpp:69-77.
"""
import numpy as np
import pytest

FLT_MAX = np.finfo(np.float32).max


def ray(O, org, d, tmin=0.0, tmax=FLT_MAX):
    r = np.zeros(1, O.RAY)
    r["org"], r["dir"], r["tmin"], r["tmax"] = org, d, tmin, tmax
    return r


# triangle in the plane z = 0: p0=(0,0,0) p1=(1,0,0) p2=(0,1,0)
# e1 = p0-p1 = (-1,0,0), e2 = p2-p0 = (0,1,0), n = cross(e1,e2) = (0*0-0*1, 0*0-(-1)*0, -1*1-0*0) = (0,0,-1)
TRI = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)


def test_tri_setup_fields(O):
    t = O.tris_setup(TRI)
    assert t["p0"][0].tolist() == [0, 0, 0]
    assert t["e1"][0].tolist() == [-1, 0, 0]
    assert t["e2"][0].tolist() == [0, 1, 0]
    assert t["n"][0].tolist() == [0, 0, -1]


def test_centre_hit_and_barycentrics(O):
    # ray from (0.25,0.5,1) along -z: nDotDir = 1, c = p0-org = (-.25,-.5,-1), r = cross(dir,c) = (-0.5,0.25,0)
    # u = dot(r,e2) = 0.25 (weights vertex 1), v = dot(r,e1) = 0.5 (vertex 2), t = dot(n,c) = 1
    t = O.tris_setup(TRI)
    h = O.trace_brute(t, ray(O, [0.25, 0.5, 1], [0, 0, -1]))
    assert h["prim"][0] == 0 and h["t"][0] == 1.0 and h["u"][0] == 0.25 and h["v"][0] == 0.5


@pytest.mark.parametrize("x,y,hit", [
    (0.0, 0.0, True), (1.0, 0.0, True), (0.0, 1.0, True),      # the three vertices (u,v,w == 0 accepted: >= 0)
    (0.5, 0.0, True), (0.0, 0.5, True), (0.5, 0.5, True),      # edge midpoints
    (-0.001, 0.5, False), (0.5, -0.001, False), (0.51, 0.51, False),
])
def test_edges_vertices_no_epsilon(O, x, y, hit):
    t = O.tris_setup(TRI)
    h = O.trace_brute(t, ray(O, [x, y, 1], [0, 0, -1]))
    assert (h["prim"][0] != O.MISS) == hit


def test_backface_cull_flag(O):
    # from below (dir +z): nDotDir = dot((0,0,-1),(0,0,1)) = -1 -> not > 0 -> kept even when one-sided;
    # from above (dir -z): nDotDir = +1 > 0 -> culled only with the flag (oneSided && !nocull)
    two_sided = O.tris_setup(TRI)
    one_sided = O.tris_setup(TRI, flags=[1])
    above, below = ray(O, [0.2, 0.2, 1], [0, 0, -1]), ray(O, [0.2, 0.2, -1], [0, 0, 1])
    assert O.trace_brute(two_sided, above)["prim"][0] == 0
    assert O.trace_brute(two_sided, below)["prim"][0] == 0
    assert O.trace_brute(one_sided, above)["prim"][0] == O.MISS
    assert O.trace_brute(one_sided, below)["prim"][0] == 0


def test_parallel_ray_is_nan_miss(O):
    t = O.tris_setup(TRI)
    assert O.trace_brute(t, ray(O, [0.2, 0.2, 0.0], [1, 0, 0]))["prim"][0] == O.MISS  # in-plane: 0/0
    assert O.trace_brute(t, ray(O, [0.2, 0.2, 1.0], [1, 0, 0]))["prim"][0] == O.MISS  # parallel above: x/0


@pytest.mark.parametrize("tmin,tmax,hit", [(0, 1.0, True), (1.0, 2.0, True), (0, 0.999, False), (1.001, 5, False)])
def test_range_is_closed(O, tmin, tmax, hit):
    t = O.tris_setup(TRI)
    h = O.trace_brute(t, ray(O, [0.25, 0.25, 1], [0, 0, -1], tmin, tmax))
    assert (h["prim"][0] != O.MISS) == hit


def test_unnormalised_direction_scales_t(O):
    t = O.tris_setup(TRI)
    h = O.trace_brute(t, ray(O, [0.25, 0.25, 1], [0, 0, -4]))
    assert h["t"][0] == 0.25


def test_coincident_triangles_later_wins(O):
    # two identical triangles: equal t; accept rule is t <= tmax, so the later-tested one replaces
    verts = np.concatenate([TRI, TRI])
    t = O.tris_setup(verts)
    h = O.trace_brute(t, ray(O, [0.25, 0.25, 1], [0, 0, -1]))
    assert h["prim"][0] == 1
    tmin_hit, ids, n = O.min_t_set(t, ray(O, [0.25, 0.25, 1], [0, 0, -1]))
    assert n == 2 and sorted(ids.tolist()) == [0, 1] and tmin_hit == 1.0
    # any-hit stops at the first accepted one
    assert O.trace_brute(t, ray(O, [0.25, 0.25, 1], [0, 0, -1]), any_hit=True)["prim"][0] == 0


def _two_leaf_tree(O, verts):
    """Hand-built v1 tree: root + two single-triangle leaves (left = tri 0, right = tri 1)."""
    tris = O.tris_setup(verts)
    nodes = np.zeros(3, O.NODE)

    def bounds(v):
        lo, hi = v.min(0), v.max(0)
        return [lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]]
    nodes["bounds"][0] = bounds(verts.reshape(-1, 3))
    nodes["prim_count"][0], nodes["first"][0] = 0, 1
    nodes["bounds"][1], nodes["prim_count"][1], nodes["first"][1] = bounds(verts[0]), 1, 0
    nodes["bounds"][2], nodes["prim_count"][2], nodes["first"][2] = bounds(verts[1]), 1, 1
    return nodes, np.array([0, 1], np.uint32), tris


def test_traverse_two_leaves_order_and_counters(O):
    # left leaf at z=0, right leaf at z=-1 (behind it along -z): both slabs pass with the
    # initial tmax (both tested before either leaf), left is intersected first (t=1), the right
    # leaf is still intersected (its slab test used the old tmax) but t=2 > tmax=1 is rejected.
    far = TRI.copy(); far[:, :, 2] = -1
    nodes, pidx, tris = _two_leaf_tree(O, np.concatenate([TRI, far]))
    hits, st, steps, tests, _ = O.traverse_batch(nodes, pidx, tris, ray(O, [0.25, 0.25, 1], [0, 0, -1]), want_stats=True)
    assert hits["prim"][0] == 0 and hits["t"][0] == 1.0
    assert st[0].tolist() == [1, 2]          # one loop iteration, two triangle tests
    # coincident leaves: equal t, the right (later visited) leaf wins the tie
    nodes, pidx, tris = _two_leaf_tree(O, np.concatenate([TRI, TRI]))
    hits, _, _, _, _ = O.traverse_batch(nodes, pidx, tris, ray(O, [0.25, 0.25, 1], [0, 0, -1]))
    assert hits["prim"][0] == 1
    hits, _, _, _, _ = O.traverse_batch(nodes, pidx, tris, ray(O, [0.25, 0.25, 1], [0, 0, -1]), any_hit=True)
    assert hits["prim"][0] == 0


def test_root_leaf_has_no_slab_test(O):
    # a root that is a leaf is intersected directly: even a ray whose slab test would fail
    # (origin outside, pointing away in x) reaches the triangle test; counters: 0 steps
    tris = O.tris_setup(np.concatenate([TRI, TRI + 5]))
    nodes = np.zeros(1, O.NODE)
    nodes["bounds"][0] = [0, 1, 0, 1, 0, 0]      # deliberately too small for triangle 1
    nodes["prim_count"][0], nodes["first"][0] = 2, 0
    r = ray(O, [5.25, 5.25, 6], [0, 0, -1])
    hits, st, _, _, _ = O.traverse_batch(nodes, np.array([0, 1], np.uint32), tris, r, want_stats=True)
    assert hits["prim"][0] == 1 and hits["t"][0] == 1.0 and st[0].tolist() == [0, 2]


def test_safe_inverse_axis_parallel_rays(O):
    # dir components of exactly 0 / -0 are clamped to +-1/FLT_EPSILON, so axis-parallel rays walk a
    # grid of boxes without inf*0 NaNs: ray along +x at y=z=0.25 through two triangles facing x
    a = np.array([[[1, 0, 0], [1, 1, 0], [1, 0, 1]]], np.float32)
    b = a.copy(); b[:, :, 0] = 2
    nodes, pidx, tris = _two_leaf_tree(O, np.concatenate([a, b]))
    for d in ([1, 0.0, 0.0], [1, -0.0, -0.0], [1, 1e-8, -1e-8]):
        hits, _, _, _, _ = O.traverse_batch(nodes, pidx, tris, ray(O, [0, 0.25, 0.25], d))
        assert hits["prim"][0] == 0 and abs(hits["t"][0] - 1.0) < 1e-6
    # from the other side the far triangle (index 1) is the closest
    hits, _, _, _, _ = O.traverse_batch(nodes, pidx, tris, ray(O, [3, 0.25, 0.25], [-1, 0.0, -0.0]))
    assert hits["prim"][0] == 1 and hits["t"][0] == 1.0


def test_nan_range_misses_after_one_step(O):
    far = TRI.copy(); far[:, :, 2] = -1
    nodes, pidx, tris = _two_leaf_tree(O, np.concatenate([TRI, far]))
    for tmin, tmax in ((np.nan, FLT_MAX), (0.0, np.nan)):
        hits, st, _, _, _ = O.traverse_batch(nodes, pidx, tris, ray(O, [0.25, 0.25, 1], [0, 0, -1], tmin, tmax), want_stats=True)
        assert hits["prim"][0] == O.MISS and st[0].tolist() == [1, 0]


def test_hit_attrs(O):
    # TraceResult.cpp: uvw = (u, v, 1-u-v); pos = w*v0 + u*v1 + v*v2; v1 = p0 - e1, v2 = p0 + e2
    t = O.tris_setup(TRI)
    r = ray(O, [0.25, 0.5, 1], [0, 0, -2])
    h = O.trace_brute(t, r)
    a = O.hit_attrs(t, r, h)[0]
    assert a["uvw"].tolist() == [0.25, 0.5, 0.25]
    assert a["pos"].tolist() == [0.25, 0.5, 0.0]
    assert a["ngeo"].tolist() == [0, 0, -1]            # n / |n|
    assert a["wo"].tolist() == [0, 0, 1]               # -normalize(dir)
    assert a["front"] == 0                             # dot(wo, ngeo) = -1 < 0


def test_calc_ray_origin_kat(O):
    # VisTrace.cpp:1495-1517. |pos| >= 1/32: integer offset iOff = int(n*256) on the float bits,
    # sign flipped for negative pos; |pos| < 1/32: pos + n/65536.
    pos = np.array([100.0, -100.0, 0.01], np.float32)
    nrm = np.array([1.0, 1.0, -1.0], np.float32)
    out = O.calc_ray_origin(pos, nrm)
    bits = np.array([100.0, -100.0], np.float32).view(np.int32)
    exp0 = np.array([bits[0] + 256], np.int32).view(np.float32)[0]
    exp1 = np.array([bits[1] - 256], np.int32).view(np.float32)[0]   # pos<0: -iOff; moves towards +y (less negative)
    assert out[0] == exp0 and out[1] == exp1
    assert out[1] > -100.0 and out[0] > 100.0
    assert out[2] == np.float32(0.01) + np.float32(-1.0) * np.float32(1 / 65536)


def test_hemisphere_cos_kat(O):
    # BSDF.cpp:69-77: z = sqrt(r1), sinTheta = sqrt(1-r1), phi = 2*pi*r2
    assert O.hemisphere_cos(1.0, 0.0).tolist() == [0.0, 0.0, 1.0]
    v = O.hemisphere_cos(0.0, 0.0)
    assert v.tolist() == [1.0, 0.0, 0.0]
    v = O.hemisphere_cos(0.25, 0.25)      # phi = pi/2
    assert abs(v[0]) < 1e-7 and abs(v[1] - np.sqrt(np.float32(0.75))) < 1e-7 and v[2] == 0.5
    assert abs(np.linalg.norm(O.hemisphere_cos(0.3, 0.7)) - 1.0) < 1e-6


# ---- skinning on Rebuild: AccelStruct.cpp:34-102 ------------------------------------------------------------
def _cm(M):
    return np.ascontiguousarray(np.asarray(M, np.float32).T.reshape(16))


def _skin1(O, n, bone=0, weight=1.0):
    sv = np.zeros((n, 3), O.SKIN_VERTEX)
    sv["weight"][:, :, 0] = weight
    sv["bone"][:, :, 0] = bone
    sv["num_bones"] = 1
    return sv


def test_skin_identity_and_roundtrip(O):
    """Identity bone x bind: the vertices come back through the p0 - (p0 - p1) round trip of SkinTriangle :68-72."""
    rng = np.random.default_rng(3)
    v = rng.normal(scale=100, size=(50, 9)).astype(np.float32)
    I = np.eye(4, dtype=np.float32)
    mats = O.skin_matrices(_cm(I)[None], _cm(I)[None])
    assert (mats == _cm(I)).all()
    out = O.skin_verts(v, _skin1(O, 50), np.zeros(50, np.uint32), mats)
    p0, p1, p2 = v[:, 0:3], v[:, 3:6], v[:, 6:9]
    exp = np.concatenate([p0, p0 - (p0 - p1), p0 + (p2 - p0)], axis=1)
    assert (out.view(np.uint32) == exp.view(np.uint32)).all()


def test_skin_exact_cases(O):
    """Exactly representable transforms: quarter turn about z plus a shift; bone*bind that cancels; a 50/50
    blend of two opposite shifts; zero bones -> origin (final starts at 0, :42)."""
    v = np.array([[1, 2, 3, 5, 2, 3, 1, 6, 3]], np.float32)
    Rz = np.array([[0, -1, 0, 10], [1, 0, 0, 20], [0, 0, 1, 30], [0, 0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    out = O.skin_verts(v, _skin1(O, 1), np.zeros(1, np.uint32), O.skin_matrices(_cm(Rz)[None], _cm(I)[None]))
    assert out.tolist() == [[8, 21, 33, 8, 25, 33, 4, 21, 33]]
    T = np.eye(4, dtype=np.float32); T[:3, 3] = [4, -8, 16]
    Tinv = np.eye(4, dtype=np.float32); Tinv[:3, 3] = [-4, 8, -16]
    out = O.skin_verts(v, _skin1(O, 1), np.zeros(1, np.uint32), O.skin_matrices(_cm(T)[None], _cm(Tinv)[None]))
    assert out.tolist() == v.tolist()
    sv = np.zeros((1, 3), O.SKIN_VERTEX)
    sv["weight"][:, :, :2] = 0.5
    sv["bone"][:, :, 1] = 1
    sv["num_bones"] = 2
    mats = O.skin_matrices(np.stack([_cm(T), _cm(Tinv)]), np.stack([_cm(I), _cm(I)]))
    assert O.skin_verts(v, sv, np.zeros(1, np.uint32), mats).tolist() == v.tolist()
    sv["num_bones"] = 0
    assert O.skin_verts(v, sv, np.zeros(1, np.uint32), mats).tolist() == [[0] * 9]
    # matrix_base selects the entity's block of matrices
    mats2 = np.concatenate([mats, O.skin_matrices(_cm(Rz)[None], _cm(I)[None])])
    out = O.skin_verts(v, _skin1(O, 1), np.array([2], np.uint32), mats2)
    assert out.tolist() == [[8, 21, 33, 8, 25, 33, 4, 21, 33]]


def test_skin_matches_float64_model(O):
    from vistrace_amd import workloads as W
    rng = np.random.default_rng(5)
    n = 300
    v = rng.normal(scale=60, size=(n, 9)).astype(np.float32)
    skin, base, nmat = W.skinned_rig(n, nents=3, bones_per_ent=5)
    bones, binds = W.rig_pose(nmat, frame=2)
    mats = O.skin_matrices(bones, binds)
    M64 = bones.reshape(nmat, 4, 4).transpose(0, 2, 1).astype(np.float64) @ binds.reshape(nmat, 4, 4).transpose(0, 2, 1).astype(np.float64)
    assert np.allclose(mats.reshape(nmat, 4, 4).transpose(0, 2, 1), M64, rtol=1e-5, atol=1e-4)
    out = O.skin_verts(v, skin, base, mats).reshape(n, 3, 3)
    exp = np.zeros((n, 3, 3))
    for t in range(n):
        for vi in range(3):
            p = np.append(v[t, vi * 3:vi * 3 + 3].astype(np.float64), 1.0)
            for q in range(skin["num_bones"][t, vi]):
                exp[t, vi] += (M64[base[t] + skin["bone"][t, vi, q]] @ p)[:3] * skin["weight"][t, vi, q]
    assert np.allclose(out, exp, rtol=1e-4, atol=2e-3)


# ---- shading frame: TraceResult.cpp:58-62, 89-103, 132-137, 175-186; frames skinned as AccelStruct.cpp:82-92 ----------------------
def _flat_frame(normal, tangent):
    return np.array([list(normal) * 3 + list(tangent) * 3], np.float32)


def test_hit_tbn_known_answers(O):
    """Flat triangle in z = 0 (geometric normal -z: cross(e1, e2), left-handed), vertex frame n = +z, t = +x: interpolation of
    equal vectors returns them, binormal = cross(t, n) = -y; a head-on ray leaves the frame alone, a grazing one (|cos| <= 0.1)
    pulls the normal towards the GEOMETRIC normal by lerp(ngeo, n, cos * 10) and re-orthogonalises tangent and binormal."""
    t = O.tris_setup(TRI)
    uvs = np.array([[0, 0, 2, 0, 0, 2]], np.float32)               # triUVArea = 4, |n| = 1 -> lod = 0.5 * log2(4) = 1
    fr = _flat_frame([0, 0, 1], [1, 0, 0])
    r = ray(O, [0.25, 0.5, 1], [0, 0, -2])
    h = O.trace_brute(t, r)
    o = O.hit_tbn(t, r, h, fr, uvs)[0]
    assert o["normal"].tolist() == [0, 0, 1] and o["tangent"].tolist() == [1, 0, 0] and o["binormal"].tolist() == [0, -1, 0]
    assert o["lod_set"] == 0 and o["lod_info"].tolist() == [0, 0]   # cone off: accel:Traverse's defaults (mipOverride)
    # cone on: `distance` is hit->distance() (AccelStruct.cpp:826), i.e. t in units of the direction AS GIVEN (|dir| = 2: t = 0.5);
    # width 0.5 + angle 0.25 * 0.5 = 0.625; normalTerm = dot(wo, ngeo) = -1
    assert h["t"][0] == 0.5
    o = O.hit_tbn(t, r, h, fr, uvs, 0.5, 0.25)[0]
    assert o["lod_set"] == 1 and o["lod_info"].tolist() == [1.0, 0.390625]
    for cw, ca in ((1.0, 0.0), (-1.0, 0.25), (-0.5, -1.0)):        # mipOverride: coneWidth < 0 || coneAngle <= 0
        assert O.hit_tbn(t, r, h, fr, uvs, cw, ca)[0]["lod_set"] == 0
    # un-normalised vertex vectors are normalised after interpolation
    o = O.hit_tbn(t, r, h, _flat_frame([0, 0, 4], [0.5, 0, 0]), uvs)[0]
    assert o["normal"].tolist() == [0, 0, 1] and o["tangent"].tolist() == [1, 0, 0] and o["binormal"].tolist() == [0, -1, 0]
    # grazing: wo = (0.8, 0, 0.6) against a bent vertex normal n = (0.6, 0, -0.8) (dot exactly 0): s = 0 -> normal = ngeo = -z,
    # tangent = normalize(t - n * dot(t, n)) with t = +y stays +y, binormal = cross(t, n) = (-1, 0, 0)
    r = ray(O, [1.05, 0.5, 0.6], [-0.8, 0, -0.6])
    h = O.trace_brute(t, r)
    assert h["prim"][0] == 0
    o = O.hit_tbn(t, r, h, _flat_frame([0.6, 0, -0.8], [0, 1, 0]), uvs)[0]
    assert np.abs(o["normal"] - [0, 0, -1]).max() < 1e-6 and np.abs(o["tangent"] - [0, 1, 0]).max() < 1e-6
    assert np.abs(o["binormal"] - [-1, 0, 0]).max() < 1e-6
    # a miss stays zero
    miss = O.trace_brute(t, ray(O, [5, 5, 1], [0, 0, -1]))
    assert not O.hit_tbn(t, ray(O, [5, 5, 1], [0, 0, -1]), miss, fr, uvs).view(np.uint8).any()


def test_hit_tbn_matches_float64_model(O):
    """Random bent frames on a random soup, hits from the brute-force intersector: the fp32 restatement against the same formulas
    in float64 (away from the branch threshold, where the two may legitimately take different sides)."""
    from vistrace_amd import workloads as W
    rng = np.random.default_rng(8)
    n = 400
    verts = rng.normal(scale=10, size=(n, 3, 3)).astype(np.float32)
    tris = O.tris_setup(verts)
    frames = W.vertex_frames(verts, 3)
    uvs = rng.uniform(-3, 3, (n, 6)).astype(np.float32)
    org = rng.normal(scale=30, size=(3000, 3)).astype(np.float32)
    tgt = verts[rng.integers(0, n, 3000)].mean(axis=1)
    rays = np.zeros(3000, O.RAY)
    rays["org"], rays["dir"], rays["tmax"] = org, tgt - org, FLT_MAX
    hits = O.trace_brute(tris, rays)
    hit = hits["prim"] != O.MISS
    assert hit.sum() > 2500
    cw, ca = 0.3, 0.01
    got = O.hit_tbn(tris, rays, hits, frames.view(np.float32).reshape(-1, 18), uvs, cw, ca)
    at = O.hit_attrs(tris, rays, hits)
    p = np.where(hit, hits["prim"], 0)
    nz = lambda x: x / np.linalg.norm(x, axis=1, keepdims=True)
    N, T = frames["normal"].astype(np.float64), frames["tangent"].astype(np.float64)
    B = np.cross(T, N)
    uvw = at["uvw"].astype(np.float64)
    interp = lambda A: uvw[:, 2:3] * A[p, 0] + uvw[:, 0:1] * A[p, 1] + uvw[:, 1:2] * A[p, 2]
    with np.errstate(invalid="ignore"):
        nn, tt, bb = nz(interp(N)), nz(interp(T)), nz(interp(B))
        wo, ng = at["wo"].astype(np.float64), at["ngeo"].astype(np.float64)
        c = np.abs((wo * nn).sum(1))
        s = np.clip(c * 10, 0, 1)[:, None]
        n2 = nz(ng * (1 - s) + nn * s)
        t2 = nz(tt - n2 * (tt * n2).sum(1, keepdims=True))
        b2 = np.cross(t2, n2)
    g = (c <= 0.1)[:, None]
    nn, tt, bb = np.where(g, n2, nn), np.where(g, t2, tt), np.where(g, b2, bb)
    sel = hit & (np.abs(c - 0.1) > 1e-5)
    assert (c[hit] <= 0.1).sum() > 20
    for name, model in (("normal", nn), ("tangent", tt), ("binormal", bb)):
        assert np.abs(got[name][sel] - model[sel]).max() < 5e-6
    v = verts.astype(np.float64)
    ln = np.linalg.norm(np.cross(v[:, 0] - v[:, 1], v[:, 2] - v[:, 0]), axis=1)
    uv = uvs.astype(np.float64)
    area = np.abs((uv[:, 2] - uv[:, 0]) * (uv[:, 5] - uv[:, 1]) - (uv[:, 4] - uv[:, 0]) * (uv[:, 3] - uv[:, 1]))
    lod = 0.5 * np.log2(area / ln)
    assert np.allclose(got["lod_info"][hit, 0], lod[p][hit], rtol=1e-5, atol=1e-5)
    cone = (ca * hits["t"].astype(np.float64) + cw) ** 2 / (wo * ng).sum(1) ** 2
    assert np.allclose(got["lod_info"][hit, 1], cone[hit], rtol=1e-5)
    assert (got["lod_set"][hit] == 1).all()


def test_skin_frames_rotate_without_translation(O):
    """angleOnly (AccelStruct.cpp:42): the vertex is (vec, 0), so a bone's translation never reaches a normal or tangent; exact
    quarter turn; 50/50 blend of a turn and identity; zero bones -> zero vector; matrix_base selects the entity's matrices."""
    Rz = np.array([[0, -1, 0, 10], [1, 0, 0, 20], [0, 0, 1, 30], [0, 0, 0, 1]], np.float32)
    I = np.eye(4, dtype=np.float32)
    fr = np.array([[1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 1, 0, 1, 1, 0, 0, 0, -1]], np.float32)
    mats = O.skin_matrices(np.stack([_cm(Rz), _cm(I)]), np.stack([_cm(I), _cm(I)]))
    out = O.skin_frames(fr, _skin1(O, 1), np.zeros(1, np.uint32), mats)
    assert out.tolist() == [[0, 1, 0, -2, 0, 0, 0, 0, 3, -1, 0, 0, -1, 1, 0, 0, 0, -1]]
    assert O.skin_frames(fr, _skin1(O, 1), np.array([1], np.uint32), mats).tolist() == fr.tolist()
    sv = np.zeros((1, 3), O.SKIN_VERTEX)
    sv["weight"][:, :, :2] = 0.5
    sv["bone"][:, :, 1] = 1
    sv["num_bones"] = 2
    out = O.skin_frames(fr, sv, np.zeros(1, np.uint32), mats)
    assert out.tolist() == [[0.5, 0.5, 0, -1, 1, 0, 0, 0, 3, -0.5, 0.5, 0, 0, 1, 0, 0, 0, -1]]
    sv["num_bones"] = 0
    assert O.skin_frames(fr, sv, np.zeros(1, np.uint32), mats).tolist() == [[0] * 18]
    # against the float64 model on a seeded rig
    from vistrace_amd import workloads as W
    n = 200
    rng = np.random.default_rng(4)
    frames = rng.normal(size=(n, 18)).astype(np.float32)
    skin, base, nmat = W.skinned_rig(n, nents=3, bones_per_ent=5)
    bones, binds = W.rig_pose(nmat, frame=1)
    m = O.skin_matrices(bones, binds)
    M64 = m.reshape(nmat, 4, 4).transpose(0, 2, 1).astype(np.float64)
    out = O.skin_frames(frames, skin, base, m).reshape(n, 6, 3)
    exp = np.zeros((n, 6, 3))
    for t in range(n):
        for half in range(2):
            for vi in range(3):
                vec = np.append(frames[t, half * 9 + vi * 3: half * 9 + vi * 3 + 3].astype(np.float64), 0.0)
                for q in range(skin["num_bones"][t, vi]):
                    exp[t, half * 3 + vi] += (M64[base[t] + skin["bone"][t, vi, q]] @ vec)[:3] * skin["weight"][t, vi, q]
    assert np.allclose(out, exp, rtol=1e-4, atol=1e-5)


# ---- alpha test inside intersect(): Primitives.h:196-208 ---------------------------------------------------------
def _alpha_scene(O, va, filter_mode, ref=0.5, tex_mat=((1, 0, 0, 0), (0, 1, 0, 0)), scale=1.0):
    """Unit right triangle in z = 0 with uvs = its xy corners and a 2x2 checker alpha plane [[0,255],[255,0]]."""
    verts = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    tris = O.tris_setup(verts, np.array([O_ALPHA], np.uint8))
    mats = np.zeros(1, O.ALPHA_MATERIAL)
    mats["tex_mat"][0], mats["tex_scale"], mats["alpha_ref"] = tex_mat, scale, ref
    mats["width"], mats["height"], mats["filter"] = 2, 2, filter_mode
    texels = np.array([0, 255, 255, 0], np.uint8)
    O.set_alpha(tris, np.array([[0, 0, 1, 0, 0, 1]], np.float32), np.zeros(1, np.uint32), mats, texels)
    return tris


O_ALPHA = 2


def _hits(O, tris, x, y):
    rays = np.zeros(1, O.RAY)
    rays["org"], rays["dir"], rays["tmax"] = (x, y, 1.0), (0, 0, -1), np.finfo(np.float32).max
    return O.trace_brute(tris, rays)["prim"][0] != O.MISS


def test_alpha_test_nearest_checker(O, va):
    try:
        tris = _alpha_scene(O, va, 0)
        assert not _hits(O, tris, 0.2, 0.2)       # texel (0,0) = 0   -> alpha 0 < 0.5: discarded
        assert _hits(O, tris, 0.7, 0.2)           # texel (1,0) = 255 -> kept
        assert _hits(O, tris, 0.2, 0.7)           # texel (0,1) = 255 -> kept
        O.set_alpha()                             # no side data: the flag is ignored
        assert _hits(O, tris, 0.2, 0.2)
        tris = _alpha_scene(O, va, 0, ref=0.0)    # alpha < 0 never holds
        assert _hits(O, tris, 0.2, 0.2)
        # TransformTexcoord: shift s by half a plane -> the checker flips; scale 2 wraps around
        tris = _alpha_scene(O, va, 0, tex_mat=((1, 0, 0.25, 0.25), (0, 1, 0, 0)))
        assert _hits(O, tris, 0.2, 0.2) and not _hits(O, tris, 0.7, 0.2)
        tris = _alpha_scene(O, va, 0, scale=2.0)
        assert not _hits(O, tris, 0.1, 0.1) and _hits(O, tris, 0.3, 0.1) and not _hits(O, tris, 0.6, 0.1)
    finally:
        O.set_alpha()


def test_alpha_sampler_definition(O):
    m = np.zeros(1, O.ALPHA_MATERIAL)
    m["width"], m["height"], m["tex_scale"] = 2, 2, 1
    tex = np.array([0, 255, 255, 0], np.uint8)
    assert O.alpha_sample(m, tex, 0.25, 0.25) == 0.0 and O.alpha_sample(m, tex, 0.75, 0.25) == 1.0
    assert O.alpha_sample(m, tex, 1.25, -0.75) == 0.0                      # repeat addressing, negative side too
    assert O.alpha_sample(m, tex, float("nan"), 0.25) == 0.0               # non-finite coordinate -> texel column 0
    m["filter"] = 1
    assert O.alpha_sample(m, tex, 0.25, 0.25) == 0.0                       # texel centre
    assert O.alpha_sample(m, tex, 0.5, 0.25) == 0.5                        # half way between two centres
    assert O.alpha_sample(m, tex, 0.5, 0.5) == 0.5
    assert O.alpha_sample(m, tex, 0.0, 0.25) == 0.5                        # wraps to the last column
    m["width"], m["height"] = 0, 0
    assert O.alpha_sample(m, tex, 0.3, 0.3) == 1.0                         # no texture: opaque


def test_alpha_test_walk_equals_brute_force(O, va):
    """Random scene with ~40 % alpha-tested triangles: the BVH walk and the brute-force loop agree (a discarded hit
    must not shrink tmax nor end the walk)."""
    from vistrace_amd import workloads as W
    rng = np.random.default_rng(17)
    n = 3000
    verts = (rng.normal(scale=30, size=(n, 1, 3)) + rng.normal(scale=4, size=(n, 3, 3))).astype(np.float32)
    flags, attribs, mats, texels = W.alpha_test_rig(n)
    tris64 = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris64)
    otris = O.tris_from_tri64(tris64)
    rays = W.sphere_rays(5000, 3, origin=(1.0, 2.0, 3.0))
    try:
        plain = O.trace_brute(otris, rays)
        O.set_alpha(otris, attribs["uv"].reshape(n, 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref, _, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)
        brute = O.trace_brute(otris, rays)
        assert (ref["t"].view(np.uint32) == brute["t"].view(np.uint32)).all()
        assert ((ref["prim"] == O.MISS) == (brute["prim"] == O.MISS)).all()
        changed = int((plain["t"] != brute["t"]).sum())
        assert 200 < changed < 4000                                         # the test discards a good share of the hits
    finally:
        O.set_alpha()
