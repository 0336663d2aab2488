"""The C oracle against an independent pure-Python fp32 restatement of the walk (oracle/py_walk.py, written from
SURVEY.md 3.2, small cases only): hits bit for bit, same traversal-step and triangle-test counters, i.e. the same
visitation order and tie-breaks -- including duplicate triangles, cull flags, clamped directions and t windows."""
import numpy as np
import pytest

FLT_MAX = np.finfo(np.float32).max


@pytest.mark.parametrize("seed,ntris,any_hit", [(1, 1, False), (2, 7, False), (3, 60, False), (4, 250, False), (5, 120, True), (6, 400, False)])
def test_c_oracle_equals_python_walk(va, O, seed, ntris, any_hit):
    from oracle import py_walk
    rng = np.random.default_rng(seed)
    verts = (rng.normal(scale=20, size=(ntris, 1, 3)) + rng.normal(scale=5, size=(ntris, 3, 3))).astype(np.float32)
    if ntris > 10:
        verts[5] = verts[4]; verts[6] = verts[4]                       # three coincident triangles: ties on t
    flags = rng.integers(0, 2, ntris).astype(np.uint8)
    tris64 = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris64)
    otris = O.tris_from_tri64(tris64)
    nodes, pidx = bvh.nodes().view(O.NODE), bvh.prim_indices()
    nr = 120
    org = rng.normal(scale=40, size=(nr, 3)).astype(np.float32)
    tgt = verts[rng.integers(0, ntris, nr)].mean(axis=1) + rng.normal(scale=1.0, size=(nr, 3))
    d = (tgt - org).astype(np.float32)
    d[::9, 1] = 0.0
    if ntris > 10:
        d[:10] = (verts[4].mean(axis=0) - org[:10]).astype(np.float32)   # aim at the coincident triangles
    rays = va.make_rays(org, d, 0.0, FLT_MAX)
    rays["tmin"][::5] = 0.2
    rays["tmax"][::7] = 1.1
    ref, st, _, _, _ = O.traverse_batch(nodes, pidx, otris, rays, any_hit=any_hit, want_stats=True)
    ties = 0
    for i in range(nr):
        prim, t, u, v, steps, tests = py_walk.walk(nodes, pidx, otris, rays[i], any_hit=any_hit)
        assert prim == ref["prim"][i], (i, prim, ref["prim"][i])
        assert (np.float32(t).view(np.uint32), np.float32(u).view(np.uint32), np.float32(v).view(np.uint32)) == \
               (ref["t"][i:i + 1].view(np.uint32)[0], ref["u"][i:i + 1].view(np.uint32)[0], ref["v"][i:i + 1].view(np.uint32)[0])
        assert (steps, tests) == (st[i, 0], st[i, 1])
        ties += prim in (4, 5, 6)
    if ntris > 10 and not any_hit:
        assert ties >= 1


def test_shading_frame_restated_twice(O):
    """oracle/py_walk.py::hit_tbn (numpy fp32 scalars, written from the reference's text) against vto_hit_tbn: normal / tangent /
    binormal and the cone term bit for bit, both branches of CalcTBN; the triangle's lod within an ulp of log2 (numpy's and libm's
    log2 need not round alike)."""
    from oracle import py_walk as P
    rng = np.random.default_rng(77)
    grazing = 0
    for it in range(1500):
        p = rng.normal(scale=20, size=(3, 3)).astype(np.float32)
        tri = O.tris_setup(p[None])
        normals, tangents = rng.normal(size=(3, 3)).astype(np.float32), rng.normal(size=(3, 3)).astype(np.float32)
        uvs = rng.uniform(-3, 3, (3, 2)).astype(np.float32)
        u = np.float32(rng.random() * 0.7); v = np.float32(rng.random() * (1 - u) * 0.9)
        d = rng.normal(size=3).astype(np.float32)
        if it % 4 == 0:                                    # almost inside the plane: the grazing branch
            d = ((p[1] - p[0]) + np.float32(0.03) * rng.normal(size=3)).astype(np.float32)
        dist = np.float32(rng.random() * 50)
        cone = (np.float32(rng.random()), np.float32(0.001 + rng.random() * 0.05)) if it % 3 else (np.float32(-1), np.float32(-1))
        ray = np.zeros(1, O.RAY); ray["dir"] = d
        hit = np.zeros(1, O.HIT); hit["prim"], hit["t"], hit["u"], hit["v"] = 0, dist, u, v
        ref = O.hit_tbn(tri, ray, hit, np.concatenate([normals.reshape(9), tangents.reshape(9)])[None], uvs.reshape(1, 6), float(cone[0]), float(cone[1]))[0]
        n, t, b, lod = P.hit_tbn((tri["p0"][0], tri["e1"][0], tri["e2"][0], tri["n"][0]), d, dist, u, v, normals, tangents, uvs, cone[0], cone[1])
        same = lambda a, c: np.array_equal(np.array(a, np.float32).view(np.uint32), np.asarray(c, np.float32).view(np.uint32)) or \
            (np.isnan(np.array(a, np.float32)) == np.isnan(np.asarray(c, np.float32))).all() and np.isnan(np.array(a, np.float32)).any()
        assert same(n, ref["normal"]) and same(t, ref["tangent"]) and same(b, ref["binormal"]), it
        assert (lod is not None) == bool(ref["lod_set"])
        if lod is not None:
            assert same([lod[1]], [ref["lod_info"][1]]) and abs(float(lod[0]) - float(ref["lod_info"][0])) <= 4e-7 * max(1.0, abs(float(lod[0])))
        wo = -d / np.linalg.norm(d)
        nn = (1 - u - v) * normals[0] + u * normals[1] + v * normals[2]
        grazing += abs(float(np.dot(wo, nn / np.linalg.norm(nn)))) <= 0.1
    assert grazing > 100


def test_skinning_restated_twice(O):
    """py_walk.mat4_mul / transform_to_bone (numpy fp32, from AccelStruct.cpp:34-92) against vto_skin_matrices, vto_skin_verts and
    vto_skin_frames: bit for bit on a seeded rig (positions through the p0 / e1 / e2 round trip of SkinTriangle :68-72)."""
    from oracle import py_walk as P
    from vistrace_amd import workloads as W
    rng = np.random.default_rng(12)
    n = 120
    verts = rng.normal(scale=40, size=(n, 9)).astype(np.float32)
    frames = rng.normal(size=(n, 18)).astype(np.float32)
    skin, base, nmat = W.skinned_rig(n, nents=3, bones_per_ent=5)
    bones, binds = W.rig_pose(nmat, frame=3)
    mats = O.skin_matrices(bones, binds)
    for k in range(nmat):
        assert np.array_equal(P.mat4_mul(bones[k], binds[k]).view(np.uint32), mats[k].view(np.uint32))
    got_v = O.skin_verts(verts, skin, base, mats)
    got_f = O.skin_frames(frames, skin, base, mats)
    for t in range(n):
        ent = mats[base[t]:]
        b = verts[t]
        p0 = b[0:3]; e1 = b[0:3] - b[3:6]; e2 = b[6:9] - b[0:3]
        pos = [p0, p0 - e1, p0 + e2]
        for vi in range(3):
            sv = skin[t, vi]
            exp = P.transform_to_bone(pos[vi], ent, sv["num_bones"], sv["weight"], sv["bone"])
            assert np.array_equal(exp.view(np.uint32), got_v[t, vi * 3: vi * 3 + 3].view(np.uint32)), (t, vi)
            for half in range(2):
                vec = frames[t, half * 9 + vi * 3: half * 9 + vi * 3 + 3]
                exp = P.transform_to_bone(vec, ent, sv["num_bones"], sv["weight"], sv["bone"], angle_only=True)
                assert np.array_equal(exp.view(np.uint32), got_f[t, half * 9 + vi * 3: half * 9 + vi * 3 + 3].view(np.uint32)), (t, vi, half)


def test_trace_result_fields_restated_twice(O):
    """py_walk.hit_attrs (numpy fp32, from TraceResult.cpp:56-85, 255-262) against vto_hit_attrs / vto_hit_shade, bit for bit."""
    from oracle import py_walk as P
    rng = np.random.default_rng(5)
    eq = lambda a, b: np.array_equal(np.array(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
    for it in range(800):
        p = rng.normal(scale=10.0 ** rng.uniform(-2, 2), size=(3, 3)).astype(np.float32)
        tri = O.tris_setup(p[None])
        u = np.float32(rng.random() * 0.8); v = np.float32(rng.random() * (1 - u))
        d = (rng.normal(size=3) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        uvs, alphas = rng.uniform(-2, 2, (3, 2)).astype(np.float32), rng.random(3).astype(np.float32)
        ray = np.zeros(1, O.RAY); ray["dir"] = d
        hit = np.zeros(1, O.HIT); hit["prim"], hit["u"], hit["v"] = 0, u, v
        ref = O.hit_attrs(tri, ray, hit)[0]
        got = P.hit_attrs((tri["p0"][0], tri["e1"][0], tri["e2"][0], tri["n"][0]), d, u, v, uvs, alphas)
        assert eq(got["wo"], ref["wo"]) and eq(got["uvw"], ref["uvw"]) and eq(got["ngeo"], ref["ngeo"]) and eq(got["pos"], ref["pos"]), it
        assert got["front"] == bool(ref["front"])
        tex, blend = O.hit_shade(u, v, uvs, alphas)
        assert eq(got["tex_uv"], tex) and eq([got["blend"]], [blend])
