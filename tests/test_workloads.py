"""Synthetic workload generators (vistrace_amd/workloads.py) against the oracle's restatement
of vistrace.CalcRayOrigin (VisTrace.cpp:1495-1517) and hemisphere_cos (BSDF.cpp:69-77)."""
import numpy as np


def test_scene_sizes_and_determinism():
    from vistrace_amd import workloads as W
    a, b = W.make_scene("S10k"), W.make_scene("S10k")
    assert a.shape == (9932, 3, 3) and (a == b).all()
    assert W.make_scene("S1k").shape == (1152, 3, 3)
    assert np.abs(a).max() <= 1000.0
    m, sub, k = W.SCENES["S1M"]
    assert m * 20 * 4 ** sub + 12 * k * k == 1000300
    m, sub, k = W.SCENES["S100k"]
    assert m * 20 * 4 ** sub + 12 * k * k == 100012
    m, sub, k = W.SCENES["S10M"]
    assert abs(m * 20 * 4 ** sub + 12 * k * k - 10_000_000) < 100_000


def test_primary_rays_shape():
    from vistrace_amd import workloads as W
    r = W.primary_rays(16, 8)
    assert len(r) == 128 and np.allclose(np.linalg.norm(r["dir"], axis=1), 1, atol=1e-6)
    assert (r["tmin"] == 0).all() and (r["tmax"] == np.finfo(np.float32).max).all()
    assert r["dir"][:, 0].min() > 0.5           # looking down +x
    assert r["dir"][0, 2] > 0 > r["dir"][-1, 2]  # row-major, top row first


def test_calc_ray_origin_matches_oracle(O):
    from vistrace_amd import workloads as W
    rng = np.random.default_rng(5)
    pos = rng.uniform(-900, 900, (500, 3)).astype(np.float32)
    pos[:50] *= 1e-4                              # exercise the |pos| < 1/32 branch
    nrm = rng.normal(size=(500, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    got = W.calc_ray_origin(pos, nrm)
    for i in range(len(pos)):
        assert (got[i].view(np.uint32) == O.calc_ray_origin(pos[i], nrm[i]).view(np.uint32)).all()


def test_hemisphere_cos_matches_oracle(O):
    from vistrace_amd import workloads as W
    u = W.uniform01(77, 0, 400).reshape(200, 2)
    got = W.hemisphere_cos(u[:, 0], u[:, 1])
    ref = np.stack([O.hemisphere_cos(a, b) for a, b in u])
    assert np.abs(got - ref).max() <= 2e-7        # numpy vs libm cosf/sinf differ by an ulp at most


def test_bounce_and_shadow_rays(O, make_bundle, va):
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    prim = W.primary_rays(32, 32)
    hits = b.oracle(prim)
    a0 = O.hit_attrs(b.otris, prim, hits)
    attrs = np.zeros(len(a0), va.HIT_ATTRS)
    for k in ("pos", "ngeo", "uvw", "wo", "front"):
        attrs[k] = a0[k]
    attrs["hit"] = hits["prim"] != O.MISS
    br = W.bounce_rays(attrs, 3)
    nrm = np.where((attrs["front"] != 0)[:, None], attrs["ngeo"], -attrs["ngeo"])
    assert ((br["dir"] * nrm).sum(1) > -1e-6).all()            # in the hemisphere of the facing normal
    assert np.allclose(np.linalg.norm(br["dir"], axis=1), 1, atol=1e-5)
    bh = b.oracle(br)
    assert (bh["prim"] == hits["prim"]).mean() < 0.01          # origin offset avoids self-hits
    sr = W.shadow_rays(attrs, W.light_positions("S1k"), 4, per_hit=2)
    assert len(sr) == 2 * len(attrs) and (sr["tmax"] > 0).all()


def test_brush_hall_scene(O, va):
    from vistrace_amd import workloads as W
    """HALL100k (workloads.make_hall): the Source-map-like scene -- a handful of huge one-quad brush faces crossing many
    small prop triangles: size, determinism, extents, and the host walk against the oracle and brute force on it."""
    verts = W.make_scene("HALL100k")
    assert verts.shape == (98340, 3, 3) and np.isfinite(verts).all()
    assert (W.make_scene("HALL100k") == verts).all()
    area = 0.5 * np.linalg.norm(np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0]), axis=1)
    assert area[:12].min() > 1e5 and np.median(area) < 50.0            # the shell's triangles dwarf the props'
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris)
    hs = va.HostScene(bvh)
    rays = np.concatenate([W.primary_rays(48, 48), W.sphere_rays(3000, 7, origin=(-300.0, 250.0, 40.0))])
    ref = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris), rays)[0]
    got = hs.trace_closest_host(rays)
    assert (got["prim"] == ref["prim"]).all() and (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    assert (ref["prim"] != 0xFFFFFFFF).all()                             # a closed hall: every ray ends on something
    brute = O.trace_brute(O.tris_from_tri64(tris), rays[:600])
    assert (brute["t"].view(np.uint32) == ref["t"][:600].view(np.uint32)).all()
