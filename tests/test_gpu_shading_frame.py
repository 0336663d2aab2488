"""The shading frame of a hit on the device (vt_hit_tbn_dev: TraceResult::CalcTBN without a normal map + CalcFootprint,
source/objects/TraceResult.cpp:58-62, 89-103, 132-137, 175-186) and the per-vertex frames that follow the bones
(vt_scene_skin_refit: SkinTriangle's normals / tangents, source/objects/AccelStruct.cpp:82-92) against the oracle.

Bar: normal / tangent / binormal and the cone term are BIT-IDENTICAL to the oracle (same unfused fp32 expression tree, correctly
rounded sqrt and divide on both sides); the triangle's lod goes through log2, which neither side rounds correctly: <= 2e-6
absolute + 1e-6 relative (north_star tolerance for derived floating-point outputs: 1e-5 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

O_MISS = 0xFFFFFFFF


def _rig(va, scene, bundle_verts, seed=5):
    from vistrace_amd import workloads as W
    n = len(bundle_verts)
    rng = np.random.default_rng(seed)
    attribs = np.zeros(n, va.TRI_ATTRIBS)
    attribs["uv"] = rng.uniform(-2, 2, (n, 3, 2)).astype(np.float32)
    attribs["alpha"] = 1.0
    frames = W.vertex_frames(bundle_verts, seed + 1).view(va.TRI_FRAME)
    scene.set_tri_attribs(attribs)
    scene.set_tri_frames(frames)
    return attribs, frames


def _tbn_on_device(va, scene, rays, cone=(-1.0, -1.0)):
    import torch
    from vistrace_amd import torch_plumbing as tp
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.trace_closest(scene, d_rays, len(rays))
    d_out = tp.empty_records(len(rays), va.HIT_TBN, dev)
    d_out.fill_(0x5A)                                             # a miss must be WRITTEN as zeros
    scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), d_out.data_ptr(), cone[0], cone[1], tp.current_stream_handle(dev))
    torch.cuda.synchronize()
    return tp.to_host(d_hits, va.HIT), tp.to_host(d_out, va.HIT_TBN)


def _assert_tbn(got, ref, hit):
    for k in ("normal", "tangent", "binormal"):
        assert (got[k][hit].view(np.uint32) == ref[k][hit].view(np.uint32)).all(), f"{k} not bit-identical"
    assert (got["lod_set"] == ref["lod_set"]).all()
    assert (got["lod_info"][:, 1].view(np.uint32) == ref["lod_info"][:, 1].view(np.uint32)).all(), "cone term not bit-identical"
    a, b = got["lod_info"][hit, 0], ref["lod_info"][hit, 0]
    fin = np.isfinite(b)
    assert (np.abs(a[fin] - b[fin]) <= 2e-6 + 1e-6 * np.abs(b[fin])).all(), "triangle lod beyond the log2 tolerance"
    assert (a[~fin].view(np.uint32) == b[~fin].view(np.uint32)).all() or (np.isnan(a[~fin]) == np.isnan(b[~fin])).all()
    assert (got.view(np.uint8).reshape(len(got), -1)[~hit] == 0).all(), "a miss must read all zeros"


@pytest.mark.parametrize("cone", [(-1.0, -1.0), (0.0, 0.002), (0.5, 0.01), (1.0, 0.0), (-1.0, 0.3)])
def test_hit_tbn_vs_oracle(va, engine, make_bundle, O, cone):
    """Camera rays, rays from inside and deliberately grazing rays (the |cos| <= 0.1 branch of CalcTBN must be exercised),
    a few misses; with the cone on and off (the two error-free mipOverride cases of AccelStruct.cpp:796-803 included)."""
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    scene = va.Scene(engine, b.host_scene)
    attribs, frames = _rig(va, scene, b.verts)
    rng = np.random.default_rng(11)
    # grazing: aim at a point of a triangle along a direction almost inside its plane
    k = 3000
    t = rng.integers(0, len(b.verts), k)
    v = b.verts[t].astype(np.float64)
    bary = rng.dirichlet([1, 1, 1], k)
    target = (v * bary[:, :, None]).sum(axis=1)
    face = np.cross(v[:, 0] - v[:, 1], v[:, 2] - v[:, 0]); face /= np.linalg.norm(face, axis=1, keepdims=True)
    inplane = v[:, 1] - v[:, 0]; inplane /= np.linalg.norm(inplane, axis=1, keepdims=True)
    d = inplane + face * rng.uniform(-0.12, 0.12, (k, 1))
    graze = va.make_rays((target - d * 0.05).astype(np.float32), d.astype(np.float32), 0.0, np.finfo(np.float32).max)
    rays = np.concatenate([W.primary_rays(48, 48), W.sphere_rays(3000, 17, origin=(0.0, 0.0, 0.0)), graze,
                           va.make_rays([[0, 0, 0]] * 8, [[1, 0, 0]] * 8, 0.0, 1e-3)])
    hits, got = _tbn_on_device(va, scene, rays, cone)
    hit = hits["prim"] != O_MISS
    assert hit.sum() > 5000 and (~hit).sum() >= 8
    ref = O.hit_tbn(b.otris, rays, hits, frames.view(np.float32).reshape(-1, 18), attribs["uv"].reshape(-1, 6), cone[0], cone[1])
    # the grazing branch was taken often (recomputed here from the pre-correction normal): enough to trust the comparison
    at = O.hit_attrs(b.otris, rays, hits)
    w = at["uvw"][:, 2:3]; u = at["uvw"][:, 0:1]; vv = at["uvw"][:, 1:2]
    p = np.where(hit, hits["prim"], 0)
    nrm = w * frames["normal"][p, 0] + u * frames["normal"][p, 1] + vv * frames["normal"][p, 2]
    with np.errstate(invalid="ignore"):
        nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)        # misses: 0 / 0, masked by `hit` below
    grazing = hit & (np.abs((at["wo"] * nrm).sum(axis=1)) <= 0.1)
    assert grazing.sum() > 200
    _assert_tbn(got, ref, hit)
    on = not (cone[0] < 0 or cone[1] <= 0)
    assert (got["lod_set"][hit] == (1 if on else 0)).all()
    # the frame is a frame: unit vectors (the un-normalised inputs included), tangent perpendicular to normal after the correction
    for kk in ("normal", "tangent", "binormal"):
        assert np.abs(np.linalg.norm(got[kk][hit].astype(np.float64), axis=1) - 1).max() < 1e-5
    assert np.abs((got["tangent"][grazing].astype(np.float64) * got["normal"][grazing]).sum(axis=1)).max() < 1e-5


def test_hit_tbn_needs_its_tables(va, engine, make_bundle):
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    scene = va.Scene(engine, b.host_scene)
    dev = torch.device("cuda", 0)
    rays = W.sphere_rays(100, 3)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.trace_closest(scene, d_rays, len(rays))
    d_out = tp.empty_records(len(rays), va.HIT_TBN, dev)
    frames = W.vertex_frames(b.verts).view(va.TRI_FRAME)
    with pytest.raises(va._lib.VisTraceError, match="vt_scene_set_tri_frames"):
        scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), d_out.data_ptr())
    with pytest.raises(va._lib.VisTraceError, match="read_tri_frames|set_tri_frames"):
        scene.read_tri_frames()
    with pytest.raises(va._lib.VisTraceError, match="triangle count"):
        scene.set_tri_frames(frames[:-1])
    bt = scene.trace_batch(rays)                                                 # traced before the frames were there
    with pytest.raises(va._lib.VisTraceError, match="vertex frames"):
        bt.tbn()
    bt.free()
    scene.set_tri_frames(frames)
    with pytest.raises(va._lib.VisTraceError, match="vt_scene_set_tri_attribs"):    # the cone asks for the triangle's lod: uvs
        scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), d_out.data_ptr(), 0.5, 0.01)
    assert (scene.read_tri_frames().view(np.uint8) == frames.view(np.uint8)).all()
    scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), 0, 0)              # n = 0: nothing to do, no buffers needed
    # cone off (accel:Traverse's defaults): the frame alone needs no uvs -- and the batch object carries it
    scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), d_out.data_ptr())
    torch.cuda.synchronize()
    bt = scene.trace_batch(rays)
    assert (bt.tbn().view(np.uint8) == tp.to_host(d_out, va.HIT_TBN).view(np.uint8)).all() and (bt.tbn()["lod_set"] == 0).all()
    bt.free()


def test_skinned_frames_follow_the_bones(va, engine, O):
    """vt_scene_skin_refit moves the per-vertex normals / tangents as SkinTriangle does (angleOnly): byte-equal to the oracle
    frame after frame (always from the bind pose), and the shading frame of hits on the posed scene equals the oracle's."""
    from vistrace_amd import workloads as W
    verts = W.make_scene("S10k")
    n = len(verts)
    bvh = va.HostBvh(va.tris_setup(verts))
    scene = va.Scene(engine, va.HostScene(bvh))
    attribs, frames = _rig(va, scene, verts, seed=9)
    skin, base, nmat = W.skinned_rig(n)
    scene.set_skin(verts, skin, base)
    rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(4000, 37, origin=(40.0, -60.0, 70.0))])
    bind18 = frames.view(np.float32).reshape(n, 18)
    for frame in range(3):
        bones, binds = W.rig_pose(nmat, frame)
        scene.skin_refit(bones, binds)
        mats = O.skin_matrices(bones, binds)
        ref_frames = O.skin_frames(bind18, skin, base, mats)
        got_frames = scene.read_tri_frames()
        assert (got_frames.view(np.uint8).reshape(n, 72) == ref_frames.view(np.uint8).reshape(n, 72)).all()
        assert not (got_frames.view(np.uint8) == frames.view(np.uint8)).all()
        posed = O.skin_verts(verts.reshape(n, 9), skin, base, mats).reshape(n, 3, 3)
        otris = O.tris_from_tri64(va.tris_setup(posed))
        hits, got = _tbn_on_device(va, scene, rays, (0.1, 0.004))
        hit = hits["prim"] != O_MISS
        assert hit.sum() > 1000
        ref = O.hit_tbn(otris, rays, hits, ref_frames, attribs["uv"].reshape(-1, 6), 0.1, 0.004)
        _assert_tbn(got, ref, hit)
    # new bind-pose frames replace the skinned ones until the next skin refit
    scene.set_tri_frames(frames)
    assert (scene.read_tri_frames().view(np.uint8) == frames.view(np.uint8)).all()


def test_batch_object_carries_the_frame(va, engine, make_bundle, O):
    """vt_batch_tbn: the batch object (what accel:TraverseBatch(buffer) returns) materialises the frame with the cone off, as
    accel:Traverse's default arguments do; equal to vt_hit_tbn_dev on the same hits; sets and empty batches too."""
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    scene = va.Scene(engine, b.host_scene)
    attribs, frames = _rig(va, scene, b.verts, seed=21)
    rays = np.concatenate([W.sphere_rays(5000, 5), va.make_rays([[0, 0, 0]] * 3, [[1, 0, 0]] * 3, 0.0, 1e-3)])
    hits, dev_tbn = _tbn_on_device(va, scene, rays)
    bt = scene.trace_batch(rays)
    assert (bt.hits().view(np.uint8) == hits.view(np.uint8)).all()
    assert (bt.tbn().view(np.uint8) == dev_tbn.view(np.uint8)).all()
    ref = O.hit_tbn(b.otris, rays, hits, frames.view(np.float32).reshape(-1, 18), attribs["uv"].reshape(-1, 6))
    _assert_tbn(bt.tbn(), ref, hits["prim"] != O_MISS)
    bt.free()
    sets = scene.trace_batch_set([rays[:1000], rays[:0], rays[1000:]], fetch_hits=True)
    assert len(sets[1].tbn()) == 0
    assert (np.concatenate([s.tbn() for s in sets]).view(np.uint8) == dev_tbn.view(np.uint8)).all()
    for s in sets:
        s.free()


def test_shading_frame_golden_fixture(va, engine):
    """The committed fixture (tests/golden/shading_frame_golden.npz, generator beside it) on the device: the frame of every fourth
    hit of the s1k fixture, cone on and off, and the vertex frames after one skinning pose."""
    import os
    import torch
    from vistrace_amd import torch_plumbing as tp
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gold, sf = np.load(os.path.join(gdir, "s1k_golden.npz")), np.load(os.path.join(gdir, "shading_frame_golden.npz"))
    verts = gold["verts"]
    scene = va.Scene(engine, va.HostScene(va.HostBvh(va.tris_setup(verts), builder="ploc")))
    attribs = np.zeros(len(verts), va.TRI_ATTRIBS)
    attribs["uv"] = sf["uv"]
    scene.set_tri_attribs(attribs)
    frames = np.ascontiguousarray(sf["frames"]).view(va.TRI_FRAME).reshape(-1)
    scene.set_tri_frames(frames)
    sel = sf["ray_index"]
    rays, hits = np.ascontiguousarray(gold["rays"].view(va.RAY)[sel]), np.ascontiguousarray(gold["hits"].view(va.HIT)[sel])
    dev = torch.device("cuda", 0)
    d_rays, d_hits = tp.to_device(rays, dev), tp.to_device(hits, dev)            # the fixture's own hit records
    assert (scene.trace_closest(rays).view(np.uint8) == hits.view(np.uint8)).all()
    d_out = tp.empty_records(len(rays), va.HIT_TBN, dev)
    hit = hits["prim"] != O_MISS
    for key, cone in (("tbn", tuple(float(x) for x in sf["cone"])), ("tbn_cone_off", (-1.0, -1.0))):
        scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), len(rays), d_out.data_ptr(), cone[0], cone[1], tp.current_stream_handle(dev))
        torch.cuda.synchronize()
        _assert_tbn(tp.to_host(d_out, va.HIT_TBN), sf[key].view(va.HIT_TBN).reshape(-1), hit)
    scene.set_skin(verts, sf["skin"].view(va.SKIN_VERTEX), sf["matrix_base"])
    scene.skin_refit(sf["bones"], sf["binds"])
    assert (scene.read_tri_frames().view(np.uint32).reshape(-1, 18) == sf["skinned_frames"].view(np.uint32)).all()


def test_hit_tbn_at_the_grazing_threshold(va, engine, O):
    """CalcTBN's correction runs on cosTheta <= 0.1 (source/objects/TraceResult.cpp:175-176).  A direction whose normalised z is
    0.1f to the bit (found by search), a shading normal of (0, 0, 1) and a vertex tangent that is NOT perpendicular to it: AT the
    threshold the frame is corrected (tangent re-orthogonalised to (1, 0, 0)), one float above it is not.  Synthetic (ray, hit)
    records through vt_hit_tbn_dev; the random cases of the other tests never sit on the branch point (round 6: the `<` mutant of
    the host class survived them)."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    verts = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)
    tris = va.tris_setup(verts)
    scene = va.Scene.from_tree(engine, va.HostBvh(tris))
    attribs = np.zeros(1, va.TRI_ATTRIBS)
    attribs["uv"][0] = [[0, 0], [1, 0], [0, 1]]
    frames = np.zeros(1, va.TRI_FRAME)
    frames["normal"][0] = [[0, 0, 1]] * 3
    frames["tangent"][0] = [[0.8, 0, 0.6]] * 3
    scene.set_tri_attribs(attribs)
    scene.set_tri_frames(frames)
    found, base = None, np.float32(0.99498743)
    for i in sorted(range(-2000, 2001), key=abs):
        d = np.array([base + np.float32(i) * np.spacing(base), 0.0, -0.1], np.float32)
        inv = np.float32(1.0) / np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2], dtype=np.float32)
        if -(d[2] * inv) == np.float32(0.1):
            found = d
            break
    assert found is not None, "no direction with a normalised z of exactly 0.1f found"
    above = found.copy()
    above[2] = -np.nextafter(np.float32(0.1), np.float32(1))
    rays = va.make_rays([[0.25, 0.25, 1.0]] * 2, [found, above])
    hits = np.zeros(2, va.HIT)
    hits["prim"], hits["t"], hits["u"], hits["v"] = 0, 2.0, 0.25, 0.25
    dev = torch.device("cuda", 0)
    d_rays, d_hits = tp.to_device(rays, dev), tp.to_device(hits, dev)
    d_out = tp.empty_records(2, va.HIT_TBN, dev)
    scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), 2, d_out.data_ptr(), -1.0, -1.0, tp.current_stream_handle(dev))
    torch.cuda.synchronize()
    got = tp.to_host(d_out, va.HIT_TBN)
    ref = O.hit_tbn(O.tris_from_tri64(tris), rays, hits, frames.view(np.float32).reshape(-1, 18), attribs["uv"].reshape(-1, 6))
    for k in ("normal", "tangent", "binormal"):
        assert (got[k].view(np.uint32) == ref[k].view(np.uint32)).all(), k
    assert got["tangent"][0, 0] == 1.0 and got["tangent"][0, 2] == 0.0            # at the threshold: corrected
    assert got["tangent"][1, 2] != 0.0                                             # one float above: as interpolated
    scene.free()
