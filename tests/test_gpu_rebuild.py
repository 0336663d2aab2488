"""Rebuild's upload step with the re-packing done on the device (vt_scene_upload_tree, vistrace_amd/csrc/scene_build.hip) against the
host lineariser (vt_scene_linearise + vt_scene_upload): the records on the device must be BYTE-EQUAL, the index tables must drive a
refit to the same bytes, hits and counters must equal the oracle's, and the host copy fetched back (vt_host_scene_download) must
equal the host lineariser's output.  Stands where the reference constructs its intersector / traverser over the finished tree
(source/objects/AccelStruct.cpp:772-773)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenes(W):
    yield "S1k", W.make_scene("S1k"), None
    yield "S10k", W.make_scene("S10k"), None
    verts, flags = W.make_terrain()
    yield "terrain", verts, flags
    yield "S100k", W.make_scene("S100k"), None


@pytest.mark.parametrize("builder", ["sah", "ploc", "sah_refined"])
def test_device_linearise_is_byte_equal_to_the_host_lineariser(va, O, engine, builder):
    from vistrace_amd import workloads as W
    for name, verts, flags in _scenes(W):
        tris = va.tris_setup(verts, flags)
        bvh = va.HostBvh(tris, builder=builder)
        host = va.Scene(engine, va.HostScene(bvh))
        dev = va.Scene.from_tree(engine, bvh)
        assert dev.upload_stats()["linearised_on_device"] == 1 and host.upload_stats()["linearised_on_device"] == 0
        hp, ht = host.read_records()
        dp, dt = dev.read_records()                       # (fetches the host copy: vt_host_scene_download)
        assert hp.tobytes() == dp.tobytes(), f"{name}/{builder}: pair records differ"
        assert ht.tobytes() == dt.tobytes(), f"{name}/{builder}: triangle records differ"
        assert dev.device_bytes == host.device_bytes
        # the downloaded host scene equals the host lineariser's output (pairs, triangles, depth, and its own single-ray walk)
        assert dev.host_scene.pairs().tobytes() == host.host_scene.pairs().tobytes()
        assert dev.host_scene.tris().tobytes() == host.host_scene.tris().tobytes()
        assert (dev.host_scene.max_depth, dev.host_scene.pair_count, dev.host_scene.tri_count, dev.host_scene.root_leaf_count) == \
               (host.host_scene.max_depth, host.host_scene.pair_count, host.host_scene.tri_count, host.host_scene.root_leaf_count)
        rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(20000, 5)])
        ref, ref_stats = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris), rays, want_stats=True)[:2]
        got = dev.trace_closest(rays)
        assert got.tobytes() == ref.tobytes(), f"{name}/{builder}: hits differ from the oracle"
        assert dev.host_scene.trace_closest_host(rays[:2000]).tobytes() == ref[:2000].tobytes()
        # the index tables (triangle -> slot, pairs by level): a refit through them writes the same bytes as through the host-built ones
        moved = (np.asarray(verts, np.float32) * np.float32(1.03125) + np.float32(0.5)).astype(np.float32)
        host.refit(moved, flags)
        dev.refit(moved, flags)
        hp, ht = host.read_records()
        dp, dt = dev.read_records()
        assert hp.tobytes() == dp.tobytes() and ht.tobytes() == dt.tobytes(), f"{name}/{builder}: records differ after a refit"
        # a host scene downloaded from the device can be uploaded again (it carries the pairs' depths)
        again = va.Scene(engine, dev.host_scene)
        dev.sync_host_scene()
        ap, at = again.read_records()
        assert len(ap) == len(dp)
        again.free(); host.free(); dev.free()


def test_device_linearise_on_degenerate_trees(va, engine):
    """no triangles, one triangle (the root is a leaf), two triangles (one pair of leaves): the device path serves them all"""
    from vistrace_amd import workloads as W
    rays = W.sphere_rays(3000, 11, origin=(0.2, 0.1, -3.0))
    rays["dir"][:1500] = (0.0, 0.0, 1.0)
    one = np.array([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]], np.float32)
    far = np.array([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]], [[50, 50, 50], [51, 50, 50], [50, 51, 50]], [[-60, 2, 3], [-61, 2, 3], [-60, 3, 3]]], np.float32)
    many = np.concatenate([one + np.float32(k) for k in range(40)])
    for verts in (np.zeros((0, 3, 3), np.float32), one, far, many):
        tris = va.tris_setup(verts)
        bvh = va.HostBvh(tris)
        host = va.Scene(engine, va.HostScene(bvh))
        dev = va.Scene.from_tree(engine, bvh)
        assert dev.trace_closest(rays).tobytes() == host.trace_closest(rays).tobytes()
        assert (dev.trace_any(rays) == host.trace_any(rays)).all()
        hp, ht = host.read_records()
        dp, dt = dev.read_records()
        assert hp.tobytes() == dp.tobytes() and ht.tobytes() == dt.tobytes()
        host.free(); dev.free()


def test_device_linearise_with_alpha_tested_triangles(va, O, engine):
    """the alpha flag is found on the device (room for the AlphaRecs behind the triangles), tables set afterwards as usual"""
    from vistrace_amd import workloads as W
    verts = np.ascontiguousarray(W.make_scene("S10k"), np.float32)
    flags, attribs, mats, texels = W.alpha_test_rig(len(verts))
    tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris)
    host = va.Scene(engine, va.HostScene(bvh))
    dev = va.Scene.from_tree(engine, bvh)
    assert dev.device_bytes == host.device_bytes
    for sc in (host, dev):
        sc.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
        sc.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
    rays = W.sphere_rays(40000, 23, origin=(3.0, -2.0, 5.0))
    assert dev.trace_closest(rays).tobytes() == host.trace_closest(rays).tobytes()
    ot = O.tris_from_tri64(tris)
    try:
        O.set_alpha(ot, attribs["uv"].reshape(len(verts), 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), ot, rays)[0]
    finally:
        O.set_alpha()
    assert dev.trace_closest(rays).tobytes() == ref.tobytes()
    host.free(); dev.free()


def test_every_host_copy_learns_of_a_device_refit(va, engine):
    """A scene may have several host copies (the host scene it was uploaded from, copies fetched with vt_host_scene_download): a
    device-side refit marks ALL of them stale, and refreshing one (vt_host_scene_sync) does not un-stale the others."""
    from vistrace_amd import workloads as W
    L = va._lib
    verts = np.ascontiguousarray(W.make_scene("S1k"), np.float32)
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris)
    first = va.HostScene(bvh)
    scene = va.Scene(engine, first)
    second = scene.download_host_scene()                  # (scene.host_scene is now `second`)
    rays = W.sphere_rays(500, 3)
    assert first.trace_closest_host(rays).tobytes() == second.trace_closest_host(rays).tobytes()
    moved = (verts + np.float32(0.125)).astype(np.float32)
    scene.refit(moved)
    for copy in (first, second):
        with pytest.raises(L.VisTraceError, match="stale|refit|sync"):
            copy.trace_closest_host(rays)
    L.check(L.lib.vt_host_scene_sync(second._h, scene._h))
    fresh = second.trace_closest_host(rays)
    assert fresh.tobytes() == scene.trace_closest(rays).tobytes()
    with pytest.raises(L.VisTraceError):
        first.trace_closest_host(rays)                    # still the old records: still refused
    L.check(L.lib.vt_host_scene_sync(first._h, scene._h))
    assert first.trace_closest_host(rays).tobytes() == fresh.tobytes()
    scene.free()


def _nested_triangles(n=600, ratio=1.03):
    """n triangles around one axis, each 3 % larger than the one before: the reference's builder algorithm (PLOC: the two
    smallest clusters are the only mutual nearest neighbours of a round) chains them into a tree ~n levels deep.  Every triangle
    lies in a slightly tilted plane of its own (z = c_k y + k / 1000), so a ray down the axis enters nearly every box of the chain
    before its closest hit: the walk really is hundreds of levels deep."""
    k = np.arange(n)
    s = ratio ** k
    c, h = 0.001 * (k % 7 + 1), 1e-3 * k
    verts = np.zeros((n, 3, 3), np.float32)
    verts[:, 0] = np.stack([-s, -s * 0.5, c * (-s * 0.5) + h], 1)
    verts[:, 1] = np.stack([s, -s * 0.5, c * (-s * 0.5) + h], 1)
    verts[:, 2] = np.stack([np.zeros(n), s, c * s + h], 1)
    return verts


def test_trees_deeper_than_255_levels(va, O, engine):
    """Round 5's device-side index tables refused trees deeper than 255 levels (an 8-bit radix sort over the depth, fixed read-back
    offsets) although the host lineariser, the host walk and the launch planner take any depth.  A 582-level PLOC chain: both upload
    paths, byte-equal records, refit through the level lists, hits and counters equal the oracle's (stack: 10 entries in LDS, the
    rest in the per-lane overflow area)."""
    from vistrace_amd import workloads as W
    verts = _nested_triangles()
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris, builder="ploc")
    hs = va.HostScene(bvh)
    assert hs.max_depth > 500
    host = va.Scene(engine, hs)
    dev = va.Scene.from_tree(engine, bvh)
    hp, ht = host.read_records()
    dp, dt = dev.read_records()
    assert hp.tobytes() == dp.tobytes() and ht.tobytes() == dt.tobytes()
    assert dev.host_scene.max_depth == hs.max_depth
    rng = np.random.default_rng(5)
    org = rng.uniform(-3, 3, (6000, 3)).astype(np.float32)
    org[:, 2] = np.where(rng.random(6000) < 0.5, 50.0, -50.0)
    d = np.zeros((6000, 3), np.float32)
    d[:, 2] = -np.sign(org[:, 2])
    d[:, :2] = rng.normal(scale=0.2, size=(6000, 2))
    rays = va.make_rays(org, d)
    ref, ref_st = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris), rays, want_stats=True)[:2]
    assert int((ref["prim"] != 0xFFFFFFFF).sum()) > 3000 and int(ref_st[:, 0].max()) > 500 and len(np.unique(ref["prim"])) > 50   # rays that walk the chain
    for sc in (host, dev):
        assert sc.trace_closest(rays).tobytes() == ref.tobytes()
        assert (sc.trace_any(rays) == (ref["prim"] != 0xFFFFFFFF)).all()
    moved = (verts * np.float32(1.25)).astype(np.float32)
    host.refit(moved)
    dev.refit(moved)
    hp, ht = host.read_records()
    dp, dt = dev.read_records()
    assert hp.tobytes() == dp.tobytes() and ht.tobytes() == dt.tobytes()
    bvh.refit(va.tris_setup(moved))
    ref2 = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(va.tris_setup(moved)), rays)[0]
    assert dev.trace_closest(rays).tobytes() == ref2.tobytes()
    host.free(); dev.free()


def test_malformed_trees_are_refused_without_touching_memory_out_of_bounds(va, engine):
    """vt_scene_upload_tree numbers the tree on the device from parent links it derives itself.  A tree whose child indices point
    backwards, outside the array, or at a node another parent already claims must be refused ("malformed tree") BEFORE any kernel
    follows such a link -- and the engine must go on working.  The tree is corrupted in place through the accessor's pointer."""
    import ctypes as C
    from vistrace_amd import workloads as W
    L = va._lib
    verts = np.ascontiguousarray(W.make_scene("S1k"), np.float32)
    tris = va.tris_setup(verts)
    rays = W.sphere_rays(2000, 3)
    good = va.Scene.from_tree(engine, va.HostBvh(tris))
    ref = good.trace_closest(rays)
    good.free()

    def corrupt(mutate):
        bvh = va.HostBvh(tris)
        n = L.lib.vt_bvh_node_count(bvh._h)
        nodes = np.ctypeslib.as_array(C.cast(L.lib.vt_bvh_nodes(bvh._h), C.POINTER(C.c_uint8)), shape=(n * 32,)).view(L.BVH_NODE)
        inner = [i for i in range(n) if nodes["prim_count"][i] == 0]
        mutate(nodes, inner, n)
        with pytest.raises(L.VisTraceError, match="malformed tree|bad prim index"):
            va.Scene.from_tree(engine, bvh)

    def backwards(nodes, inner, n):             # an inner node deep in the tree points back at the root's children: a cycle
        nodes["first"][inner[-1]] = 1

    def outside(nodes, inner, n):
        nodes["first"][inner[len(inner) // 2]] = n + 1000

    def claimed_twice(nodes, inner, n):         # two parents for one pair of children
        nodes["first"][inner[1]] = nodes["first"][inner[2]]

    def leaf_range_outside(nodes, inner, n):
        leaf = next(i for i in range(n) if nodes["prim_count"][i] != 0)
        nodes["first"][leaf] = 0xFFFFFF00

    for mutate in (backwards, outside, claimed_twice, leaf_range_outside):
        corrupt(mutate)
        again = va.Scene.from_tree(engine, va.HostBvh(tris))          # the engine (and its staging block) are still sound
        assert again.trace_closest(rays).tobytes() == ref.tobytes(), mutate.__name__
        again.free()
