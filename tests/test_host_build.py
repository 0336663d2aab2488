"""Host side of Rebuild (CPU): BVH build (AccelStruct.cpp:763-770 equivalent) and the
re-pack for the device.  Structural invariants + edge cases; no GPU needed."""
import numpy as np
import pytest


def check_tree(va, tris, bvh):
    nodes, pidx = bvh.nodes(), bvh.prim_indices()
    n = len(tris)
    if n == 0:
        assert len(nodes) == 0 and len(pidx) == 0
        return
    assert sorted(pidx.tolist()) == list(range(n))            # every triangle exactly once
    seen = np.zeros(len(nodes), bool)
    seen[0] = True
    leaves = 0
    for i, nd in enumerate(nodes):
        assert seen[i], "unreachable node"
        b = nd["bounds"]
        if nd["prim_count"] == 0:
            l = int(nd["first"])
            assert l > i and l % 2 == 1 and l + 1 < len(nodes)   # children adjacent, after the parent, left odd
            for c in (nodes[l], nodes[l + 1]):
                cb = c["bounds"]
                assert (cb[0::2] >= b[0::2]).all() and (cb[1::2] <= b[1::2]).all()
            seen[l] = seen[l + 1] = True
        else:
            leaves += 1
            first, cnt = int(nd["first"]), int(nd["prim_count"])
            for idx in pidx[first:first + cnt]:
                t = tris[idx]
                p = np.stack([t["p0"], t["p0"] - t["e1"], t["p0"] + t["e2"]])
                assert (p.min(0) >= b[0::2]).all() and (p.max(0) <= b[1::2]).all()
    assert len(nodes) == 2 * leaves - 1


def check_linearised(hs, bvh, tris):
    nodes, pidx = bvh.nodes(), bvh.prim_indices()
    pairs, ltris = hs.pairs(), hs.tris()
    assert len(ltris) == len(tris)
    if len(nodes) == 0:
        assert len(pairs) == 0 and hs.root_leaf_count == 0
        return
    if nodes[0]["prim_count"] != 0:
        assert hs.root_leaf_count == nodes[0]["prim_count"] and len(pairs) == 0
        assert ltris["prim"].tolist() == pidx.tolist()
        return
    assert len(pairs) == (len(nodes) - 1) // 2
    # walk both trees together: same bounds, same leaf contents, depth-first numbering
    order = []
    stack = [(int(nodes[0]["first"]), 0, 1)]
    max_depth = 0
    while stack:
        fc, pi, depth = stack.pop()
        order.append(pi)
        max_depth = max(max_depth, depth)
        for side in (1, 0):
            nd, ch = nodes[fc + side], pairs[pi]["child"][side]
            assert (nd["bounds"] == ch["bounds"]).all() and nd["prim_count"] == ch["prim_count"]
            if nd["prim_count"] == 0:
                stack.append((int(nd["first"]), int(ch["first"]), depth + 1))
            else:
                f, c = int(nd["first"]), int(nd["prim_count"])
                got = ltris["prim"][int(ch["first"]):int(ch["first"]) + c]
                assert got.tolist() == pidx[f:f + c].tolist()
    assert order == list(range(len(pairs)))                   # depth-first, left-first numbering
    assert hs.max_depth == max_depth
    # two leaf siblings are contiguous in the leaf-ordered triangle array (the kernel relies on it)
    for p in pairs:
        l, r = p["child"]
        if l["prim_count"] and r["prim_count"]:
            assert int(r["first"]) == int(l["first"]) + int(l["prim_count"])
    src = tris[ltris["prim"]]
    for k in ("p0", "e1", "e2", "n", "flags"):
        assert (ltris[k] == src[k]).all()


@pytest.mark.parametrize("builder", ["sah", "ploc", "sah_refined"])
@pytest.mark.parametrize("name", ["S1k", "S10k"])
def test_scene_trees(va, name, builder):
    from vistrace_amd import workloads as W
    tris = va.tris_setup(W.make_scene(name))
    bvh = va.HostBvh(tris, builder=builder)
    check_tree(va, tris, bvh)
    check_linearised(va.HostScene(bvh), bvh, tris)
    counts = bvh.nodes()["prim_count"]
    assert counts.max() <= 16 and (counts > 1).any()          # multi-triangle leaves (SAH termination / PLOC leaf collapse)


@pytest.mark.parametrize("builder", ["sah", "ploc", "sah_refined"])
@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 64, 257])
def test_small_and_empty(va, n, builder):
    rng = np.random.default_rng(n)
    verts = rng.uniform(-10, 10, (n, 3, 3)).astype(np.float32)
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris, builder=builder)
    check_tree(va, tris, bvh)
    hs = va.HostScene(bvh)
    check_linearised(hs, bvh, tris)
    if n == 1:
        assert hs.root_leaf_count == 1


def test_degenerate_inputs(va):
    # all triangles identical (same centroid/Morton code), and a flat scene (zero extent in z)
    same = np.tile(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32), (100, 1, 1))
    for verts in (same, np.concatenate([same, same + np.array([3, 0, 0], np.float32)])):
        tris = va.tris_setup(verts)
        for builder in ("sah", "ploc", "sah_refined"):
            bvh = va.HostBvh(tris, builder=builder)
            check_tree(va, tris, bvh)
            check_linearised(va.HostScene(bvh), bvh, tris)


def test_flags_travel_to_records(va):
    from vistrace_amd import workloads as W
    verts, flags = W.make_terrain(8)
    flags = flags.copy(); flags[::3] = 0
    tris = va.tris_setup(verts, flags)
    assert (tris["flags"] == flags).all()
    hs = va.HostScene(va.HostBvh(tris))
    lt = hs.tris()
    assert (lt["flags"] == flags[lt["prim"]]).all()


@pytest.mark.parametrize("name", ["S1k", "S10k", "S100k"])
def test_both_builders(va, O, name):
    """Default binned-SAH builder (task-parallel) and the reference-algorithm PLOC builder: same structural invariants,
    both deterministic for any thread count (S100k is large enough for the SAH builder's parallel top + subtree tasks),
    and the oracle finds the same t,u,v on both trees (only tie-broken indices may differ)."""
    from vistrace_amd import workloads as W
    tris = va.tris_setup(W.make_scene(name))
    sah = va.HostBvh(tris, builder="sah")
    assert (va.HostBvh(tris).nodes().view(np.uint8) == sah.nodes().view(np.uint8)).all()      # "sah" IS the default
    check_tree(va, tris, sah)
    check_linearised(va.HostScene(sah), sah, tris)
    for nt in (1, 3):
        again = va.HostBvh(tris, nthreads=nt, builder="sah")
        assert (again.nodes().view(np.uint8) == sah.nodes().view(np.uint8)).all()
        assert (again.prim_indices() == sah.prim_indices()).all()
    ploc = va.HostBvh(tris, builder="ploc")
    check_tree(va, tris, ploc)
    assert (va.HostBvh(tris, nthreads=1, builder="ploc").nodes().view(np.uint8) == ploc.nodes().view(np.uint8)).all()
    rays = np.concatenate([W.primary_rays(48, 48), W.sphere_rays(2000, 4, origin=(100.0, -50.0, 20.0))])
    ot = O.tris_from_tri64(tris)
    a, _, sa, _, _ = O.traverse_batch(sah.nodes().view(O.NODE), sah.prim_indices(), ot, rays)
    b, _, sb, _, _ = O.traverse_batch(ploc.nodes().view(O.NODE), ploc.prim_indices(), ot, rays)
    for k in ("t", "u", "v"):
        same = a["prim"] == b["prim"]
        assert (a[k][same].view(np.uint32) == b[k][same].view(np.uint32)).all()
    assert (a["t"].view(np.uint32) == b["t"].view(np.uint32)).all()
    if name == "S100k":
        assert sa < 0.85 * sb                                  # fewer traversal steps on the SAH tree from 100 k triangles up: why it is the default
    # opt-in refinement (re-insertion of the worst inner nodes): same leaves and triangle order, a valid tree, the same
    # for any thread count, same t,u,v -- and fewer steps where the tree is large enough for it to matter
    ref = va.HostBvh(tris, builder="sah_refined")
    check_tree(va, tris, ref)
    check_linearised(va.HostScene(ref), ref, tris)
    assert (ref.prim_indices() == sah.prim_indices()).all() and len(ref.nodes()) == len(sah.nodes())
    assert (va.HostBvh(tris, nthreads=1, builder="sah_refined").nodes().view(np.uint8) == ref.nodes().view(np.uint8)).all()
    leaves = lambda t: np.sort(t.nodes()[t.nodes()["prim_count"] > 0], order=["first"])
    assert (leaves(ref).view(np.uint8) == leaves(sah).view(np.uint8)).all()
    c, _, sc, _, _ = O.traverse_batch(ref.nodes().view(O.NODE), ref.prim_indices(), ot, rays)
    assert (a["t"].view(np.uint32) == c["t"].view(np.uint32)).all()
    if name == "S100k":
        assert sc < 1.01 * sa                                  # the default builder's region refinement has taken most of it already
    with pytest.raises(KeyError):
        va.HostBvh(tris, builder="nope")


def test_vt_builder_environment_override(va):
    """vt_bvh_build (what the host class calls) follows VT_BUILDER=ploc / sah; vt_bvh_build_ex names the builder."""
    import ctypes as C
    import os
    from vistrace_amd import workloads as W
    tris = va.tris_setup(W.make_scene("S1k"))
    lib = va._lib.lib

    def build_default():
        h = C.c_void_p()
        va._lib.check(lib.vt_bvh_build(va._lib.ptr(tris), len(tris), 0, C.byref(h)))
        n = lib.vt_bvh_node_count(h)
        lib.vt_bvh_free(h)
        return n
    old = os.environ.pop("VT_BUILDER", None)
    try:
        n_default = build_default()
        assert n_default == len(va.HostBvh(tris, builder="sah").nodes())
        os.environ["VT_BUILDER"] = "ploc"
        assert build_default() == len(va.HostBvh(tris, builder="ploc").nodes())
        os.environ["VT_BUILDER"] = "sah_refined"
        assert build_default() == len(va.HostBvh(tris, builder="sah_refined").nodes())
    finally:
        os.environ.pop("VT_BUILDER", None)
        if old is not None:
            os.environ["VT_BUILDER"] = old


def test_host_refit_keeps_topology_and_bounds_the_moved_triangles(va):
    from vistrace_amd import workloads as W
    verts = W.make_scene("S1k")
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris)
    before = bvh.nodes()
    rng = np.random.default_rng(0)
    moved = (verts + rng.normal(scale=3.0, size=verts.shape)).astype(np.float32)
    mtris = va.tris_setup(moved)
    bvh.refit(mtris)
    after = bvh.nodes()
    assert (after["prim_count"] == before["prim_count"]).all() and (after["first"] == before["first"]).all()
    assert not (after["bounds"] == before["bounds"]).all()
    check_tree(va, mtris, bvh)                       # every node still bounds its children / triangles
    check_linearised(va.HostScene(bvh), bvh, mtris)
    bvh.refit(tris)                                  # moving back restores the original bounds exactly
    assert (bvh.nodes().view(np.uint8) == before.view(np.uint8)).all()


def test_default_builder_tree_is_pinned(va):
    """The binned-SAH tree (+ re-insertion) is pinned by hash: round 5 rewrote the builder's working set (triangle records permuted
    in place, occupancy-masked bins, children's bounds accumulated by the partition: 2.3 x faster) and the tree had to stay
    bit-identical to the one rounds 2-4 built -- the headline's steps per ray and every golden vector depend on it."""
    import hashlib
    from vistrace_amd import workloads as W
    want = {("S10k", "sah"): "428727de4d34", ("S10k", "sah_refined"): "36bc20fdf2fc", ("S100k", "sah"): "45e26ff4d67d",
            ("S100k", "sah_refined"): "f7a360558e64"}
    for (name, builder), h in want.items():
        tris = va.tris_setup(W.make_scene(name))
        for nt in (1, 3, 8):
            bvh = va.HostBvh(tris, nthreads=nt, builder=builder)
            assert hashlib.sha1(bvh.nodes().tobytes() + bvh.prim_indices().tobytes()).hexdigest()[:12] == h, (name, builder, nt)
    verts, flags = W.make_terrain()
    bvh = va.HostBvh(va.tris_setup(verts, flags), nthreads=4)
    assert hashlib.sha1(bvh.nodes().tobytes() + bvh.prim_indices().tobytes()).hexdigest()[:12] == "53135742de7e"
