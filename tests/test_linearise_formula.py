"""The numbering behind vt_scene_upload_tree (vistrace_amd/csrc/scene_build.hip), restated in numpy and checked against the host
lineariser on the CPU: every node's place in the depth-first order follows from sums over its ancestors, with no walk from the
root --

    below pair P, child x comes behind P itself, P's leaf children (emitted when P is numbered, left before right) and,
    if x is the right child, the whole left subtree.

The device kernels (lin_counts / lin_offsets / lin_emit) evaluate exactly these sums, one thread per node; the GPU tests compare
their output byte for byte (tests/test_gpu_rebuild.py).  This test pins the FORMULA where no GPU is needed."""
import numpy as np
import pytest


def number_nodes(nodes):
    """pair index of every inner node, first triangle slot of every node (pair's block for inner nodes, own slot for leaves), depth"""
    n = len(nodes)
    first, pc = nodes["first"].astype(np.int64), nodes["prim_count"].astype(np.int64)
    inner = pc == 0
    parent = np.full(n, -1, np.int64)
    parent[first[inner]] = np.nonzero(inner)[0]
    parent[first[inner] + 1] = np.nonzero(inner)[0]
    # bottom-up counts (children sit behind their parents in the v1 layout: a descending sweep sees children first)
    cnt, tcount = np.zeros(n, np.int64), pc.copy()
    for i in range(n - 1, -1, -1):
        if inner[i]:
            cnt[i] = 1 + cnt[first[i]] + cnt[first[i] + 1]
            tcount[i] = tcount[first[i]] + tcount[first[i] + 1]
    pidx, tbase, depth = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int64)
    x = np.arange(n)
    at_start = np.ones(n, bool)
    alive = parent[x] >= 0
    while alive.any():
        P = np.where(alive, parent[x], 0)
        L = first[P]
        right = x != L
        pcL, pcR = pc[L], pc[L + 1]
        leaf_step = at_start & ~inner                    # a leaf's first step: its slot inside its pair's own block
        add_t_leaf = np.where(right & (pcL != 0), pcL, 0)
        add_p = 1 + np.where(right & (pcL == 0), cnt[L], 0)
        add_t = pcL + pcR + np.where(right & (pcL == 0), tcount[L], 0)
        tbase += np.where(alive, np.where(leaf_step, add_t_leaf, add_t), 0)
        pidx += np.where(alive & ~leaf_step, add_p, 0)
        depth += alive
        at_start[:] = False
        x = np.where(alive, P, x)
        alive = alive & (parent[x] >= 0)
    return inner, pidx, tbase, depth + 1, cnt, tcount


@pytest.mark.parametrize("builder", ["sah", "ploc"])
def test_depth_first_numbering_from_ancestor_sums(va, builder):
    from vistrace_amd import workloads as W
    for verts in (W.make_scene("S1k"), W.make_scene("S10k")[:3000], W.make_terrain()[0]):
        tris = va.tris_setup(np.ascontiguousarray(verts, np.float32))
        bvh = va.HostBvh(tris, builder=builder)
        hs = va.HostScene(bvh)
        nodes, prims = bvh.nodes(), bvh.prim_indices()
        inner, pidx, tbase, depth, cnt, tcount = number_nodes(nodes)
        assert cnt[0] == hs.pair_count and tcount[0] == len(tris) and depth[inner].max() == hs.max_depth
        pairs, htris = hs.pairs(), hs.tris()
        # what lin_emit writes: pair p = the two children of the inner node numbered p, `first` re-targeted
        for i in np.nonzero(inner)[0][:: max(1, int(inner.sum()) // 400)]:
            p = pidx[i]
            for side in (0, 1):
                c = nodes["first"][i] + side
                want = tbase[c] if nodes["prim_count"][c] else pidx[c]
                got = pairs[p]["child"][side]
                assert got["first"] == want and got["prim_count"] == nodes["prim_count"][c]
                assert got["bounds"].tobytes() == nodes["bounds"][c].tobytes()
        # ... and every leaf's triangles at its slot, in primitive-index order
        leaves = np.nonzero(~inner)[0]
        for i in leaves[:: max(1, len(leaves) // 400)]:
            k = nodes["prim_count"][i]
            assert (htris["prim"][tbase[i]: tbase[i] + k] == prims[nodes["first"][i]: nodes["first"][i] + k]).all()
