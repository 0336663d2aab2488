"""Golden vectors (tests/golden/s1k_golden.npz, made by tests/golden/make_golden.py).

CPU part: the oracle must reproduce the committed vectors on the committed tree, agree with
the brute-force intersector, and the host BVH builder must rebuild the committed tree from
the committed vertices (deterministic build).  The GPU part lives in test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s1k_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


def test_oracle_reproduces_golden(O, gold):
    tris = O.tris_setup(gold["verts"])
    hits, stats, _, _, _ = O.traverse_batch(gold["nodes"].view(O.NODE), gold["prim_indices"], tris,
                                            gold["rays"].view(O.RAY), want_stats=True)
    g = gold["hits"].view(O.HIT)
    assert (hits["prim"] == g["prim"]).all()
    for k in ("t", "u", "v"):
        assert (hits[k].view(np.uint32) == g[k].view(np.uint32)).all()
    assert (stats == gold["stats"]).all()
    occ, _, _, _, _ = O.traverse_batch(gold["nodes"].view(O.NODE), gold["prim_indices"], tris,
                                       gold["rays"].view(O.RAY), any_hit=True)
    assert ((occ["prim"] != O.MISS).astype(np.uint8) == gold["occluded"]).all()


def test_golden_matches_brute_force(O, gold):
    tris = O.tris_setup(gold["verts"])
    rays = gold["rays"].view(O.RAY)
    brute = O.trace_brute(tris, rays)
    g = gold["hits"].view(O.HIT)
    assert ((brute["prim"] == O.MISS) == (g["prim"] == O.MISS)).all()
    assert (brute["t"].view(np.uint32) == g["t"].view(np.uint32)).all()
    for i in np.nonzero(brute["prim"] != g["prim"])[0]:      # tie-broken index: must be in the min-t set
        _, ids, n = O.min_t_set(tris, rays[i:i + 1])
        assert g["prim"][i] in ids[:n]
    # any-hit <=> closest-hit finds something in the same interval
    assert ((g["prim"] != O.MISS).astype(np.uint8) == gold["occluded"]).all()
    assert 0 < int((g["prim"][-256:] != O.MISS).sum()) < 256      # the windowed rays do both


def test_builder_rebuilds_golden_tree(va, gold):
    tris = va.tris_setup(gold["verts"])
    bvh = va.HostBvh(tris, builder="ploc")                     # the fixture holds the PLOC tree
    assert (bvh.nodes().view(np.uint8) == gold["nodes"].view(np.uint8)).all()
    assert (bvh.prim_indices() == gold["prim_indices"]).all()
    bvh1 = va.HostBvh(tris, nthreads=1, builder="ploc")                          # thread count must not change the tree
    assert (bvh1.nodes().view(np.uint8) == gold["nodes"].view(np.uint8)).all()


def test_product_tri_setup_matches_oracle(va, O, gold):
    mine = va.tris_setup(gold["verts"])
    ref = O.tris_setup(gold["verts"])
    for k in ("p0", "e1", "e2", "n"):
        assert (mine[k].view(np.uint32) == ref[k].view(np.uint32)).all()
    assert (mine["prim"] == np.arange(len(mine))).all()


TERRAIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "terrain_golden.npz")


def test_terrain_golden_cull_flags(va, O):
    """One-sided heightfield (VT_TRI_CULL_BACKFACE on every triangle): oracle and builder vs the fixture."""
    g = np.load(TERRAIN)
    tris = O.tris_setup(g["verts"], g["flags"])
    hits, stats, _, _, _ = O.traverse_batch(g["nodes"].view(O.NODE), g["prim_indices"], tris, g["rays"].view(O.RAY), want_stats=True)
    ref = g["hits"].view(O.HIT)
    assert (hits["prim"] == ref["prim"]).all() and (stats == g["stats"]).all()
    for k in ("t", "u", "v"):
        assert (hits[k].view(np.uint32) == ref[k].view(np.uint32)).all()
    n_above = int((ref["prim"][:1024] != O.MISS).sum()); n_below = int((ref["prim"][1024:2048] != O.MISS).sum())
    assert n_above != n_below and n_above > 0 and n_below > 0        # the cull bit changes the answer
    bvh = va.HostBvh(va.tris_setup(g["verts"], g["flags"]), builder="ploc")
    assert (bvh.nodes().view(np.uint8) == g["nodes"].view(np.uint8)).all()


def test_oracle_reproduces_the_shading_frame_fixture(O, gold):
    """tests/golden/shading_frame_golden.npz: the shading frame (CalcTBN without a normal map + CalcFootprint) of every fourth
    hit of the s1k fixture, cone on and off, and the vertex frames skinned by one pose -- the oracle must reproduce them bit for bit
    (the log2 inside the triangle's lod included: same libm on the same inputs)."""
    sf = np.load(os.path.join(os.path.dirname(GOLDEN), "shading_frame_golden.npz"))
    tris = O.tris_setup(gold["verts"])
    sel = sf["ray_index"]
    rays, hits = gold["rays"].view(O.RAY)[sel], gold["hits"].view(O.HIT)[sel]
    for key, cone in (("tbn", tuple(float(x) for x in sf["cone"])), ("tbn_cone_off", (-1.0, -1.0))):
        got = O.hit_tbn(tris, rays, hits, sf["frames"], sf["uv"].reshape(-1, 6), cone[0], cone[1])
        assert (got.view(np.uint8) == sf[key].view(O.TBN).view(np.uint8)).all(), key
    assert (sf["tbn"].view(O.TBN)["lod_set"][hits["prim"] != O.MISS] == 1).all() and not sf["tbn_cone_off"].view(O.TBN)["lod_set"].any()
    skinned = O.skin_frames(sf["frames"], sf["skin"].view(O.SKIN_VERTEX), sf["matrix_base"], O.skin_matrices(sf["bones"], sf["binds"]))
    assert (skinned.view(np.uint32) == sf["skinned_frames"].view(np.uint32)).all()
