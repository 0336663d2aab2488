"""Property test of the oracle's BVH walk (bvh v1 semantics) against its own brute-force
intersector on random small scenes, built by the product's builders (default SAH and PLOC): for every ray the walk returns the
minimum t bit for bit (the slab test never prunes the closest triangle), the reported index lies in the set of
triangles attaining it, any-hit agrees, and the walk never tests more triangles than there are.  CPU only."""
import numpy as np
from hypothesis import given, settings, strategies as st

FLT_MAX = np.finfo(np.float32).max


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 2**31 - 1), ntris=st.integers(1, 400), scale=st.sampled_from([1e-3, 1.0, 50.0, 1e4]),
       cull=st.booleans(), window=st.booleans())
def test_walk_equals_brute_force(va, O, seed, ntris, scale, cull, window):
    rng = np.random.default_rng(seed)
    centres = rng.normal(scale=scale, size=(ntris, 1, 3))
    verts = (centres + rng.normal(scale=0.3 * scale, size=(ntris, 3, 3))).astype(np.float32)
    if ntris > 3:
        verts[1] = verts[0]                                   # duplicate triangle: a tie on t
        verts[2, 1] = verts[2, 0]                             # zero-area triangle
    flags = (rng.integers(0, 2, ntris).astype(np.uint8) if cull else None)
    tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris, builder="ploc" if seed % 2 else "sah")
    otris = O.tris_from_tri64(tris)
    nr = 300
    org = rng.normal(scale=2.0 * scale, size=(nr, 3)).astype(np.float32)
    tgt = verts[rng.integers(0, ntris, nr), rng.integers(0, 3, nr)] + rng.normal(scale=0.05 * scale, size=(nr, 3))
    d = (tgt - org).astype(np.float32)
    d[::7, rng.integers(0, 3)] = 0.0                          # zero direction components: safe_inverse clamp
    rays = va.make_rays(org, d, 0.0, FLT_MAX)
    if window:
        rays["tmin"] = 0.25
        rays["tmax"] = 1.5
    ref, stats, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, want_stats=True)
    brute = O.trace_brute(otris, rays)
    assert (ref["t"].view(np.uint32) == brute["t"].view(np.uint32)).all()
    assert ((ref["prim"] == O.MISS) == (brute["prim"] == O.MISS)).all()
    for i in np.nonzero(ref["prim"] != brute["prim"])[0]:
        _, ids, n = O.min_t_set(otris, rays[i:i + 1])
        assert ref["prim"][i] in ids[:n]
    occ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, any_hit=True)[0]
    assert ((occ["prim"] != O.MISS) == (ref["prim"] != O.MISS)).all()
    assert (stats[:, 1] <= ntris).all()
