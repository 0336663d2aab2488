"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle.

Bar (BASELINE.json north_star): primitive index bit-exact; t, u, v and Pos() within 1e-5
relative -- these tests hold the stronger property that t, u, v are BIT-IDENTICAL (both
sides evaluate the same unfused fp32 expression tree).  Counters (traversal steps and
triangle tests per ray) must match the oracle exactly too: same visitation order.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLT_MAX = np.finfo(np.float32).max
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s1k_golden.npz")
REL_TOL = 1e-5   # north_star tolerance for floating-point outputs derived on the device (Pos, normal)


def assert_hits_equal(got, ref):
    assert (got["prim"] == ref["prim"]).all(), f"{int((got['prim'] != ref['prim']).sum())} primitive indices differ"
    for k in ("t", "u", "v"):
        assert (got[k].view(np.uint32) == ref[k].view(np.uint32)).all(), f"{k} not bit-identical"


def upload(va, engine, bundle):
    return va.Scene(engine, bundle.host_scene)


def stats_on_device(va, scene, rays):
    import torch
    from vistrace_amd import torch_plumbing as tp
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    d_hits, d_stats = tp.trace_stats(scene, d_rays, len(rays))
    torch.cuda.synchronize()
    return tp.to_host(d_hits, va.HIT), tp.to_host(d_stats, va.RAY_STATS)


# ---- golden vectors ----------------------------------------------------------------------------
def test_golden_vectors(va, engine):
    gold = np.load(GOLDEN)
    tris = va.tris_setup(gold["verts"])
    scene = va.Scene(engine, va.HostScene(va.HostBvh(tris, builder="ploc")))
    rays = gold["rays"].view(va.RAY)
    assert_hits_equal(scene.trace_closest(rays), gold["hits"].view(va.HIT))
    assert (scene.trace_any(rays) == gold["occluded"]).all()
    _, st = stats_on_device(va, scene, rays)
    assert (st["steps"] == gold["stats"][:, 0]).all() and (st["tests"] == gold["stats"][:, 1]).all()


def test_terrain_golden_vectors(va, engine):
    gold = np.load(os.path.join(os.path.dirname(GOLDEN), "terrain_golden.npz"))
    scene = va.Scene(engine, va.HostScene(va.HostBvh(va.tris_setup(gold["verts"], gold["flags"]), builder="ploc")))
    rays = gold["rays"].view(va.RAY)
    assert_hits_equal(scene.trace_closest(rays), gold["hits"].view(va.HIT))
    _, st = stats_on_device(va, scene, rays)
    assert (st["steps"] == gold["stats"][:, 0]).all() and (st["tests"] == gold["stats"][:, 1]).all()


# ---- analytic known answers, on the device ---------------------------------------------------------
TRI = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)


def one_ray(va, org, d, tmin=0.0, tmax=FLT_MAX):
    return va.make_rays([org], [d], tmin, tmax)


def test_single_triangle_kats(va, engine):
    scene = va.build_scene(engine, TRI)                      # root is a leaf: no slab test at all
    h = scene.trace_closest(one_ray(va, [0.25, 0.5, 1], [0, 0, -1]))
    assert h["prim"][0] == 0 and h["t"][0] == 1.0 and h["u"][0] == 0.25 and h["v"][0] == 0.5
    for x, y, hit in [(0, 0, 1), (1, 0, 1), (0, 1, 1), (0.5, 0.5, 1), (-0.001, 0.5, 0), (0.51, 0.51, 0)]:
        assert (scene.trace_closest(one_ray(va, [x, y, 1], [0, 0, -1]))["prim"][0] != va._lib.VT_MISS) == bool(hit)
    assert scene.trace_closest(one_ray(va, [0.2, 0.2, 0], [1, 0, 0]))["prim"][0] == va._lib.VT_MISS   # 0/0
    for tmin, tmax, hit in [(0, 1.0, 1), (1.0, 2.0, 1), (0, 0.999, 0), (1.001, 5, 0)]:
        h = scene.trace_closest(one_ray(va, [0.25, 0.25, 1], [0, 0, -1], tmin, tmax))
        assert (h["prim"][0] != va._lib.VT_MISS) == bool(hit)
    one_sided = va.build_scene(engine, TRI, np.array([1], np.uint8))
    assert one_sided.trace_closest(one_ray(va, [0.2, 0.2, 1], [0, 0, -1]))["prim"][0] == va._lib.VT_MISS
    assert one_sided.trace_closest(one_ray(va, [0.2, 0.2, -1], [0, 0, 1]))["prim"][0] == 0


def test_coincident_triangles_tie_break(va, engine, O):
    verts = np.concatenate([TRI] * 6)                        # six identical triangles: every t ties
    b_tris = va.tris_setup(verts)
    bvh = va.HostBvh(b_tris)
    scene = va.Scene(engine, va.HostScene(bvh))
    rays = one_ray(va, [0.25, 0.25, 1], [0, 0, -1])
    ref, _, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(b_tris), rays)
    assert_hits_equal(scene.trace_closest(rays), ref)        # same later-visited-wins index as the oracle


def test_empty_scene_and_empty_batch(va, engine):
    scene = va.build_scene(engine, np.zeros((0, 3, 3), np.float32))
    rays = va.make_rays([[0, 0, 0]] * 5, [[1, 0, 0]] * 5)
    h = scene.trace_closest(rays)
    assert (h["prim"] == va._lib.VT_MISS).all() and (h["t"] == 0).all()
    assert (scene.trace_any(rays) == 0).all()
    scene2 = va.build_scene(engine, TRI)
    assert len(scene2.trace_closest(rays[:0])) == 0


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 257, 1000])
def test_ragged_batch_sizes(va, engine, make_bundle, n):
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    scene = upload(va, engine, b)
    rays = W.sphere_rays(n, 1000 + n, origin=(10.0, 20.0, 30.0))
    assert_hits_equal(scene.trace_closest(rays), b.oracle(rays))


# ---- seeded scenes -------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["S1k", "S10k", "HALL100k"])     # HALL100k: a brush hall -- huge faces across small props
def test_scene_parity_with_counters(va, engine, make_bundle, name):
    from vistrace_amd import workloads as W
    b = make_bundle(name)
    scene = upload(va, engine, b)
    prim = W.primary_rays(128, 64)
    sph = W.sphere_rays(8192, 5, origin=(-300.0, 250.0, 40.0))
    win = sph[:2048].copy(); win["tmin"], win["tmax"] = 350.0, 1200.0
    rays = np.concatenate([prim, sph, win])
    ref, ref_st = b.oracle(rays, want_stats=True)
    assert_hits_equal(scene.trace_closest(rays), ref)
    got, st = stats_on_device(va, scene, rays)
    assert_hits_equal(got, ref)
    assert (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()
    occ = scene.trace_any(rays)
    assert (occ == (ref["prim"] != O_MISS)).all()
    any_ref, any_st = b.oracle(rays, any_hit=True, want_stats=True)
    assert (occ == (any_ref["prim"] != O_MISS)).all()
    # any-hit counters (vt_trace_any_stats_dev): the steps and tests of the reference's walk up to its early-out
    import torch
    from vistrace_amd import torch_plumbing as tp
    d_occ, d_st = tp.trace_any_stats(scene, tp.to_device(rays, torch.device("cuda", 0)), len(rays))
    torch.cuda.synchronize()
    st_any = tp.to_host(d_st, va.RAY_STATS)
    assert (d_occ.cpu().numpy() == occ).all()
    assert (st_any["steps"] == any_st[:, 0]).all() and (st_any["tests"] == any_st[:, 1]).all()


O_MISS = 0xFFFFFFFF


def test_single_ray_config1(va, engine, make_bundle, O):
    """BASELINE config 1: S10k, single-ray calls, uniform sphere directions from the centre;
    parity vs the brute-force intersector (t identical, index in the min-t set)."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    rays = W.sphere_rays(10000, W.SEED + 1)
    single = np.concatenate([scene.trace_closest(rays[i:i + 1]) for i in range(64)])   # truly one ray per call
    batch = scene.trace_closest(rays)
    assert_hits_equal(single, batch[:64])
    brute = O.trace_brute(b.otris, rays)
    assert (batch["t"].view(np.uint32) == brute["t"].view(np.uint32)).all()
    for i in np.nonzero(batch["prim"] != brute["prim"])[0]:
        _, ids, n = O.min_t_set(b.otris, rays[i:i + 1])
        assert batch["prim"][i] in ids[:n]


def test_backface_cull_terrain(va, engine, make_bundle):
    from vistrace_amd import workloads as W
    b = make_bundle("terrain")
    scene = upload(va, engine, b)
    down = W.sphere_rays(4096, 21, origin=(0.0, 0.0, 60.0))      # from above and below the heightfield
    up = W.sphere_rays(4096, 22, origin=(3.0, -4.0, -30.0))
    rays = np.concatenate([down, up])
    ref = b.oracle(rays)
    got = scene.trace_closest(rays)
    assert_hits_equal(got, ref)
    n_down, n_up = int((got["prim"][:4096] != O_MISS).sum()), int((got["prim"][4096:] != O_MISS).sum())
    assert n_down > 0 and n_up > 0 and n_down != n_up            # the cull bit really changes the answer


def test_weird_rays(va, engine, make_bundle):
    """Zero / negative-zero / tiny direction components (safe_inverse clamp), NaN and inf inputs,
    NaN ranges: whatever the reference algorithm yields, the device yields the same."""
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    scene = upload(va, engine, b)
    base = W.sphere_rays(512, 9, origin=(5.0, 6.0, 7.0))
    rays = np.concatenate([base] * 8)
    d = rays["dir"]
    d[0:512, 0] = 0.0
    d[512:1024, 1] = -0.0
    d[1024:1536, 2] = 1e-9
    d[1536:2048, :2] = 0.0                                       # axis-parallel
    rays["tmin"][2048:2560] = np.nan
    rays["tmax"][2560:3072] = np.nan
    d[3072:3328, 0] = np.nan
    d[3328:3584, 1] = np.inf
    rays["org"][3584:3840, 2] = np.nan
    rays["tmax"][3840:] = 1e-30                                  # nothing in range
    ref, ref_st = b.oracle(rays, want_stats=True)
    got, st = stats_on_device(va, scene, rays)
    assert_hits_equal(got, ref)
    assert_hits_equal(scene.trace_closest(rays), ref)
    assert (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()
    # every kind of non-finite origin / direction: the reference walks (possibly the whole tree) and misses;
    # the product kernels skip the walk and report the same miss, any-hit included
    bad = np.concatenate([base[:64]] * 9)
    vals = [np.nan, np.inf, -np.inf]
    for k in range(9):
        bad["dir" if k % 2 == 0 else "org"][k * 64:(k + 1) * 64, k % 3] = vals[k // 3]
    bad["dir"][512:] = np.nan                                    # fully poisoned: the reference visits every pair
    bad["dir"][544:] = np.inf
    ref_bad, st_bad = b.oracle(bad, want_stats=True)
    assert (ref_bad["prim"] == O_MISS).all() and st_bad[512:544, 0].min() == b.host_scene.pair_count
    assert_hits_equal(scene.trace_closest(bad), ref_bad)
    assert (scene.trace_any(bad) == 0).all()
    got_bad, st2 = stats_on_device(va, scene, bad)
    assert_hits_equal(got_bad, ref_bad)
    assert (st2["steps"] == st_bad[:, 0]).all() and (st2["tests"] == st_bad[:, 1]).all()


def test_launch_options_do_not_change_results(va, make_bundle):
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    eng = va.Engine(0)
    scene = va.Scene(eng, b.host_scene)
    rays = np.concatenate([W.primary_rays(96, 96), W.sphere_rays(6000, 3, origin=(100.0, 100.0, -200.0))])
    ref = b.oracle(rays)
    configs = [dict(persistent=0), dict(persistent=2), dict(persistent=2, auto_static_factor=0), dict(persistent=1, coherent_detect=0),
               dict(persistent=0, static_overflow_mb=0), dict(persistent=1, fetch_dma=0), dict(persistent=1, fetch_dma=1, lds_entries=2),
               dict(persistent=1, fetch_dma=1, lds_entries=64), dict(refill_threshold=1, tri_threshold=1, block_rays=64),
               dict(refill_threshold=64, tri_threshold=64, block_rays=4096), dict(blocks_per_cu=1),
               dict(fetch_dma=0, lds_entries=1, refill_threshold=3, tri_threshold=7),
               dict(persistent=1, fetch_dma=1, lds_entries=10, xcd_cursors=1, block_rays=64),
               dict(persistent=1, xcd_cursors=1, block_rays=4096, blocks_per_cu=2), dict(xcd_cursors=0, spin_wait=0),
               dict(persistent=1, max_claim=8, block_rays=64), dict(persistent=1, max_claim=1), dict(max_claim=0)]
    for cfg in configs:
        for k, v in cfg.items():
            eng.set_option(k, v)
        assert_hits_equal(scene.trace_closest(rays), ref)
        assert (scene.trace_any(rays) == (ref["prim"] != O_MISS)).all()


def test_hit_attrs_vs_oracle(va, engine, make_bundle, O):
    """TraceResult core (TraceResult.cpp:45-86, 255-262) on the device: Pos(), uvw, geometric
    normal, wo within 1e-5 relative; frontFacing and indices exact."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(4096, 8, origin=(-50.0, 10.0, 300.0))])
    rays["dir"][:100] *= 3.5                                     # un-normalised directions too
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.trace_closest(scene, d_rays, len(rays))
    attrs = tp.to_host(tp.hit_attrs(scene, d_rays, d_hits, len(rays)), va.HIT_ATTRS)
    hits = tp.to_host(d_hits, va.HIT)
    ref_hits = b.oracle(rays)
    assert_hits_equal(hits, ref_hits)
    ref = O.hit_attrs(b.otris, rays, ref_hits)
    hit = ref_hits["prim"] != O_MISS
    assert (attrs["hit"] == hit).all() and (attrs["prim"] == ref_hits["prim"])[hit].all()
    assert (attrs["t"].view(np.uint32) == ref_hits["t"].view(np.uint32))[hit].all()
    assert (attrs["front"] == ref["front"])[hit].all()
    for k in ("pos", "uvw", "ngeo", "wo"):
        err = np.abs(attrs[k][hit] - ref[k][hit])
        scale = np.maximum(np.abs(ref[k][hit]).max(axis=1, keepdims=True), 1e-30)
        assert (err / scale).max() <= REL_TOL, k
    # ... and, stronger than the 1e-5 the task asks for: the device evaluates the same unfused fp32 expression trees with correctly
    # rounded divides and square roots, so every field is BIT-identical (round 6: the tolerance alone let `w = 1 - (u + v)` for
    # `1 - u - v` through -- mutant 21 of scripts/mutants.sh)
    for k in ("pos", "uvw", "ngeo", "wo"):
        assert (attrs[k][hit].view(np.uint32) == ref[k][hit].view(np.uint32)).all(), k
    # A record that no trace produces but the entry point accepts: a ray PARALLEL to its triangle.  dot(wo, n) is exactly 0 and
    # TraceResult.cpp:85 says frontFacing = dot >= 0 (mutant 22: > 0).
    flat = va.make_rays([[0.0, 0.0, 1.0]] * 4, [[1.0, 0.0, 0.0], [0.0, -2.0, 0.0], [3.0, 4.0, 0.0], [-1.0, 1.0, 0.0]])
    tri = va.tris_setup(np.array([[[-5, -5, 0], [5, -5, 0], [0, 5, 0]]], np.float32))
    fscene = va.Scene.from_tree(engine, va.HostBvh(tri))
    fhits = np.zeros(4, va.HIT)
    fhits["prim"], fhits["t"], fhits["u"], fhits["v"] = 0, 1.0, 0.25, 0.25
    d_fr, d_fh = tp.to_device(flat, dev), tp.to_device(fhits, dev)
    fat = tp.to_host(tp.hit_attrs(fscene, d_fr, d_fh, 4), va.HIT_ATTRS)
    fref = O.hit_attrs(O.tris_from_tri64(tri), flat, fhits)
    assert (fref["front"] == 1).all() and (fat["front"] == 1).all()
    assert (fat["pos"].view(np.uint32) == fref["pos"].view(np.uint32)).all()
    fscene.free()


# ---- BASELINE sizes ----------------------------------------------------------------------------------
def test_config2_primary_1m_rays_s100k(va, engine, make_bundle):
    """BASELINE config 2: 1024x1024 pinhole into the 100k-triangle scene, all 1 048 576 rays
    against the oracle (the oracle finishes this in well under a second on the GPU host)."""
    from vistrace_amd import workloads as W
    b = make_bundle("S100k")
    scene = upload(va, engine, b)
    rays = W.primary_rays(1024, 1024)
    assert_hits_equal(scene.trace_closest(rays), b.oracle(rays))


def test_full_size_properties_s1m(va, engine, make_bundle):
    """BASELINE config 3/4 sizes: 16 Mi incoherent rays into the 1M-triangle scene.  The oracle
    covers a 1 Mi-ray slice; the whole batch is checked through size-independent properties:
    determinism across launch modes, any-hit <=> closest-hit, t inside [tmin, tmax], the hit
    point lies on the reported triangle's plane and inside it (barycentrics), misses are clean."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S1M")
    scene = upload(va, engine, b)
    dev = torch.device("cuda", 0)
    side = 4096
    n = side * side
    prim = W.primary_rays(side, side)
    d_prim = tp.to_device(prim, dev)
    d_h0 = tp.trace_closest(scene, d_prim, n)
    attrs = tp.to_host(tp.hit_attrs(scene, d_prim, d_h0, n), va.HIT_ATTRS)
    rays = W.bounce_rays(attrs, W.SEED + 3)
    del attrs, d_prim, d_h0
    d_rays = tp.to_device(rays, dev)
    hits = tp.to_host(tp.trace_closest(scene, d_rays, n), va.HIT)
    # oracle slice
    sl = slice(0, 1 << 20)
    assert_hits_equal(hits[sl], b.oracle(rays[sl]))
    # determinism across the other launch modes (static one-ray-per-lane kernel, direct fetch)
    saved = {k: engine.get_option(k) for k in ("persistent", "fetch_dma", "coherent_detect")}
    for cfg in (dict(persistent=0), dict(persistent=1, fetch_dma=0), dict(persistent=1, fetch_dma=1, coherent_detect=0)):
        for k, v in cfg.items():
            engine.set_option(k, v)
        again = tp.to_host(tp.trace_closest(scene, d_rays, n), va.HIT)
        assert (again.view(np.uint8) == hits.view(np.uint8)).all()
    for k, v in saved.items():
        engine.set_option(k, v)
    # any-hit <=> closest-hit
    occ = tp.trace_any(scene, d_rays, n).cpu().numpy()
    hit = hits["prim"] != O_MISS
    assert (occ == hit).all()
    assert (hits["t"][~hit] == 0).all() and (hits["u"][~hit] == 0).all()
    assert hit.mean() > 0.99                                      # closed room
    h, r = hits[hit], rays[hit]
    assert (h["t"] >= r["tmin"]).all() and (h["t"] <= r["tmax"]).all()
    assert (h["u"] >= 0).all() and (h["v"] >= 0).all() and (h["u"] + h["v"] <= 1 + 1e-6).all()
    assert h["prim"].max() < len(b.tris)
    # org + t*dir reproduces the barycentric point on the triangle
    T = b.tris[h["prim"]]
    w = 1.0 - h["u"].astype(np.float64) - h["v"]
    pos_tri = (w[:, None] * T["p0"] + h["u"][:, None] * (T["p0"].astype(np.float64) - T["e1"]) +
               h["v"][:, None] * (T["p0"].astype(np.float64) + T["e2"]))
    pos_ray = r["org"].astype(np.float64) + h["t"][:, None].astype(np.float64) * r["dir"]
    err = np.abs(pos_tri - pos_ray).max(axis=1)
    assert np.percentile(err, 99.9) < 0.05 and err.max() < 2.0    # Source units; fp32 t at ~1e3 scale


def test_batch_object_matches_the_separate_calls(va, engine, make_bundle):
    """vt_batch (what accel:TraverseBatch(buffer) holds): hits, attrs and shade fetched lazily equal the separate
    vt_trace_closest / vt_hit_attrs_dev / vt_hit_shade_dev calls byte for byte."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    n = len(b.tris)
    attribs = np.zeros(n, va.TRI_ATTRIBS)
    rng = np.random.default_rng(5)
    attribs["uv"] = rng.random((n, 3, 2), dtype=np.float32)
    attribs["alpha"] = rng.random((n, 3), dtype=np.float32)
    attribs["ent_id"] = rng.integers(0, 100, n)
    attribs["material"] = rng.integers(0, 7, n)
    rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(5000, 12, origin=(3000.0, 0.0, 0.0))])   # hits and misses
    batch = scene.trace_batch(rays)
    assert len(batch) == len(rays)
    with pytest.raises(va._lib.VisTraceError):
        batch.shade()                                           # no side table yet when the batch was traced
    hits = batch.hits()
    assert_hits_equal(hits, b.oracle(rays))
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.trace_closest(scene, d_rays, len(rays))
    ref_attrs = tp.to_host(tp.hit_attrs(scene, d_rays, d_hits, len(rays)), va.HIT_ATTRS)
    assert (batch.attrs().view(np.uint8) == ref_attrs.view(np.uint8)).all()
    batch.free()
    scene.set_tri_attribs(attribs)
    batch2 = scene.trace_batch(rays)
    d_shade = tp.empty_records(len(rays), va.HIT_SHADE, dev)
    scene.hit_shade_dev(d_hits.data_ptr(), len(rays), d_shade.data_ptr(), tp.current_stream_handle(dev))
    torch.cuda.synchronize()
    assert (batch2.shade().view(np.uint8) == tp.to_host(d_shade, va.HIT_SHADE).view(np.uint8)).all()
    assert (batch2.hits().view(np.uint8) == hits.view(np.uint8)).all()
    empty = scene.trace_batch(rays[:0])
    assert len(empty) == 0 and len(empty.hits()) == 0
    scene.free()


def test_batch_pipeline_options(va, engine, make_bundle):
    """vt_batch_trace_closest_ex: the batch streams to the device in 256 Ki-ray chunks; with VT_BATCH_FETCH_HITS the hit records
    come back behind the trace, with VT_BATCH_CHECK_RANGES the staging copy applies the range checks of AccelStruct::Traverse
    (AccelStruct.cpp:805-806) and names the FIRST offender; an image width tiles every chunk.  Every form = the plain batch."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    side = 640                                                              # 409 600 rays: two chunks, the second one ragged
    rays = np.concatenate([W.primary_rays(side, side), W.sphere_rays(70001, 9, origin=(2.0, 1.0, 0.5))])
    plain = scene.trace_batch(rays)
    ref = plain.hits().copy()
    assert_hits_equal(ref[:20000], b.oracle(rays[:20000]))
    for kw in ({"fetch_hits": True}, {"check_ranges": True}, {"fetch_hits": True, "check_ranges": True, "image_width": side},
               {"image_width": 100}, {"fetch_hits": True, "image_width": 4}):
        bt = scene.trace_batch(rays, **kw)
        assert bt.hits().tobytes() == ref.tobytes(), kw
        assert bt.attrs().tobytes() == plain.attrs().tobytes(), kw
        assert bt.rays().tobytes() == rays.tobytes(), kw
        bt.free()
    # the first bad ray is reported, wherever it sits (first chunk, chunk boundary, last ray), and no batch comes back
    for bad_at in ([5], [262143, 262144, 300000], [len(rays) - 1], [400000, 7]):
        r = rays.copy()
        for k, i in enumerate(bad_at):
            if k % 2 == 0:
                r["tmin"][i] = -1.0
            else:
                r["tmax"][i] = r["tmin"][i]
        with pytest.raises(ValueError, match=f"ray {min(bad_at)}:"):
            scene.trace_batch(r, check_ranges=True, fetch_hits=True)
        assert scene.trace_batch(r).hits() is not None                      # without the flag such rays are traced (and miss)
    nan = rays[:1000].copy()
    nan["tmax"][::7] = np.nan                                               # NaN ranges pass the reference's checks too (and miss)
    bt = scene.trace_batch(nan, check_ranges=True, fetch_hits=True)
    assert (bt.hits()["prim"][::7] == O_MISS).all()
    for n in (0, 1, 63, 300):                                               # tiny batches through the same pipeline
        bt = scene.trace_batch(rays[:n], check_ranges=True, fetch_hits=True)
        assert bt.hits().tobytes() == ref[:n].tobytes()
    # several batches alive at once, freed out of order (device and pinned blocks are recycled)
    live = [scene.trace_batch(rays[k * 1000:(k + 40) * 1000], fetch_hits=True) for k in range(6)]
    for k in (3, 0, 5, 1, 4, 2):
        assert live[k].hits().tobytes() == ref[k * 1000:(k + 40) * 1000].tobytes()
        live[k].free()
    scene.free()


def test_batch_outliving_its_engine(va, make_bundle):
    """vt_engine_close releases the device arrays of live batches: what a batch has downloaded stays readable, what it has
    not is lost (the getter fails, nothing touches freed memory), and freeing the batch afterwards is safe."""
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    eng = va.Engine(0)
    scene = va.Scene(eng, b.host_scene)
    rays = W.sphere_rays(5000, 21)
    batch = scene.trace_batch(rays)
    hits = batch.hits()                                        # downloaded before the engine goes away
    assert_hits_equal(hits, b.oracle(rays))
    scene.free()
    eng.close()
    assert (batch.hits().view(np.uint8) == hits.view(np.uint8)).all()
    with pytest.raises(va._lib.VisTraceError) as err:
        batch.attrs()
    assert "engine has been closed" in str(err.value)
    batch.free()


def test_hit_shade_vs_oracle(va, engine, make_bundle, O):
    """texUV / blendFactor / entIdx / submatIdx (TraceResult.cpp:70,73-78) from the per-triangle
    side table: floats bit-identical to the oracle, ids exact, misses flagged."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    scene = upload(va, engine, b)
    rng = np.random.default_rng(3)
    n_tri = len(b.tris)
    attribs = np.zeros(n_tri, va.TRI_ATTRIBS)
    attribs["uv"] = rng.uniform(-2, 2, (n_tri, 3, 2)).astype(np.float32)
    attribs["alpha"] = rng.uniform(0, 1, (n_tri, 3)).astype(np.float32)
    attribs["ent_id"] = rng.integers(0, 65535, n_tri)
    attribs["material"] = rng.integers(0, 100, n_tri)
    scene.set_tri_attribs(attribs)
    rays = np.concatenate([W.sphere_rays(3000, 17, origin=(0.0, 0.0, 0.0)), va.make_rays([[0, 0, 0]] * 8, [[1, 0, 0]] * 8, 0.0, 1e-3)])
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.trace_closest(scene, d_rays, len(rays))
    d_out = tp.empty_records(len(rays), va.HIT_SHADE, dev)
    scene.hit_shade_dev(d_hits.data_ptr(), len(rays), d_out.data_ptr(), tp.current_stream_handle(dev))
    out = tp.to_host(d_out, va.HIT_SHADE)
    hits = tp.to_host(d_hits, va.HIT)
    hit = hits["prim"] != O_MISS
    assert hit[:3000].all() and not hit[3000:].any()
    assert (out["ent_id"][~hit] == O_MISS).all() and (out["material"][~hit] == O_MISS).all() and (out["blend"][~hit] == 0).all()
    assert (out["ent_id"][hit] == attribs["ent_id"][hits["prim"][hit]]).all()
    assert (out["material"][hit] == attribs["material"][hits["prim"][hit]]).all()
    for i in np.nonzero(hit)[0][:500]:
        a = attribs[hits["prim"][i]]
        tex, blend = O.hit_shade(hits["u"][i], hits["v"][i], a["uv"], a["alpha"])
        assert (out["tex_uv"][i].view(np.uint32) == tex.view(np.uint32)).all()
        assert np.float32(out["blend"][i]).view(np.uint32) == blend.view(np.uint32)
    with pytest.raises(va._lib.VisTraceError):
        scene.set_tri_attribs(attribs[:-1])


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_triangle_soups(va, engine, O, seed):
    """Seeded random soups: mixed scales, degenerate (zero-area / collinear / duplicate) triangles,
    random cull flags, rays from inside/outside with random windows -- device == oracle, bit for bit,
    and oracle == brute force on t (index inside the min-t set)."""
    from vistrace_amd import workloads as W
    verts, flags, org, d, tmin, tmax = W.random_soup(seed)
    tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris)
    scene = va.Scene(engine, va.HostScene(bvh))
    rays = va.make_rays(org, d, tmin, tmax)
    otris = O.tris_from_tri64(tris)
    ref, ref_st, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, want_stats=True)
    got, st = stats_on_device(va, scene, rays)
    assert_hits_equal(got, ref)
    assert (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()
    assert (scene.trace_any(rays) == (ref["prim"] != O_MISS)).all()
    brute = O.trace_brute(otris, rays)
    assert ((brute["prim"] == O_MISS) == (ref["prim"] == O_MISS)).all()
    assert (brute["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()
    for i in np.nonzero(brute["prim"] != ref["prim"])[0][:200]:
        _, ids, cnt = O.min_t_set(otris, rays[i:i + 1], max_ids=64)
        assert ref["prim"][i] in ids[:cnt]


def test_device_ray_generation_matches_host(va, engine, make_bundle, O):
    """vt_gen_primary_dev / vt_gen_bounce_dev against the numpy generators (workloads.py): origins and
    RNG samples exact, directions within 1e-6 (device vs host cos/sin/double rounding), and the traced
    results of both ray sets agree wherever the rays are bit-identical."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    dev = torch.device("cuda", 0)
    w, h = 96, 64
    cam = (12.5, -30.0, 7.0)
    d_prim = tp.empty_records(w * h, va.RAY, dev)
    engine.gen_primary_dev(w, h, d_prim.data_ptr(), pos=cam, forward=(1.0, 0.2, -0.1), stream=tp.current_stream_handle(dev))
    got = tp.to_host(d_prim, va.RAY)
    ref = W.primary_rays(w, h, pos=cam, forward=(1.0, 0.2, -0.1))
    assert (got["org"] == ref["org"]).all() and (got["tmin"] == 0).all() and (got["tmax"] == ref["tmax"]).all()
    assert np.abs(got["dir"] - ref["dir"]).max() <= 1e-6
    for rw, rh in ((97, 65), (1, 1), (2049, 3)):          # ragged against the 8 x 256 pixels a block generates
        d_r = tp.empty_records(rw * rh + 1, va.RAY, dev)
        d_r.zero_()
        engine.gen_primary_dev(rw, rh, d_r.data_ptr(), pos=cam, forward=(1.0, 0.2, -0.1), stream=tp.current_stream_handle(dev))
        g = tp.to_host(d_r, va.RAY)
        r = W.primary_rays(rw, rh, pos=cam, forward=(1.0, 0.2, -0.1))
        assert (g["org"][:-1] == r["org"]).all() and np.abs(g["dir"][:-1] - r["dir"]).max() <= 1e-6
        assert g["tmax"][-1] == 0 and (g["dir"][-1] == 0).all()       # nothing written behind the image
    d_hits = tp.trace_closest(scene, d_prim, w * h)
    d_attrs = tp.hit_attrs(scene, d_prim, d_hits, w * h)
    attrs = tp.to_host(d_attrs, va.HIT_ATTRS)
    d_b = tp.empty_records(w * h, va.RAY, dev)
    engine.gen_bounce_dev(d_attrs.data_ptr(), w * h, 777, d_b.data_ptr(), stream=tp.current_stream_handle(dev))
    got_b = tp.to_host(d_b, va.RAY)
    ref_b = W.bounce_rays(attrs, 777)
    assert (attrs["hit"] == 1).all()
    assert (got_b["org"].view(np.uint32) == ref_b["org"].view(np.uint32)).all()      # CalcRayOrigin: integer exact
    assert np.abs(got_b["dir"] - ref_b["dir"]).max() <= 2e-6
    assert np.allclose(np.linalg.norm(got_b["dir"], axis=1), 1.0, atol=1e-5)
    # vistrace.CalcRayOrigin at its branch point, source/VisTrace.cpp:1495-1517: abs(pos) < 1/32 takes the float offset, anything else
    # -- exactly 1/32 included -- the integer one (mutant 27 of scripts/mutants.sh: <=).  Records no trace would produce by chance.
    edge = attrs[:64].copy()
    vals = np.array([1 / 32, -1 / 32, np.nextafter(np.float32(1 / 32), np.float32(0)), np.nextafter(np.float32(1 / 32), np.float32(1)), 0.0, -0.0, 1e-30, 7.5], np.float32)
    for j in range(64):
        edge["pos"][j] = (vals[j % 8], vals[(j // 8) % 8], vals[(j * 3 + 1) % 8])
        nrm = np.array([(-1) ** j * 0.6, 0.0, 0.8], np.float32)
        edge["ngeo"][j], edge["wo"][j], edge["front"][j], edge["hit"][j] = nrm, nrm, 1, 1
    d_e = tp.to_device(edge, dev)
    d_eb = tp.empty_records(64, va.RAY, dev)
    engine.gen_bounce_dev(d_e.data_ptr(), 64, 5, d_eb.data_ptr(), stream=tp.current_stream_handle(dev))
    got_e, ref_e = tp.to_host(d_eb, va.RAY), W.bounce_rays(edge, 5)
    assert (got_e["org"].view(np.uint32) == ref_e["org"].view(np.uint32)).all()
    for j in range(64):                                                     # ... and the oracle's restatement says the same
        assert (O.calc_ray_origin(edge["pos"][j], edge["ngeo"][j]).view(np.uint32) == got_e["org"][j].view(np.uint32)).all()
    # a missed record becomes a null ray
    attrs2 = attrs.copy(); attrs2["hit"][::7] = 0
    d_a2 = tp.to_device(attrs2, dev)
    engine.gen_bounce_dev(d_a2.data_ptr(), w * h, 777, d_b.data_ptr(), stream=tp.current_stream_handle(dev))
    nb = tp.to_host(d_b, va.RAY)
    assert (nb["tmax"][::7] == np.float32(1e-30)).all() and (nb["tmax"][1::7] == got_b["tmax"][1::7]).all()
    assert (scene.trace_closest(nb)["prim"][::7] == O_MISS).all()
    assert_hits_equal(scene.trace_closest(got_b), b.oracle(got_b))


def test_auto_mode_picks_kernel_by_batch_size(va, make_bundle):
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    eng = va.Engine(0)                                   # defaults: persistent = 2 (auto)
    scene = va.Scene(eng, b.host_scene)
    assert eng.get_option("persistent") == 2
    small = W.sphere_rays(5000, 1)
    assert_hits_equal(scene.trace_closest(small), b.oracle(small))
    assert eng.get_option("last_persistent") == 0        # a few rays per resident lane -> one ray per lane
    eng.set_option("auto_static_factor", 0)
    assert_hits_equal(scene.trace_closest(small), b.oracle(small))
    assert eng.get_option("last_persistent") == 1 and eng.get_option("last_fetch_dma") == 1


def test_config4_shadow_rays_any_hit(va, engine, make_bundle):
    """BASELINE config 4 (any-hit shadow rays, tMax early-out) at 16 Mi rays on S1M: occluded <=> the
    closest-hit kernel finds a hit in the same interval; oracle any-hit on a 256 Ki slice."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S1M")
    scene = upload(va, engine, b)
    dev = torch.device("cuda", 0)
    side = 2048
    n0 = side * side
    d_prim = tp.empty_records(n0, va.RAY, dev)
    engine.gen_primary_dev(side, side, d_prim.data_ptr(), stream=tp.current_stream_handle(dev))
    d_h = tp.trace_closest(scene, d_prim, n0)
    attrs = tp.to_host(tp.hit_attrs(scene, d_prim, d_h, n0), va.HIT_ATTRS)
    rays = W.shadow_rays(attrs, W.light_positions("S1M"), W.SEED + 4, per_hit=4)      # 16 Mi rays
    n = len(rays)
    d_rays = tp.to_device(rays, dev)
    occ = tp.trace_any(scene, d_rays, n).cpu().numpy()
    hits = tp.to_host(tp.trace_closest(scene, d_rays, n), va.HIT)
    assert (occ == (hits["prim"] != O_MISS)).all()
    assert 0.05 < occ.mean() < 0.95                      # lights are both visible and hidden
    sl = slice(0, 1 << 18)
    ref = b.oracle(rays[sl], any_hit=True)
    assert (occ[sl] == (ref["prim"] != O_MISS)).all()
    assert (hits["t"][hits["prim"] != O_MISS] <= rays["tmax"][hits["prim"] != O_MISS]).all()


@pytest.mark.parametrize("builder", ["ploc", "sah_refined"])
def test_other_trees_on_device(va, engine, O, builder):
    """The kernels are tree-agnostic: on the reference-algorithm PLOC tree (every other scene test runs on the default
    binned-SAH tree; the golden-vector tests on the PLOC tree too) and on the opt-in refined SAH tree the device equals
    the oracle, counters included."""
    from vistrace_amd import workloads as W
    tris = va.tris_setup(W.make_scene("S10k"))
    bvh = va.HostBvh(tris, builder=builder)
    scene = va.Scene(engine, va.HostScene(bvh))
    rays = np.concatenate([W.primary_rays(96, 64), W.sphere_rays(6000, 12, origin=(-200.0, 30.0, 10.0))])
    ref, ref_st, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris), rays, want_stats=True)
    got, st = stats_on_device(va, scene, rays)
    assert_hits_equal(got, ref)
    assert (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()


def test_device_refit_matches_host_refit(va, engine, O):
    """vt_scene_refit (device, level by level) produces exactly the records of the host path
    (vt_tris_setup + vt_bvh_refit + vt_scene_linearise), and tracing the refitted scene equals the
    oracle on the refitted tree and the brute-force intersector on the moved triangles."""
    from vistrace_amd import workloads as W
    verts = W.make_scene("S10k")
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris)
    scene = va.Scene(engine, va.HostScene(bvh))
    rng = np.random.default_rng(1)
    moved = (verts + rng.normal(scale=5.0, size=(len(verts), 1, 3)) + rng.normal(scale=0.5, size=verts.shape)).astype(np.float32)
    scene.refit(moved)
    mtris = va.tris_setup(moved)
    bvh.refit(mtris)
    ref_hs = va.HostScene(bvh)
    pairs, dtris = scene.read_records()
    assert (dtris.view(np.uint8) == ref_hs.tris().view(np.uint8)).all()
    assert (pairs.view(np.uint8) == ref_hs.pairs().view(np.uint8)).all()
    rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(4000, 31, origin=(50.0, 60.0, -70.0))])
    otris = O.tris_from_tri64(mtris)
    ref, _, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)
    got = scene.trace_closest(rays)
    assert_hits_equal(got, ref)
    brute = O.trace_brute(otris, rays)
    assert (brute["t"].view(np.uint32) == got["t"].view(np.uint32)).all()
    # the single-ray path walks the host copy: stale until it is synchronised with the device
    with pytest.raises(va._lib.VisTraceError) as stale:
        scene.host_scene.trace_closest_host(rays)
    assert "vt_host_scene_sync" in str(stale.value)
    scene.sync_host_scene()
    assert (scene.host_scene.tris().view(np.uint8) == dtris.view(np.uint8)).all()
    assert_hits_equal(scene.host_scene.trace_closest_host(rays), ref)
    other = va.HostScene(va.HostBvh(va.tris_setup(W.make_scene("S1k"))))
    with pytest.raises(va._lib.VisTraceError):
        va._lib.check(va._lib.lib.vt_host_scene_sync(other._h, scene._h))
    with pytest.raises(va._lib.VisTraceError):
        scene.refit(moved[:-1])


def test_device_skin_refit_matches_host_pipeline(va, engine, O):
    """vt_scene_skin_refit (matrices -> skinned vertices -> records -> level-wise refit, all on the device)
    equals the oracle's SkinTriangle restatement pushed through the host path (vt_tris_setup + vt_bvh_refit +
    vt_scene_linearise) bit for bit, frame after frame, and tracing the posed scene equals the oracle."""
    from vistrace_amd import workloads as W
    verts = W.make_scene("S10k")
    n = len(verts)
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris)
    scene = va.Scene(engine, va.HostScene(bvh))
    with pytest.raises(va._lib.VisTraceError):
        scene.skin_refit(np.eye(4, dtype=np.float32).reshape(1, 16), np.eye(4, dtype=np.float32).reshape(1, 16))
    skin, base, nmat = W.skinned_rig(n)
    scene.set_skin(verts, skin, base)
    rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(4000, 37, origin=(40.0, -60.0, 70.0))])
    for frame in range(3):
        bones, binds = W.rig_pose(nmat, frame)
        scene.skin_refit(bones, binds)
        posed = O.skin_verts(verts.reshape(n, 9), skin, base, O.skin_matrices(bones, binds)).reshape(n, 3, 3)
        mtris = va.tris_setup(posed)
        bvh.refit(mtris)
        ref_hs = va.HostScene(bvh)
        pairs, dtris = scene.read_records()
        assert (dtris.view(np.uint8) == ref_hs.tris().view(np.uint8)).all()
        assert (pairs.view(np.uint8) == ref_hs.pairs().view(np.uint8)).all()
        otris = O.tris_from_tri64(mtris)
        ref, _, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)
        assert_hits_equal(scene.trace_closest(rays), ref)
    bad = skin.copy()
    bad["num_bones"][0, 0] = 4
    with pytest.raises(va._lib.VisTraceError):
        scene.set_skin(verts, bad, base)


@pytest.mark.parametrize("name", ["terrain", "S10k"])
def test_bounce_loop_matches_composition_and_oracle(va, engine, make_bundle, name):
    """vt_bounce_loop_dev (compacted queue of live paths) row by row against (a) the uncompacted composition
    trace -> vt_hit_attrs_dev -> vt_gen_bounce_dev(seed + d) over all n paths and (b) the oracle's traversal of
    each depth's rays; live counts = hits of the previous depth."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle(name)
    scene = upload(va, engine, b)
    dev = torch.device("cuda", 0)
    n, depth, seed = 20001, 5, 4242
    start = W.sphere_rays(n, 77, origin=(3.0, -4.0, 60.0) if name == "terrain" else (15.0, 25.0, -35.0))
    d_start = tp.to_device(start, dev)
    d_rows, live = tp.bounce_loop(scene, d_start, n, depth, seed)
    torch.cuda.synchronize()
    rows = tp.to_host(d_rows, va.HIT).reshape(depth, n)
    assert (tp.to_host(d_start, va.RAY).view(np.uint8) == start.view(np.uint8)).all()      # input untouched
    d_rays = d_start
    expect_live = n
    for d in range(depth):
        assert live[d] == expect_live
        rays = tp.to_host(d_rays, va.RAY)
        d_hits = tp.trace_closest(scene, d_rays, n)
        hits = tp.to_host(d_hits, va.HIT)
        assert_hits_equal(hits, b.oracle(rays))
        assert_hits_equal(rows[d], hits)
        expect_live = int((hits["prim"] != O_MISS).sum())
        d_attrs = tp.hit_attrs(scene, d_rays, d_hits, n)
        d_next = tp.empty_records(n, va.RAY, dev)
        engine.gen_bounce_dev(d_attrs.data_ptr(), n, seed + d, d_next.data_ptr(), stream=tp.current_stream_handle(dev))
        d_rays = d_next
    if name == "terrain":
        assert live[1] < n and live[-1] < live[1]          # an open scene: paths leave the queue
    # depth 1 = a plain trace; depth 0 and n 0 are no-ops
    d_one, live1 = tp.bounce_loop(scene, d_start, n, 1, seed)
    torch.cuda.synchronize()
    assert_hits_equal(tp.to_host(d_one, va.HIT), rows[0])
    assert live1 == [n]
    assert scene.bounce_loop_dev(d_start.data_ptr(), 0, 3, seed, d_rows.data_ptr()) == [0, 0, 0]


def test_bounce_loop_dead_paths_and_long_queues(va, O):
    """Branch points of the queue step that a 20 001-path loop does not reach, each named by a mutant that survived the suite
    (scripts/mutants.sh 52 / 55 / 56): (a) every path alive going INTO a depth at which some die -- the miss fill is skipped and the
    queue step itself writes the rows of the paths that missed; (b) exactly ONE path dead -- the fill may not be skipped; (c) a queue
    of more than 4096 x 256 entries -- every thread of the one-block scan owns several 16-byte groups of counts and carries between
    them.  The rows are handed over full of garbage and compared with the call-by-call composition."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    verts, tflags = W.make_terrain()                     # one-sided heightfield, |x|, |y| <= 100, heights 0 .. 20
    eng = va.Engine(0)
    scene = va.Scene.from_tree(eng, va.HostBvh(va.tris_setup(verts, tflags)))
    stream = tp.current_stream_handle(dev)

    def loop_and_composition(start, depth, seed):
        n = len(start)
        d_start = tp.to_device(start, dev)
        d_rows = torch.full((depth * n * 16,), 0xAB, dtype=torch.uint8, device=dev)
        live = scene.bounce_loop_dev(d_start.data_ptr(), n, depth, seed, d_rows.data_ptr(), stream)
        torch.cuda.synchronize()
        rows = tp.to_host(d_rows, va.HIT).reshape(depth, n)
        d_rays, alive = d_start, n
        for d in range(depth):
            assert live[d] == alive, (d, live)
            d_hits = tp.trace_closest(scene, d_rays, n)
            hits = tp.to_host(d_hits, va.HIT)
            assert_hits_equal(rows[d], hits)
            alive = int((hits["prim"] != O_MISS).sum())
            d_attrs = tp.hit_attrs(scene, d_rays, d_hits, n)
            d_next = tp.empty_records(n, va.RAY, dev)
            eng.gen_bounce_dev(d_attrs.data_ptr(), n, seed + d, d_next.data_ptr(), stream=stream)
            d_rays = d_next
        return live

    def rays_from_below(n, seed):                        # (the stored normals of this heightfield point down: its front is its underside)
        xy = (W.uniform01(seed, 0, 2 * n).reshape(n, 2) * 160.0 - 80.0)
        jit = (W.uniform01(seed, 2 * n, 2 * n).reshape(n, 2) * 0.2 - 0.1)
        rays = np.zeros(n, dtype=va.RAY)
        rays["org"] = np.concatenate([xy, np.full((n, 1), -50.0)], 1).astype(np.float32)
        rays["dir"] = np.concatenate([jit, np.full((n, 1), 1.0)], 1).astype(np.float32)
        rays["tmax"] = np.float32(3.0e38)
        return rays

    n = 30011
    up = rays_from_below(n, 31)
    live = loop_and_composition(up, 4, 7)
    assert live[1] == n and live[2] < n                  # (a): nobody dies at depth 0, many at depth 1
    one_down = up.copy()
    one_down["dir"][12345] = (0.0, 0.0, -1.0)
    live = loop_and_composition(one_down, 3, 7)
    assert live[1] == n - 1                              # (b)
    big = 2 * 4096 * 256 + 12345                         # (c)
    live = loop_and_composition(W.sphere_rays(big, 9, origin=(3.0, -4.0, 60.0)), 3, 11)
    assert live[1] < big and live[2] < live[1]
    scene.free()
    eng.close()


def test_bounce_loop_kernel_forms_and_alpha_scene(va, O):
    """The loop's traces behind depth 0 read their ray count from device memory (trace_kernel_devn): persistent with and without
    the DMA fetch, one ray per lane, and a batch far smaller than the grid give the same rows as the call-by-call composition.
    A scene with alpha-tested triangles takes the older form (count read back per depth): same contract."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    verts, tflags = W.make_terrain()                     # open scene: paths leave the queue
    n_tri = len(verts)
    flags, attribs, mats, texels = W.alpha_test_rig(n_tri)

    def composition(eng, scene, d_start, n, depth, seed):
        rows, d_rays = [], d_start
        for d in range(depth):
            d_hits = tp.trace_closest(scene, d_rays, n)
            rows.append(tp.to_host(d_hits, va.HIT).copy())
            d_attrs = tp.hit_attrs(scene, d_rays, d_hits, n)
            d_next = tp.empty_records(n, va.RAY, dev)
            eng.gen_bounce_dev(d_attrs.data_ptr(), n, seed + d, d_next.data_ptr(), stream=tp.current_stream_handle(dev))
            d_rays = d_next
        return rows

    for alpha in (False, True):
        eng = va.Engine(0)
        tris = va.tris_setup(verts, (tflags | (flags & 2)) if alpha else tflags)
        scene = va.Scene.from_tree(eng, va.HostBvh(tris))
        if alpha:
            scene.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
            scene.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
        for n in (70001, 300):
            start = W.sphere_rays(n, 5, origin=(3.0, -4.0, 60.0))
            d_start = tp.to_device(start, dev)
            ref = None
            for cfg in (dict(persistent=1, fetch_dma=1), dict(persistent=1, fetch_dma=0), dict(persistent=0), dict(persistent=2)):
                for k, v in cfg.items():
                    eng.set_option(k, v)
                d_rows, live = tp.bounce_loop(scene, d_start, n, 4, 99)
                torch.cuda.synchronize()
                rows = tp.to_host(d_rows, va.HIT).reshape(4, n)
                if ref is None:
                    ref = composition(eng, scene, d_start, n, 4, 99)
                    assert live[0] == n and live[1] < n and live[3] <= live[2] <= live[1]
                for d in range(4):
                    assert_hits_equal(rows[d], ref[d])
                    if d:
                        assert live[d] == int((ref[d - 1]["prim"] != O_MISS).sum())
        scene.free()
        eng.close()


def test_bounce_loop_only_enqueues(va, make_bundle):
    """vt_bounce_loop_dev returns with its work in flight: behind a long-running kernel on the same stream the call comes back at
    once (round 5 waited for the live-path count once per depth), and the live counts arrive in stream order."""
    import ctypes as C
    import time
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    eng = va.Engine(0)
    scene = va.Scene.from_tree(eng, b.bvh)
    dev = torch.device("cuda", 0)
    n, depth = 200_000, 6
    d_start = tp.to_device(W.sphere_rays(n, 8, origin=(1.0, 2.0, 3.0)), dev)
    d_rows = tp.empty_records(n * depth, va.HIT, dev)
    sh = tp.current_stream_handle(dev)
    live = (C.c_uint64 * depth)()
    L = va._lib
    L.check(L.lib.vt_bounce_loop_dev(scene._h, d_start.data_ptr(), n, depth, 7, d_rows.data_ptr(), live, sh or None))   # warm-up: allocations
    torch.cuda.synchronize()
    ref = list(live)
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for _ in range(40):                                        # ~1 GiB fills: tens of milliseconds of work queued ahead of the loop
        big.fill_(1.0)
    live2 = (C.c_uint64 * depth)()
    t0 = time.perf_counter()
    L.check(L.lib.vt_bounce_loop_dev(scene._h, d_start.data_ptr(), n, depth, 7, d_rows.data_ptr(), live2, sh or None))
    t_call = time.perf_counter() - t0
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t_drain = time.perf_counter() - t1
    assert t_drain > 5 * t_call and t_call < 0.01, (t_call, t_drain)      # the call did not wait for the stream
    assert list(live2) == ref and ref[0] == n and ref[1] <= n
    scene.free()
    eng.close()


def test_reserved_cus_do_not_change_results(va, make_bundle):
    """Engine option reserved_cus (persistent grid leaves room on some CUs for a concurrent collective):
    blocks that leave take no rays with them -- results, any-hit flags and counters stay exact."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    eng = va.Engine(0)
    eng.set_option("persistent", 1)
    scene = va.Scene(eng, b.host_scene)
    rays = np.concatenate([W.primary_rays(256, 256), W.sphere_rays(70000, 9, origin=(20.0, -10.0, 35.0))])
    ref = b.oracle(rays)
    for want, limit in ((32, 2), (32, 0), (7, 1), (128, 2)):
        eng.set_option("reserved_limit", limit)
        eng.set_option("reserved_cus", want)
        got = eng.get_option("reserved_cus")
        assert 0 < got <= want and eng.get_option("reserved_limit") == limit
        assert_hits_equal(scene.trace_closest(rays), ref)
        assert (scene.trace_any(rays) == (ref["prim"] != O_MISS)).all()
        hits, st = stats_on_device(va, scene, rays[:5000])
        assert_hits_equal(hits, ref[:5000])
    eng.set_option("reserved_cus", 0)
    assert eng.get_option("reserved_cus") == 0
    assert_hits_equal(scene.trace_closest(rays), ref)
    with pytest.raises(va._lib.VisTraceError):
        eng.set_option("reserved_cus", 100000)


def test_persistent_launches_leave_their_cursors_clean(va, make_bundle):
    """A persistent launch starts from zeroed ray-block cursors and its last wave puts them back (no fill between launches):
    three times round the ring of launch slots, with batch sizes, kernel kinds and cursor modes changing from launch to
    launch, every result must stay exact."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    eng = va.Engine(0)
    eng.set_option("persistent", 1)
    scene = va.Scene(eng, b.host_scene)
    rays = np.concatenate([W.primary_rays(128, 128), W.sphere_rays(50000, 21, origin=(-30.0, 15.0, 40.0))])
    ref = b.oracle(rays)
    occ = ref["prim"] != O_MISS
    sizes = [len(rays), 1, 64, 65, 4097, 30000, 257, len(rays) - 1]
    for k in range(52):
        n = sizes[k % len(sizes)]
        eng.set_option("xcd_cursors", 1 if k % 5 == 3 else 0)
        eng.set_option("reserved_cus", 32 if k % 7 == 4 else 0)
        if k % 3 == 2:
            assert (scene.trace_any(rays[:n]) == occ[:n]).all(), f"launch {k}"
        else:
            assert_hits_equal(scene.trace_closest(rays[:n]), ref[:n])
    hits, st = stats_on_device(va, scene, rays)
    assert_hits_equal(hits, ref)


@pytest.mark.parametrize("persistent", [0, 1])
def test_ray_image_width_changes_no_result(va, make_bundle, persistent):
    """Engine option ray_image_width: lanes take 4 x 16 pixel tiles of an image-order batch instead of consecutive rays.  Scheduling
    only -- hits, any-hit flags and the per-ray counters must stay exact for row lengths that are multiples of 8, of 4 only, of
    neither (hint ignored), for images whose height is no multiple of 16 (the rest is taken in order) and for batches that are no
    image at all."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    eng = va.Engine(0)
    eng.set_option("persistent", persistent)
    scene = va.Scene(eng, b.host_scene)
    for width, height in ((256, 256), (100, 37), (36, 50), (255, 64), (8, 16), (4, 15)):
        rays = W.primary_rays(width, height)
        ref = b.oracle(rays)
        eng.set_option("ray_image_width", width)
        assert eng.get_option("ray_image_width") == width
        assert_hits_equal(scene.trace_closest(rays), ref)
        assert (scene.trace_any(rays) == (ref["prim"] != O_MISS)).all()
        hits, st = stats_on_device(va, scene, rays)
        assert_hits_equal(hits, ref)
        eng.set_option("ray_image_width", 0)
        _, st0 = stats_on_device(va, scene, rays)
        assert (st["steps"] == st0["steps"]).all() and (st["tests"] == st0["tests"]).all()
    rays = W.sphere_rays(70001, 77, origin=(5.0, -20.0, 30.0))         # not an image: any multiple of 4 is still harmless
    eng.set_option("ray_image_width", 128)
    assert_hits_equal(scene.trace_closest(rays), b.oracle(rays))
    eng.set_option("ray_image_width", 0)


def test_host_buffer_pipeline_ragged(va, engine, make_bundle):
    """vt_trace_closest / vt_trace_any with caller (pageable) buffers above the pipelining threshold and a ragged
    last chunk: identical to the device-resident path, and to the oracle on a sample."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    n = 3 * (1 << 20) + 12345
    rays = W.sphere_rays(n, 123, origin=(-40.0, 10.0, 25.0))
    got = scene.trace_closest(rays)
    occ = scene.trace_any(rays)
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    ref = tp.to_host(tp.trace_closest(scene, d_rays, n), va.HIT)
    assert_hits_equal(got, ref)
    assert (occ == (ref["prim"] != O_MISS)).all()
    idx = np.concatenate([np.arange(0, 5000), np.arange((1 << 20) - 2500, (1 << 20) + 2500), np.arange(n - 5000, n)])
    assert_hits_equal(got[idx], b.oracle(rays[idx]))
    again = scene.trace_closest(rays[: 2 * (1 << 20) + 1])          # a second call re-uses the staging buffers
    assert_hits_equal(again, ref[: 2 * (1 << 20) + 1])


def test_alpha_test_at_its_thresholds(va, engine, O):
    """Known answers of the alpha comparison (Primitives.h:205: the hit is dropped when alpha < alphaRef, so alpha == alphaRef
    PASSES) at its branch points, which random textures do not reach (scripts/mutants.sh 61 / 65 / 66 survived them): texel a
    against a reference between a / 256 and a / 255, alpha exactly equal to the reference, and a material without a texture
    (alpha 1) under references on both sides of 1.  One flagged triangle per case above an unflagged floor: a dropped hit shows
    as the floor's index."""
    cases = [  # (texel or None = no texture, alphaRef, passes)
        (255, 1.0, True), (0, 0.0, True), (179, 0.7, True), (178, 0.7, False), (128, 0.5, True), (127, 0.5, False),
        (None, 1.0, True), (None, 1.5, False), (None, 0.0, True), (1, 0.003921569, True), (254, 1.0, False),
    ]
    k = len(cases)
    verts = np.zeros((k + 1, 3, 3), np.float32)
    for i in range(k):
        verts[i] = [[4.0 * i, 0.0, 1.0], [4.0 * i + 3.0, 0.0, 1.0], [4.0 * i, 3.0, 1.0]]
    verts[k] = [[-100.0, -100.0, 0.0], [500.0, -100.0, 0.0], [-100.0, 500.0, 0.0]]           # the floor: no alpha test
    flags = np.full(k + 1, 2, np.uint8); flags[k] = 0
    from vistrace_amd import workloads as W
    _, attribs, mats, _ = W.alpha_test_rig(k + 1, nmats=k)
    attribs["uv"] = 0.25
    attribs["material"] = np.arange(k + 1) % k
    texels = []
    for i, (a, ref, _) in enumerate(cases):
        mats["tex_mat"][i] = [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]]
        mats["tex_scale"][i] = 1.0
        mats["alpha_ref"][i] = np.float32(ref)
        mats["width"][i] = mats["height"][i] = 0 if a is None else 1
        mats["filter"][i] = 0
        mats["offset"][i] = len(texels)
        if a is not None:
            texels.append(a)
    texels = np.array(texels, np.uint8)
    tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris)
    scene = va.Scene(engine, va.HostScene(bvh))
    scene.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
    scene.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
    rays = np.zeros(k, va.RAY)
    rays["org"] = [[4.0 * i + 0.75, 0.75, 5.0] for i in range(k)]
    rays["dir"] = (0.0, 0.0, -1.0)
    rays["tmax"] = np.float32(100.0)
    otris = O.tris_from_tri64(tris)
    try:
        O.set_alpha(otris, attribs["uv"].reshape(k + 1, 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)[0]
    finally:
        O.set_alpha()
    expect = np.array([i if ok else k for i, (_, _, ok) in enumerate(cases)], np.uint32)
    assert (ref["prim"] == expect).all(), (ref["prim"], expect)          # the oracle agrees with the reading of :205
    saved = {key: engine.get_option(key) for key in ("persistent", "fetch_dma")}
    try:
        for cfg in (dict(persistent=1, fetch_dma=1), dict(persistent=1, fetch_dma=0), dict(persistent=0)):
            for key, v in cfg.items():
                engine.set_option(key, v)
            got = scene.trace_closest(rays)
            assert (got["prim"] == expect).all(), (cfg, got["prim"], expect)
            assert_hits_equal(got, ref)
            assert_hits_equal(stats_on_device(va, scene, rays)[0], ref)
            assert (scene.trace_any(rays) == 1).all()                    # (the floor is always there)
    finally:
        for key, v in saved.items():
            engine.set_option(key, v)
    scene.free()


def test_alpha_test_in_kernel(va, engine, O):
    """Primitives.h:196-208 on the device (ALPHA kernel variants): texUV, TransformTexcoord, the alpha plane lookup
    defined in include/vistrace_hip.h and the reference comparison -- hits, any-hit flags and counters equal the
    oracle's; tracing such a scene without its side data fails loudly."""
    from vistrace_amd import workloads as W
    verts = W.make_scene("S10k")
    n = len(verts)
    flags, attribs, mats, texels = W.alpha_test_rig(n)
    tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris)
    scene = va.Scene(engine, va.HostScene(bvh))
    rays = np.concatenate([W.primary_rays(96, 96), W.sphere_rays(30000, 41, origin=(-120.0, 80.0, 15.0))])
    with pytest.raises(va._lib.VisTraceError) as err:
        scene.trace_closest(rays[:10])
    assert err.value.code == va._lib.VT_ERR_UNSUPPORTED
    scene.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
    scene.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
    otris = O.tris_from_tri64(tris)
    try:
        plain = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)[0]
        O.set_alpha(otris, attribs["uv"].reshape(n, 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref, ref_st, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, want_stats=True)
        any_ref = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, any_hit=True)[0]
    finally:
        O.set_alpha()
    assert 500 < int((plain["prim"] != ref["prim"]).sum())               # the test changes a good share of the results
    assert_hits_equal(scene.trace_closest(rays), ref)
    assert (scene.trace_any(rays) == (any_ref["prim"] != O_MISS)).all()
    got, st = stats_on_device(va, scene, rays)
    assert_hits_equal(got, ref)
    assert (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()
    assert_hits_equal(scene.trace_closest(rays[:200]), ref[:200])        # tiny host batch, one ray per lane
    # every kernel form runs the alpha test as lane states (parked candidate -> AlphaRec -> texels): same bytes
    saved = {k: engine.get_option(k) for k in ("persistent", "fetch_dma")}
    try:
        for cfg in (dict(persistent=0), dict(persistent=1, fetch_dma=0), dict(persistent=1, fetch_dma=1)):
            for k, v in cfg.items():
                engine.set_option(k, v)
            assert_hits_equal(scene.trace_closest(rays), ref)
            assert (scene.trace_any(rays) == (any_ref["prim"] != O_MISS)).all()
    finally:
        for k, v in saved.items():
            engine.set_option(k, v)
    # camera rays declared an image (ray_image_width): the one-ray-per-lane ALPHA kernel walks them as 4 x 16 tiles (round 5) and
    # the engine's auto rule picks it -- and every alpha_threshold (how many parked candidates the AlphaRec block waits for) gives
    # the same bytes and counters
    cam = W.primary_rays(128, 96)
    try:
        O.set_alpha(otris, attribs["uv"].reshape(n, 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        cam_ref, cam_st = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, cam, want_stats=True)[:2]
    finally:
        O.set_alpha()
    saved = {k: engine.get_option(k) for k in ("ray_image_width", "alpha_threshold", "persistent")}
    try:
        for width, thr, mode in ((128, 4, saved["persistent"]), (128, 1, 2), (64, 16, 0), (128, 64, 1), (0, 2, 1)):
            engine.set_option("ray_image_width", width); engine.set_option("alpha_threshold", thr); engine.set_option("persistent", mode)
            assert_hits_equal(scene.trace_closest(cam), cam_ref)
            got, st = stats_on_device(va, scene, cam)
            assert_hits_equal(got, cam_ref)
            assert (st["steps"] == cam_st[:, 0]).all() and (st["tests"] == cam_st[:, 1]).all()
            assert_hits_equal(scene.trace_closest(rays), ref)
    finally:
        for k, v in saved.items():
            engine.set_option(k, v)
    # materials beyond the table: the reference's test is skipped for them, the geometric hit stands
    attribs2 = attribs.copy()
    attribs2["material"][::3] = len(mats) + 5
    scene.set_tri_attribs(attribs2.view(va.TRI_ATTRIBS))
    try:
        O.set_alpha(otris, attribs2["uv"].reshape(n, 6), attribs2["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref2 = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)[0]
    finally:
        O.set_alpha()
    assert int((ref2["prim"] != ref["prim"]).sum()) > 100
    assert_hits_equal(scene.trace_closest(rays), ref2)
    scene.free()
    # a scene uploaded WITHOUT flagged triangles whose refit switches the test on (the record array grows once)
    tris0 = va.tris_setup(verts)
    bvh0 = va.HostBvh(tris0)
    scene0 = va.Scene(engine, va.HostScene(bvh0))
    scene0.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
    scene0.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
    assert_hits_equal(scene0.trace_closest(rays), plain)
    scene0.refit(verts, flags)
    otris0 = O.tris_from_tri64(tris0)
    otris0["flags"] = flags
    try:
        O.set_alpha(otris0, attribs["uv"].reshape(n, 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref0 = O.traverse_batch(bvh0.nodes().view(O.NODE), bvh0.prim_indices(), otris0, rays)[0]
    finally:
        O.set_alpha()
    assert_hits_equal(scene0.trace_closest(rays), ref0)
    scene0.free()


def test_scene_outliving_its_engine_is_inert(va, make_bundle):
    """Closing an engine releases the device memory of its scenes; a scene object that is still referenced
    afterwards reports the closed engine instead of touching freed state, and can be freed safely."""
    from vistrace_amd import workloads as W
    b = make_bundle("S1k")
    eng = va.Engine(0)
    scene = va.Scene(eng, b.host_scene)
    rays = W.sphere_rays(500, 2)
    assert_hits_equal(scene.trace_closest(rays), b.oracle(rays))
    eng.close()
    with pytest.raises(va._lib.VisTraceError) as err:
        scene.trace_closest(rays)
    assert "closed" in str(err.value)
    scene.free()


def test_refit_with_non_finite_vertices_is_refused(va, engine, make_bundle):
    """A refit / skinning pass that produces NaN or infinite vertices would leave NaN boxes that every ray walks
    into (the slab test ignores NaN terms): the call fails loudly and the scene refuses to trace until a clean refit."""
    from vistrace_amd import workloads as W
    verts = W.make_scene("S1k")
    n = len(verts)
    scene = va.Scene(engine, va.HostScene(va.HostBvh(va.tris_setup(verts))))
    rays = W.sphere_rays(2000, 4, origin=(5.0, 5.0, 5.0))
    ref = scene.trace_closest(rays)
    bad = verts.copy(); bad[17, 1, 2] = np.nan; bad[400:410] = np.inf
    with pytest.raises(va._lib.VisTraceError) as err:
        scene.refit(bad)
    assert "11 triangles" in str(err.value)
    with pytest.raises(va._lib.VisTraceError):
        scene.trace_closest(rays)
    scene.refit(verts)                                           # a clean refit heals the scene
    assert_hits_equal(scene.trace_closest(rays), ref)
    skin, base, nmat = W.skinned_rig(n)
    scene.set_skin(verts, skin, base)
    bones, binds = W.rig_pose(nmat, 0)
    bones_bad = bones.copy(); bones_bad[3, 5] = np.nan
    with pytest.raises(va._lib.VisTraceError):
        scene.skin_refit(bones_bad, binds)
    with pytest.raises(va._lib.VisTraceError):
        scene.trace_any(rays)
    scene.skin_refit(bones, binds)
    scene.trace_closest(rays)


def test_page_locked_host_arrays_skip_the_staging(va, engine, make_bundle):
    """vt_host_register: host arrays the caller has page-locked are read and written in place by the copy engines (no staging
    copies); same bytes as the staged path, for a ragged batch above the pipelining threshold, closest hit and any hit."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = upload(va, engine, b)
    n = (1 << 21) + (1 << 20) + 4321                                          # > 2 chunks of 1 Mi rays, ragged tail
    rays = np.ascontiguousarray(np.concatenate([W.primary_rays(1024, 1024), W.sphere_rays(n - (1 << 20), 31, origin=(1.0, 2.0, 3.0))]))
    staged = scene.trace_closest(rays)
    occ_staged = scene.trace_any(rays)
    hits = np.zeros(n, va.HIT)
    va.host_register(rays)
    va.host_register(hits)
    try:
        scene.trace_closest(rays, out=hits)
        assert hits.tobytes() == staged.tobytes()
        assert (scene.trace_any(rays) == occ_staged).all()                     # rays locked, result array not: staged path again
    finally:
        va.host_unregister(rays)
        va.host_unregister(hits)
    assert_hits_equal(staged[:30000], b.oracle(rays[:30000]))
    scene.free()


@pytest.mark.parametrize("with_alpha", [False, True])
def test_records_beyond_4_gib(va, engine, O, with_alpha):
    """Maximum sizes: a scene whose triangle records (and AlphaRecs) lie beyond 4 GiB.  Building one takes 45 M triangles; the
    upload's test hook VT_TEST_RECORD_GAP leaves an unused gap of 2^26 records (4 GiB) between the pairs and the triangles of a
    10 k-triangle scene instead, which moves every triangle fetch, the refit's record writes, the TraceResult kernels' reads and
    the AlphaRec table past the 32-bit byte offsets the DMA-fetch kernel uses -- so the engine must pick the kernels that form
    64-bit addresses, and their results must not change: hits, any-hit, counters, hit attributes, device refit."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    verts = W.make_scene("S10k")
    n = len(verts)
    flags = None
    if with_alpha:
        flags, attribs, mats, texels = W.alpha_test_rig(n)
    tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris)
    host_scene = va.HostScene(bvh)
    os.environ["VT_TEST_RECORD_GAP"] = str((1 << 26) + 1000)
    try:
        hooks_were_on = os.environ.get("VT_ENABLE_TEST_HOOKS") == "1"      # (the group-fixture run of this suite has them on)
        if not hooks_were_on:
            small = va.Scene(engine, host_scene)      # the hook alone is dead: it needs VT_ENABLE_TEST_HOOKS=1 beside it
            assert small.device_bytes < (1 << 30)
            small.free()
            os.environ["VT_ENABLE_TEST_HOOKS"] = "1"
        scene = va.Scene(engine, host_scene)
    finally:
        del os.environ["VT_TEST_RECORD_GAP"]
        if not hooks_were_on:
            os.environ.pop("VT_ENABLE_TEST_HOOKS", None)
    assert scene.device_bytes > (1 << 32)
    otris = O.tris_from_tri64(tris)
    rays = np.concatenate([W.primary_rays(96, 96), W.sphere_rays(30000, 41, origin=(-120.0, 80.0, 15.0))])
    try:
        if with_alpha:
            scene.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
            scene.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
            O.set_alpha(otris, attribs["uv"].reshape(n, 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        ref, ref_st, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, want_stats=True)
        any_ref = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, any_hit=True)[0]
    finally:
        O.set_alpha()
    saved = {k: engine.get_option(k) for k in ("persistent", "fetch_dma")}
    try:
        for cfg in (dict(persistent=0), dict(persistent=1, fetch_dma=1), dict(persistent=2)):
            for k, v in cfg.items():
                engine.set_option(k, v)
            assert_hits_equal(scene.trace_closest(rays), ref)
            if cfg.get("persistent") == 1:
                assert engine.get_option("last_persistent") == 1 and engine.get_option("last_fetch_dma") == 0, "DMA fetch beyond 4 GiB"
            assert (scene.trace_any(rays) == (any_ref["prim"] != O_MISS)).all()
            got, st = stats_on_device(va, scene, rays)
            assert_hits_equal(got, ref)
            assert (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()
    finally:
        for k, v in saved.items():
            engine.set_option(k, v)
    # the records are where the gap says, and the TraceResult kernel reads them there
    pairs, dtris = scene.read_records()
    assert (dtris.view(np.uint8) == host_scene.tris().view(np.uint8)).all() and (pairs.view(np.uint8) == host_scene.pairs().view(np.uint8)).all()
    dev = torch.device("cuda", 0)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.trace_closest(scene, d_rays, len(rays))
    attrs = tp.to_host(tp.hit_attrs(scene, d_rays, d_hits, len(rays)), va.HIT_ATTRS)
    oa = O.hit_attrs(otris, rays, ref)
    hit = ref["prim"] != O_MISS
    assert np.allclose(attrs["pos"][hit], oa["pos"][hit], rtol=REL_TOL, atol=1e-4) and (attrs["hit"] == hit).all()
    if not with_alpha:
        # device refit writes the moved triangles' records beyond 4 GiB too
        moved = (verts + np.float32(0.125)).astype(np.float32)
        scene.refit(moved)
        mtris = va.tris_setup(moved)
        bvh.refit(mtris)
        ref2 = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(mtris), rays)[0]
        assert_hits_equal(scene.trace_closest(rays), ref2)
    scene.free()
