"""Merged launches (vt_trace_closest_multi_dev / vt_trace_any_multi_dev): several batches, one grid start, one drain.

Bar: the results of a merged launch are BYTE-EQUAL to those of one call per batch (which the other GPU tests pin to the
oracle), and equal to the oracle directly -- for ragged batch sizes including 0 and 1, batches in image order beside
batches that are not, result arrays scattered over one allocation or over several, both kernels (persistent waves and
one ray per lane) and the alpha-test variants.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

O_MISS = 0xFFFFFFFF


def assert_hits_equal(got, ref):
    assert (got["prim"] == ref["prim"]).all(), f"{int((got['prim'] != ref['prim']).sum())} primitive indices differ"
    for k in ("t", "u", "v"):
        assert (got[k].view(np.uint32) == ref[k].view(np.uint32)).all(), f"{k} not bit-identical"


def ray_sets(W, sizes, seed=3):
    """one ray set per size: camera images where the size is a whole image, scattered rays otherwise"""
    out = []
    for k, n in enumerate(sizes):
        side = int(round(n ** 0.5))
        if n >= 256 and side * side == n:
            out.append((W.primary_rays(side, side, pos=(float(3 * k), -2.0 * k, 1.0 * k)), side))
        else:
            out.append((W.sphere_rays(n, seed + k, origin=(10.0 * (k % 3), -5.0 * k, 7.0)), 0))
    return out


@pytest.mark.parametrize("sizes", [
    (4096, 0, 1, 65, 1024, 30000, 63, 16384, 0),          # ragged, with empty batches in front of, between and behind others
    (64,) * 40,                                            # many tiny batches: more batches than some grids have waves
    (1,),                                                  # a "merged" launch of one batch of one ray
    (0, 0, 0),                                             # nothing at all
    (65536,) * 16,                                         # configs[1] cut into 16 segments of 64 Ki rays
])
def test_merged_launch_equals_separate_calls(va, engine, make_bundle, sizes):
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    b = make_bundle("S10k")
    scene = va.Scene(engine, b.host_scene)
    sets = ray_sets(W, sizes)
    d_rays = [tp.to_device(r, dev) if len(r) else None for r, _ in sets]
    # separate calls (engine option for the image hint), each into its own array
    sep = []
    for (r, width), d in zip(sets, d_rays):
        if len(r) == 0:
            sep.append(np.zeros(0, va.HIT))
            continue
        engine.set_option("ray_image_width", width)
        sep.append(tp.to_host(tp.trace_closest(scene, d, len(r)), va.HIT))
    engine.set_option("ray_image_width", 0)
    torch.cuda.synchronize()
    # merged: results scattered over ONE allocation in reverse batch order (offsets relative to batch 0 may be negative)
    total = sum(len(r) for r, _ in sets)
    d_all = torch.full(((total + 16) * 16,), 0xAB, dtype=torch.uint8, device=dev)
    offs, cur = [], total
    for r, _ in sets:
        cur -= len(r)
        offs.append(cur)
    stream = tp.current_stream_handle(dev)
    batches = [(d.data_ptr() if d is not None else 0, d_all.data_ptr() + 16 * o if len(r) else 0, len(r), width)
               for (r, width), d, o in zip(sets, d_rays, offs)]
    scene.trace_multi_dev(batches, stream)
    torch.cuda.synchronize()
    got_all = tp.to_host(d_all, va.HIT)
    for k, ((r, _), o) in enumerate(zip(sets, offs)):
        got = got_all[o:o + len(r)]
        assert got.tobytes() == sep[k].tobytes(), f"batch {k} of {len(sets)} differs from its separate call"
        if len(r) and k < 6:
            assert_hits_equal(got, b.oracle(r))
    assert (tp.to_host(d_all, np.uint8)[total * 16:] == 0xAB).all(), "a merged launch wrote outside its batches"
    # any-hit, results in separate allocations
    d_occ = [torch.full((max(len(r), 1),), 7, dtype=torch.uint8, device=dev) for r, _ in sets]
    scene.trace_multi_dev([(d.data_ptr() if d is not None else 0, o.data_ptr(), len(r), width)
                           for (r, width), d, o in zip(sets, d_rays, d_occ)], stream, any_hit=True)
    torch.cuda.synchronize()
    for k, ((r, _), o) in enumerate(zip(sets, d_occ)):
        if len(r):
            assert (o.cpu().numpy()[:len(r)] == (sep[k]["prim"] != O_MISS)).all(), f"any-hit batch {k}"


def test_merged_launch_validates_its_arguments(va, engine, make_bundle):
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    scene = va.Scene(engine, make_bundle("S1k").host_scene)
    rays = W.sphere_rays(1000, 5)
    d_rays = tp.to_device(rays, dev)
    d_hits = tp.empty_records(2000, va.HIT, dev)
    p, h = d_rays.data_ptr(), d_hits.data_ptr()
    E = va._lib.VisTraceError
    with pytest.raises(E, match="overlap"):
        scene.trace_multi_dev([(p, h, 1000), (p, h + 16 * 500, 1000)])
    with pytest.raises(E, match="aligned"):
        scene.trace_multi_dev([(p, h, 500), (p, h + 16 * 500 + 8, 500)])
    with pytest.raises(E, match="NULL"):
        scene.trace_multi_dev([(p, h, 500), (0, h + 16 * 500, 500)])
    desc = np.zeros(1, va._lib.BATCH_DESC)
    desc["d_rays"], desc["d_out"], desc["n"], desc["reserved"] = p, h, 10, 1
    assert va._lib.lib.vt_trace_closest_multi_dev(scene._h, va._lib.ptr(desc), 1, None) == va._lib.VT_ERR_INVALID_ARG
    scene.trace_multi_dev([])                                   # no batches: nothing to do
    scene.trace_multi_dev([(p, h, 1000), (p, h + 16 * 1000, 1000)], tp.current_stream_handle(dev))   # the same rays twice is fine
    torch.cuda.synchronize()
    got = tp.to_host(d_hits, va.HIT)
    assert got[:1000].tobytes() == got[1000:].tobytes()


def test_merged_launch_round_the_slot_ring(va, make_bundle):
    """40 merged launches with changing batch tables (the table travels through the slot's pinned block, which must not be
    rewritten while an earlier launch on that slot may still read it), alternating with plain launches."""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    b = make_bundle("S10k")
    eng = va.Engine(0)
    eng.set_option("persistent", 1)
    scene = va.Scene(eng, b.host_scene)
    rays = W.sphere_rays(60000, 11, origin=(-30.0, 15.0, 40.0))
    ref = b.oracle(rays)
    d_rays = tp.to_device(rays, dev)
    stream = tp.current_stream_handle(dev)
    rng = np.random.default_rng(5)
    outs = []
    for it in range(40):
        cuts = np.sort(rng.choice(np.arange(1, len(rays)), size=int(rng.integers(1, 90)), replace=False))
        bounds = np.concatenate([[0], cuts, [len(rays)]])
        d_hits = torch.zeros(len(rays) * 16, dtype=torch.uint8, device=dev)
        batches = [(d_rays.data_ptr() + 32 * int(lo), d_hits.data_ptr() + 16 * int(lo), int(hi - lo)) for lo, hi in zip(bounds[:-1], bounds[1:])]
        scene.trace_multi_dev(batches, stream)
        outs.append(d_hits)
        if it % 4 == 1:
            outs.append(tp.trace_closest(scene, d_rays, len(rays)))
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        assert_hits_equal(tp.to_host(o, va.HIT), ref), f"launch {k}"


def test_merged_launch_with_alpha_tested_triangles(va, engine, O):
    """the ALPHA kernel variants read their batches through the same table"""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    verts = W.make_scene("S1k")
    n = len(verts)
    rng = np.random.default_rng(12)
    flags = np.where(rng.random(n) < 0.4, va._lib.VT_TRI_ALPHATEST, 0).astype(np.uint8)
    tris = va.tris_setup(verts, flags)
    scene = va.Scene(engine, va.HostScene(va.HostBvh(tris)))
    attribs = np.zeros(n, va.TRI_ATTRIBS)
    attribs["uv"] = rng.random((n, 3, 2)).astype(np.float32) * 4
    attribs["material"] = rng.integers(0, 2, n)
    mats = np.zeros(2, va.ALPHA_MATERIAL)
    mats["tex_mat"][:, 0, 0] = 1
    mats["tex_mat"][:, 1, 1] = 1
    mats["tex_scale"], mats["alpha_ref"] = 1.0, 0.5
    mats["width"], mats["height"], mats["filter"] = (8, 16), (8, 4), (0, 1)
    mats["offset"] = (0, 64)
    texels = rng.integers(0, 256, 128).astype(np.uint8)
    scene.set_tri_attribs(attribs)
    scene.set_alpha(mats, texels)
    rays = np.concatenate([W.primary_rays(64, 64), W.sphere_rays(9000, 2)])
    d_rays = tp.to_device(rays, dev)
    whole = tp.to_host(tp.trace_closest(scene, d_rays, len(rays)), va.HIT)
    d_hits = tp.empty_records(len(rays), va.HIT, dev)
    cuts = [0, 1, 700, 4096, 4097, 9000, len(rays)]
    scene.trace_multi_dev([(d_rays.data_ptr() + 32 * lo, d_hits.data_ptr() + 16 * lo, hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])],
                          tp.current_stream_handle(dev))
    torch.cuda.synchronize()
    assert tp.to_host(d_hits, va.HIT).tobytes() == whole.tobytes()


def test_batch_set_equals_single_batches(va, engine, make_bundle):
    """vt_batch_trace_closest_set: host ray arrays -> staged uploads -> ONE merged launch -> one vt_batch each, equal to the batches
    traced one by one (hits, attrs, rays), with images beside scattered rays, empty and one-ray batches, the range checks naming
    batch and ray, and blocks recycled across calls."""
    from vistrace_amd import workloads as W
    b = make_bundle("S10k")
    scene = va.Scene(engine, b.host_scene)
    sets = ray_sets(W, (4096, 0, 1, 300, 65536, 12345, 1024))
    arrs, widths = [r for r, _ in sets], [w for _, w in sets]
    singles = [scene.trace_batch(r) for r in arrs]
    for rep in range(3):
        got = scene.trace_batch_set(arrs, image_widths=widths, check_ranges=True, fetch_hits=rep != 1)
        assert len(got) == len(arrs)
        for k, (g, one) in enumerate(zip(got, singles)):
            assert len(g) == len(arrs[k])
            assert g.hits().tobytes() == one.hits().tobytes(), f"batch {k}"
            assert g.attrs().tobytes() == one.attrs().tobytes(), f"batch {k}"
            assert g.rays().tobytes() == arrs[k].tobytes(), f"batch {k}"
        for g in reversed(got):
            g.free()
    assert_hits_equal(singles[4].hits(), b.oracle(arrs[4]))
    bad = [a.copy() for a in arrs]
    bad[5]["tmax"][77] = -3.0
    bad[6]["tmin"][0] = -1.0
    with pytest.raises(ValueError, match="batch 5, ray 77"):
        scene.trace_batch_set(bad, check_ranges=True)
    assert scene.trace_batch_set([]) == []
    assert len(scene.trace_batch_set([arrs[0][:0]])[0]) == 0
    scene.free()


def test_merged_launch_on_degenerate_scenes(va, engine):
    """empty scene (every ray misses), a scene whose root is a leaf (no node step at all) and one triangle: merged launches give what
    separate calls give, for closest hit and any hit"""
    import torch
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    rays = W.sphere_rays(5000, 17, origin=(0.3, 0.2, -4.0))
    rays["dir"][:2500] = (0.0, 0.0, 1.0)
    d_rays = tp.to_device(rays, dev)
    tri = np.array([[[-1, -1, 0], [1, -1, 0], [0, 1, 0]]], np.float32)
    quad = np.array([[[-1, -1, 0], [1, -1, 0], [1, 1, 0]], [[-1, -1, 0], [1, 1, 0], [-1, 1, 0]], [[-1, -1, 1], [1, -1, 1], [1, 1, 1]]], np.float32)
    for verts in (np.zeros((0, 3, 3), np.float32), tri, quad):
        tris = va.tris_setup(verts)
        scene = va.Scene(engine, va.HostScene(va.HostBvh(tris)))
        whole = tp.to_host(tp.trace_closest(scene, d_rays, len(rays)), va.HIT)
        occ = tp.trace_any(scene, d_rays, len(rays)).cpu().numpy()
        d_hits = tp.empty_records(len(rays), va.HIT, dev)
        d_occ = torch.zeros(len(rays), dtype=torch.uint8, device=dev)
        cuts = [0, 7, 7, 2500, 2501, 5000]
        scene.trace_multi_dev([(d_rays.data_ptr() + 32 * lo, d_hits.data_ptr() + 16 * lo, hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])],
                              tp.current_stream_handle(dev))
        scene.trace_multi_dev([(d_rays.data_ptr() + 32 * lo, d_occ.data_ptr() + lo, hi - lo) for lo, hi in zip(cuts[:-1], cuts[1:])],
                              tp.current_stream_handle(dev), any_hit=True)
        torch.cuda.synchronize()
        assert tp.to_host(d_hits, va.HIT).tobytes() == whole.tobytes()
        assert (d_occ.cpu().numpy() == occ).all()
        if len(verts):
            assert (whole["prim"][:2500] != O_MISS).sum() > 0            # the parallel rays do hit the triangle(s)
        else:
            assert (whole["prim"] == O_MISS).all()
        scene.free()


def test_batch_set_of_empty_buffers_on_a_fresh_engine(va, make_bundle):
    """accel:TraverseBatch({""}) as an accel's FIRST call: a set of empty buffers only, with the hit records asked for, on an engine
    that has never traced a host batch (its staging pipeline and events do not exist yet) -- empty batches come back, no HIP error."""
    b = make_bundle("S10k")
    eng = va.Engine(0)
    scene = va.Scene(eng, b.host_scene)
    empty = np.zeros(0, va.RAY)
    got = scene.trace_batch_set([empty, empty], check_ranges=True, fetch_hits=True)
    assert [len(g) for g in got] == [0, 0] and len(got[0].hits()) == 0 and len(got[1].attrs()) == 0
    for g in got:
        g.free()
    scene.free()
    eng.close()


def test_open_batch_set_survives_its_engine_and_its_scene(va, make_bundle):
    """A set that is still open (buffers added, not traced) when its engine is closed or its scene freed: the batches' device memory
    goes with the engine, later calls on the set fail with a message, and abort / trace free the shells (no dangling pointers: this
    test crashes or trips the allocator otherwise)."""
    import ctypes as C
    from vistrace_amd import workloads as W
    L = va._lib
    b = make_bundle("S10k")
    rays = np.ascontiguousarray(W.sphere_rays(5000, 3))
    bad = C.c_uint64(0)
    for how in ("close_engine_then_trace", "close_engine_then_abort", "free_scene_then_trace", "free_scene_then_add"):
        eng = va.Engine(0)
        scene = va.Scene(eng, b.host_scene)
        h = C.c_void_p()
        assert L.lib.vt_batch_set_begin(scene._h, 0, C.byref(h)) == L.VT_OK
        assert L.lib.vt_batch_set_add(h, L.ptr(rays), len(rays), 0, C.byref(bad)) == L.VT_OK
        assert L.lib.vt_batch_set_add(h, L.ptr(rays), len(rays), 0, C.byref(bad)) == L.VT_OK
        outs = (C.c_void_p * 2)()
        if how.startswith("close_engine"):
            eng.close()
            if how.endswith("trace"):
                assert L.lib.vt_batch_set_trace(h, outs) == L.VT_ERR_INVALID_ARG and b"engine has been closed" in L.lib.vt_last_error()
            else:
                L.lib.vt_batch_set_abort(h)
            scene.free()
        else:
            scene.free()
            if how.endswith("add"):
                assert L.lib.vt_batch_set_add(h, L.ptr(rays), len(rays), 0, C.byref(bad)) == L.VT_ERR_INVALID_ARG
                assert b"scene has been freed" in L.lib.vt_last_error()
                L.lib.vt_batch_set_abort(h)
            else:
                assert L.lib.vt_batch_set_trace(h, outs) == L.VT_ERR_INVALID_ARG and b"scene has been freed" in L.lib.vt_last_error()
            eng.close()
    # and the engine is as usable as before
    eng = va.Engine(0)
    scene = va.Scene(eng, b.host_scene)
    got = scene.trace_batch_set([rays])
    assert_hits_equal(got[0].hits(), b.oracle(rays))
    got[0].free()
    scene.free()
    eng.close()
