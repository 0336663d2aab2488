#!/usr/bin/env python3
"""Long randomized parity soak on the GPU: many seeded triangle soups (degenerate / duplicate triangles, cull
flags, extreme scales) x random rays (windows, zero components), device vs oracle bit for bit incl. counters,
for the persistent, static and auto kernels.  Usage: python scripts/soak_parity.py [first_seed] [count] [seconds]
(stops cleanly after `seconds`, prints a progress line every 100 seeds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vistrace_amd as va
from oracle import binding as O
from vistrace_amd import torch_plumbing as tp

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else float("inf")
eng = va.Engine(0)
dev = torch.device("cuda", 0)
bad = 0
t0 = time.time()
done = 0
for seed in range(first, first + count):
    if time.time() - t0 > budget:
        break
    if done and done % 100 == 0:
        print(f"soak: {done} seeds done, {bad} mismatches, {time.time() - t0:.0f}s", flush=True)
    done += 1
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 20000)) if seed % 7 else int(rng.integers(20000, 90000))   # the larger ones reach the SAH builder's parallel subtree tasks
    spread = 10.0 ** rng.uniform(0, 3)
    centre = rng.uniform(-spread, spread, (n, 1, 3))
    scale = 10.0 ** rng.uniform(-4, 2, (n, 1, 1))
    verts = (centre + rng.normal(size=(n, 3, 3)) * scale).astype(np.float32)
    if n > 40:
        verts[::17, 1] = verts[::17, 0]
        verts[3::29] = verts[2::29][: len(verts[3::29])]
    flags = (rng.random(n) < rng.random()).astype(np.uint8)
    rig = None
    if seed % 4 == 0:                                         # alpha-tested triangles: ALPHA kernel variants
        from vistrace_amd import workloads as W
        rig = W.alpha_test_rig(n, nmats=int(rng.integers(1, 9)), alpha_fraction=float(rng.random()), seed=seed)
        flags = flags | rig[0]
        rig[1]["uv"][::11] *= np.float32(1e6)                 # absurd texture coordinates
        rig[1]["uv"][5::23, 0, 0] = np.nan
    tris = va.tris_setup(verts, flags)
    poisoned = seed % 11 == 0 and n > 50
    if poisoned:                             # a few non-finite triangles: never hit, must not derail a builder
        verts[1::max(2, n // 7), rng.integers(0, 3), rng.integers(0, 3)] = [np.nan, np.inf, -np.inf][seed % 3]
        tris = va.tris_setup(verts, flags)
    bvh = va.HostBvh(tris, nthreads=int(rng.integers(1, 9)), builder=("sah", "ploc", "sah_refined")[seed % 3])
    host_scene = va.HostScene(bvh)
    if seed % 2 == 0:                                         # round 5: the tree re-packed ON THE DEVICE (vt_scene_upload_tree) -- the records
        scene = va.Scene.from_tree(eng, bvh)                  # must equal the host lineariser's byte for byte, and everything below runs on them
        dp, dt = scene.read_records()                         # (fetches the host copy back: vt_host_scene_download)
        if not (dp.tobytes() == host_scene.pairs().tobytes() and dt.tobytes() == host_scene.tris().tobytes()
                and scene.host_scene.trace_closest_host(np.zeros(0, va.RAY)).shape == (0,)):
            bad += 1
            print(f"DEVICE RE-PACK MISMATCH seed {seed} n {n}", flush=True)
    else:
        scene = va.Scene(eng, host_scene)
    otris = O.tris_from_tri64(tris)
    if rig is not None:
        scene.set_tri_attribs(rig[1].view(va.TRI_ATTRIBS))
        scene.set_alpha(rig[2].view(va.ALPHA_MATERIAL), rig[3])
        host_scene.set_alpha_host(rig[1].view(va.TRI_ATTRIBS), rig[2].view(va.ALPHA_MATERIAL), rig[3])
        O.set_alpha(otris, rig[1]["uv"].reshape(n, 6), rig[1]["material"], rig[2].view(O.ALPHA_MATERIAL), rig[3])
    else:
        O.set_alpha()
    m = int(rng.integers(1, 30000))
    org = rng.uniform(-2 * spread, 2 * spread, (m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3)).astype(np.float32) * np.float32(10.0 ** rng.uniform(-3, 3))
    k = m // 3
    d[:k] = (centre[rng.integers(0, n, k), 0] - org[:k]).astype(np.float32)
    d[::37, rng.integers(0, 3)] = 0.0
    d[5::101] = np.nan                                        # fully poisoned: the reference walks the whole tree and misses
    d[7::103, rng.integers(0, 3)] = np.inf
    org[9::107, rng.integers(0, 3)] = -np.inf
    tmin = np.where(rng.random(m) < 0.3, rng.uniform(0, spread, m), 0.0).astype(np.float32)
    tmax = np.where(rng.random(m) < 0.3, tmin + rng.uniform(1e-3, 4 * spread, m), np.finfo(np.float32).max).astype(np.float32)
    rays = va.make_rays(org, d, tmin, tmax)
    ref, ref_st, _, _, _ = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, want_stats=True)
    any_ref = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays, any_hit=True)[0]
    # the single-ray latency path (host walk) against the same reference
    hk = min(m, 3000)
    if not ((host_scene.trace_closest_host(rays[:hk]).view(np.uint8) == ref[:hk].view(np.uint8)).all()
            and (host_scene.trace_any_host(rays[:hk]) == (any_ref["prim"][:hk] != O.MISS)).all()):
        bad += 1
        print(f"HOST WALK MISMATCH seed {seed} n {n}", flush=True)
    for mode in (1, 0, 2):
        eng.set_option("persistent", mode)
        # every third scene: the batch is declared an image of some row length (lanes then take 4 x 16 tiles; results must not move)
        eng.set_option("ray_image_width", int(rng.integers(1, 64)) * int(rng.choice([1, 4, 4, 8])) if seed % 3 == 0 else 0)
        d_rays = tp.to_device(rays, dev)
        d_hits, d_stats = tp.trace_stats(scene, d_rays, m)
        torch.cuda.synchronize()
        got, st = tp.to_host(d_hits, va.HIT), tp.to_host(d_stats, va.RAY_STATS)
        ok = (got.view(np.uint8) == ref.view(np.uint8)).all() and (st["steps"] == ref_st[:, 0]).all() and (st["tests"] == ref_st[:, 1]).all()
        ok = ok and (scene.trace_closest(rays).view(np.uint8) == ref.view(np.uint8)).all()
        ok = ok and (scene.trace_any(rays) == (any_ref["prim"] != O.MISS)).all()
        # single-call / tiny host batches (pinned slots + spin wait) and the persistent grid with reserved CUs
        for k in (1, 3, 256):
            kk = min(k, m)
            ok = ok and (scene.trace_closest(rays[:kk]).view(np.uint8) == ref[:kk].view(np.uint8)).all()
            ok = ok and (scene.trace_any(rays[-kk:]) == (any_ref["prim"][-kk:] != O.MISS)).all()
        if mode == 1 and seed % 4 == 1:                       # two launches in flight on two streams (scratch ring)
            s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            h1, h2 = tp.empty_records(m, va.HIT, dev), tp.empty_records(m, va.HIT, dev)
            torch.cuda.synchronize()
            for _ in range(3):
                scene.trace_closest_dev(d_rays.data_ptr(), m, h1.data_ptr(), s1.cuda_stream)
                scene.trace_closest_dev(d_rays.data_ptr(), m, h2.data_ptr(), s2.cuda_stream)
            torch.cuda.synchronize()
            ok = ok and (tp.to_host(h1, va.HIT).view(np.uint8) == ref.view(np.uint8)).all() and (tp.to_host(h2, va.HIT).view(np.uint8) == ref.view(np.uint8)).all()
        if seed % 2 == 0:                                     # round 4: the same rays cut into random batches, ONE merged launch (closest + any)
            ncut = int(rng.integers(1, 24))
            cuts = np.sort(rng.integers(0, m + 1, ncut))
            bounds = np.concatenate([[0], cuts, [m]])
            hm = torch.zeros(m * 16, dtype=torch.uint8, device=dev)
            om = torch.full((m,), 9, dtype=torch.uint8, device=dev)
            wid = [int(rng.integers(1, 40)) * 4 if rng.random() < 0.3 else 0 for _ in range(len(bounds) - 1)]
            scene.trace_multi_dev([(d_rays.data_ptr() + 32 * int(lo), hm.data_ptr() + 16 * int(lo), int(hi - lo), w)
                                   for lo, hi, w in zip(bounds[:-1], bounds[1:], wid)], tp.current_stream_handle(dev))
            scene.trace_multi_dev([(d_rays.data_ptr() + 32 * int(lo), om.data_ptr() + int(lo), int(hi - lo), w)
                                   for lo, hi, w in zip(bounds[:-1], bounds[1:], wid)], tp.current_stream_handle(dev), any_hit=True)
            torch.cuda.synchronize()
            ok = ok and (tp.to_host(hm, va.HIT).view(np.uint8) == ref.view(np.uint8)).all()
            ok = ok and (om.cpu().numpy() == (any_ref["prim"] != O.MISS)).all()
        if mode == 2 and seed % 6 == 0:                       # round 4: the batch object through the chunked pipeline, and a set of three
            bt = scene.trace_batch(rays, fetch_hits=True, image_width=int(rng.choice([0, 8, 20])))
            ok = ok and (bt.hits().view(np.uint8) == ref.view(np.uint8)).all()
            bt.free()
            third = m // 3
            st3 = scene.trace_batch_set([rays[:third], rays[third:third], rays[third:]], fetch_hits=True)
            ok = ok and (np.concatenate([b3.hits() for b3 in st3]).view(np.uint8) == ref.view(np.uint8)).all()
            for b3 in st3:
                b3.free()
        if mode == 1 and seed % 3 == 0:
            eng.set_option("reserved_cus", 32)
            ok = ok and (scene.trace_closest(rays).view(np.uint8) == ref.view(np.uint8)).all()
            eng.set_option("reserved_cus", 0)
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} mode {mode} n {n} m {m}", flush=True)
    if seed % 5 == 2 and rig is None and not poisoned:        # round 4: the shading frame of every hit (CalcTBN without a normal map + CalcFootprint)
        from vistrace_amd import workloads as W
        attribs = np.zeros(n, va.TRI_ATTRIBS)
        attribs["uv"] = rng.uniform(-4, 4, (n, 3, 2)).astype(np.float32)
        frames = W.vertex_frames(verts, seed).view(va.TRI_FRAME)
        scene.set_tri_attribs(attribs)
        scene.set_tri_frames(frames)
        cone = (float(rng.uniform(0, 1)), float(rng.uniform(1e-4, 0.05))) if seed % 2 else (-1.0, -1.0)
        d_rays = tp.to_device(rays, dev)
        d_hits = tp.trace_closest(scene, d_rays, m)
        d_tbn = tp.empty_records(m, va.HIT_TBN, dev)
        scene.hit_tbn_dev(d_rays.data_ptr(), d_hits.data_ptr(), m, d_tbn.data_ptr(), cone[0], cone[1], tp.current_stream_handle(dev))
        torch.cuda.synchronize()
        got = tp.to_host(d_tbn, va.HIT_TBN)
        exp = O.hit_tbn(otris, rays, ref, frames.view(np.float32).reshape(-1, 18), attribs["uv"].reshape(-1, 6), cone[0], cone[1])
        same = lambda a, b: ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()     # NaN payloads may differ
        ok = all(same(got[k], exp[k]) for k in ("normal", "tangent", "binormal")) and same(got["lod_info"][:, 1], exp["lod_info"][:, 1])
        ok = ok and (got["lod_set"] == exp["lod_set"]).all()
        la, lb = got["lod_info"][:, 0], exp["lod_info"][:, 0]
        fin = np.isfinite(lb)
        ok = ok and (np.abs(la[fin] - lb[fin]) <= 4e-6 + 2e-6 * np.abs(lb[fin])).all() and same(la[~fin], lb[~fin])
        if not ok:
            bad += 1
            print(f"SHADING FRAME MISMATCH seed {seed} n {n} m {m}", flush=True)
    if seed % 5 == 1 and rig is None and not poisoned:                         # device refit against the host pipeline, then trace parity
        moved = (verts + rng.normal(scale=0.05 * spread, size=verts.shape)).astype(np.float32)
        scene.refit(moved, flags)
        mtris = va.tris_setup(moved, flags)
        bvh.refit(mtris)
        ref_hs = va.HostScene(bvh)
        pairs, dtris = scene.read_records()
        ok = (dtris.view(np.uint8) == ref_hs.tris().view(np.uint8)).all() and (pairs.view(np.uint8) == ref_hs.pairs().view(np.uint8)).all()
        ref2 = O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(mtris), rays)[0]
        ok = ok and (scene.trace_closest(rays).view(np.uint8) == ref2.view(np.uint8)).all()
        if not ok:
            bad += 1
            print(f"REFIT MISMATCH seed {seed} n {n}", flush=True)
print(f"soak: seeds {first}..{first + done - 1}, {bad} mismatches, {time.time() - t0:.0f}s")
sys.exit(1 if bad else 0)
