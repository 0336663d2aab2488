#!/usr/bin/env python3
"""The shading-frame kernels on S1M: vt_hit_tbn_dev over 16 Mi hits (camera rays, bounce rays; cone off / on) beside
vt_hit_attrs_dev and vt_hit_shade_dev, and what the per-vertex frames add to vt_scene_skin_refit.
Usage: python scripts/shading_frame_rate.py [side]   (side^2 rays, default 4096)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vistrace_amd as va
from vistrace_amd import torch_plumbing as tp
from vistrace_amd import workloads as W

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = side * side
eng = va.Engine(0)
dev = torch.device("cuda", 0)
verts = W.make_scene("S1M")
nt = len(verts)
scene = va.Scene(eng, va.HostScene(va.HostBvh(va.tris_setup(verts))))
rng = np.random.default_rng(1)
attribs = np.zeros(nt, va.TRI_ATTRIBS)
attribs["uv"] = rng.uniform(-2, 2, (nt, 3, 2)).astype(np.float32)
scene.set_tri_attribs(attribs)
frames = W.vertex_frames(verts).view(va.TRI_FRAME)
scene.set_tri_frames(frames)
sh = tp.current_stream_handle(dev)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


pos, fwd = W.camera_pose("S1M", 0)
d_prim = tp.to_device(W.primary_rays(side, side, pos=pos, forward=fwd), dev)
d_h = tp.trace_closest(scene, d_prim, n)
d_a = tp.hit_attrs(scene, d_prim, d_h, n)
d_bounce = tp.empty_records(n, va.RAY, dev)
eng.gen_bounce_dev(d_a.data_ptr(), n, 7, d_bounce.data_ptr(), stream=sh)
d_hb = tp.trace_closest(scene, d_bounce, n)
d_out = tp.empty_records(n, va.HIT_TBN, dev)
d_sh = tp.empty_records(n, va.HIT_SHADE, dev)
d_at = tp.empty_records(n, va.HIT_ATTRS, dev)
for name, d_r, d_hits in (("camera", d_prim, d_h), ("bounce", d_bounce, d_hb)):
    hit_frac = float((tp.to_host(d_hits, va.HIT)["prim"] != 0xFFFFFFFF).mean())
    t_at = timed(lambda: scene.hit_attrs_dev(d_r.data_ptr(), d_hits.data_ptr(), n, d_at.data_ptr(), sh))
    t_sh = timed(lambda: scene.hit_shade_dev(d_hits.data_ptr(), n, d_sh.data_ptr(), sh))
    t_off = timed(lambda: scene.hit_tbn_dev(d_r.data_ptr(), d_hits.data_ptr(), n, d_out.data_ptr(), -1.0, -1.0, sh))
    t_on = timed(lambda: scene.hit_tbn_dev(d_r.data_ptr(), d_hits.data_ptr(), n, d_out.data_ptr(), 0.0, 0.002, sh))
    # algorithmic bytes per hit: ray 32 + hit 16 + slot 4 + record 64 + frame 72 (+ attribs 48 with the cone) + 48 written
    b_off, b_on = 32 + 16 + 4 + 64 + 72 + 48, 32 + 16 + 4 + 64 + 72 + 48 + 48
    print(f"{name} rays, {n} records ({hit_frac:.2f} hit): hit_attrs {t_at:.3f} ms, hit_shade {t_sh:.3f} ms, "
          f"hit_tbn cone off {t_off:.3f} ms = {n * b_off / t_off / 1e9:.2f} TB/s of algorithmic bytes, "
          f"cone on {t_on:.3f} ms = {n * b_on / t_on / 1e9:.2f} TB/s")

skin, base, nmat = W.skinned_rig(nt, nents=64, bones_per_ent=32)
bones, binds = W.rig_pose(nmat, 0)
plain = va.Scene(eng, va.HostScene(va.HostBvh(va.tris_setup(verts))))
for label, sc in (("positions only", plain), ("positions + vertex frames", scene)):
    sc.set_skin(verts, skin, base)
    for _ in range(10):
        sc.skin_refit(bones, binds)
    t0 = time.perf_counter()
    for _ in range(20):
        sc.skin_refit(bones, binds)
    print(f"vt_scene_skin_refit, {label}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per frame ({nt} triangles, {nmat} matrix pairs)")
