#!/usr/bin/env python3
"""Device-resident bounce loop (vt_bounce_loop_dev) against the same work issued call by call
(trace -> vt_hit_attrs_dev -> vt_gen_bounce_dev over all n paths), S1M or terrain, side^2 primary paths."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vistrace_amd as va
from vistrace_amd import torch_plumbing as tp
from vistrace_amd import workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="S1M")
ap.add_argument("--side", type=int, default=4096)
ap.add_argument("--depth", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda", 0)
eng = va.Engine(0)
if args.scene == "terrain":
    verts, flags = W.make_terrain(512)
    scene = va.build_scene(eng, verts, flags)
    cam = dict(pos=(0.0, 0.0, 80.0), forward=(1.0, 0.0, -0.6))
else:
    scene = va.build_scene(eng, W.make_scene(args.scene))
    cam = {}
n = args.side * args.side
d_prim = tp.empty_records(n, va.RAY, dev)
eng.gen_primary_dev(args.side, args.side, d_prim.data_ptr(), stream=tp.current_stream_handle(dev), **cam)
d_rows = tp.empty_records(n * args.depth, va.HIT, dev)
sh = tp.current_stream_handle(dev)

def loop():
    return scene.bounce_loop_dev(d_prim.data_ptr(), n, args.depth, 99, d_rows.data_ptr(), sh)

def composed():
    d_rays = d_prim
    d_next = [tp.empty_records(n, va.RAY, dev), tp.empty_records(n, va.RAY, dev)]
    d_attrs = tp.empty_records(n, va.HIT_ATTRS, dev)
    for d in range(args.depth):
        row = d_rows[d * n * 16:(d + 1) * n * 16]
        scene.trace_closest_dev(d_rays.data_ptr(), n, row.data_ptr(), sh)
        if d + 1 < args.depth:
            scene.hit_attrs_dev(d_rays.data_ptr(), row.data_ptr(), n, d_attrs.data_ptr(), sh)
            eng.gen_bounce_dev(d_attrs.data_ptr(), n, 99 + d, d_next[d & 1].data_ptr(), sh)
            d_rays = d_next[d & 1]

for name, fn in (("vt_bounce_loop_dev", loop), ("call-by-call composition", composed)):
    live = fn(); torch.cuda.synchronize()
    ref = d_rows.clone() if name.startswith("vt_") else ref
    if not name.startswith("vt_"):
        assert torch.equal(ref, d_rows), "loop and composition disagree"
    for _ in range(2):                    # clocks and caches: the first launches on a cold device are up to 20 % slower (profiles/r6/notes.md section 4)
        fn()
    torch.cuda.synchronize()
    each = []
    for _ in range(7):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        each.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.median(each))
    rays = sum(live) if live else n * args.depth
    print(f"{name:28s} {args.scene} {n} paths x depth {args.depth}: {ms:8.2f} ms  live per depth {live if live else '(all n)'}"
          f"  {rays / ms / 1e3:8.1f} Mrays/s (live rays)  [median of 7 calls; min {min(each):.2f}, max {max(each):.2f} ms]")
