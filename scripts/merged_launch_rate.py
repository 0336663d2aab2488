#!/usr/bin/env python3
"""Many small batches: separate launches, two streams, ONE merged launch (vt_trace_closest_multi_dev).

    python scripts/merged_launch_rate.py [--scene S100k] [--side 1024] [--reps 40]

For B = 1, 2, 4, 8, 16 independent camera-ray batches of side x side rays (different poses) and for one image cut into 16
row bands, prints Grays/s of: B plain launches on one stream; the same alternating between two streams; one merged launch
(kernel chosen by the engine, and forced either way).  Wall clock around `reps` repetitions, device synchronised on both
sides; results of every form are compared byte for byte with the plain launches."""
from __future__ import annotations

import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="S100k")
    ap.add_argument("--side", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--kind", default="primary", choices=["primary", "bounce"])
    args = ap.parse_args()

    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, RAY

    dev = torch.device("cuda", 0)
    eng = va.Engine(0)
    tris = va.tris_setup(W.make_scene(args.scene))
    scene = va.Scene(eng, va.HostScene(va.HostBvh(tris, nthreads=16)))
    side, n = args.side, args.side * args.side
    s0 = torch.cuda.current_stream(dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def make_batch(t):
        pos, fwd = W.camera_pose(args.scene, t)
        d = tp.empty_records(n, RAY, dev)
        eng.gen_primary_dev(side, side, d.data_ptr(), pos=tuple(pos), forward=tuple(fwd), stream=s0.cuda_stream)
        if args.kind == "bounce":
            h = tp.trace_closest(scene, d, n)
            a = tp.hit_attrs(scene, d, h, n)
            r = tp.empty_records(n, RAY, dev)
            eng.gen_bounce_dev(a.data_ptr(), n, W.SEED + t, r.data_ptr(), stream=s0.cuda_stream)
            return r
        return d

    width = side if args.kind == "primary" else 0
    rays = [make_batch(t) for t in range(16)]
    outs_a = [tp.empty_records(n, HIT, dev) for _ in range(16)]
    outs_b = [tp.empty_records(n, HIT, dev) for _ in range(16)]
    torch.cuda.synchronize()

    def timed(fn):
        for _ in range(3):                                   # warm: the zeroing in front of a measurement empties L2 / Infinity Cache
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps

    def plain(B, outs):
        eng.set_option("ray_image_width", width)
        for k in range(B):
            scene.trace_closest_dev(rays[k].data_ptr(), n, outs[k].data_ptr(), s0.cuda_stream)

    def two_streams(B, outs):
        eng.set_option("ray_image_width", width)
        for k in range(B):
            scene.trace_closest_dev(rays[k].data_ptr(), n, outs[k].data_ptr(), (s1 if k & 1 else s2).cuda_stream)

    def merged(B, outs):
        scene.trace_multi_dev([(rays[k].data_ptr(), outs[k].data_ptr(), n, width) for k in range(B)], s0.cuda_stream)

    def same(B):
        return all(torch.equal(outs_a[k], outs_b[k]) for k in range(B))

    print(f"# {args.scene} {args.kind}, batches of {side} x {side} rays; Grays/s (ms per repetition)")
    print("# B  plain  two_streams  merged_auto  merged_static  merged_persistent")
    for B in (1, 2, 4, 8, 16):
        row = []
        eng.set_option("persistent", 2)
        t = timed(lambda: plain(B, outs_a)); row.append((t, True))
        if B > 1:
            for o in outs_b: o.zero_()
            t = timed(lambda: two_streams(B, outs_b)); row.append((t, same(B)))
        else:
            row.append((float("nan"), True))
        res3 = {}
        for mode in (0, 1, 2):
            eng.set_option("persistent", mode)
            for o in outs_b: o.zero_()
            t = timed(lambda: merged(B, outs_b)); res3[mode] = (t, same(B))
        row += [res3[2], res3[0], res3[1]]
        eng.set_option("persistent", 2)
        print(f"{B:3d} " + "  ".join(f"{B * n / t / 1e9:6.2f} ({t * 1e3:.3f}){'' if ok else ' MISMATCH'}" for t, ok in row), flush=True)

    # one image cut into 16 bands of side / 16 rows: what a caller with many small ray sets pays per set
    rows = side // 16
    m = rows * side
    print(f"# one {side} x {side} image as 16 bands of {rows} rows ({m} rays each)")
    for o in outs_a[:1] + outs_b[:1]: o.zero_()
    eng.set_option("ray_image_width", width)

    def bands_plain():
        for k in range(16):
            scene.trace_closest_dev(rays[0].data_ptr() + 32 * m * k, m, outs_a[0].data_ptr() + 16 * m * k, s0.cuda_stream)

    def bands_merged():
        scene.trace_multi_dev([(rays[0].data_ptr() + 32 * m * k, outs_b[0].data_ptr() + 16 * m * k, m, width) for k in range(16)], s0.cuda_stream)

    def whole():
        scene.trace_closest_dev(rays[0].data_ptr(), n, outs_a[1].data_ptr(), s0.cuda_stream)

    tw = timed(whole)
    tp_ = timed(bands_plain)
    res = [f"whole image, one launch: {n / tw / 1e9:.2f} ({tw * 1e3:.3f})", f"16 plain launches: {n / tp_ / 1e9:.2f} ({tp_ * 1e3:.3f})"]
    for mode in (2, 0, 1):
        eng.set_option("persistent", mode)
        tm = timed(bands_merged)
        ok = torch.equal(outs_a[0], outs_b[0]) and torch.equal(outs_a[0], outs_a[1])
        res.append(f"merged ({['static', 'persistent', 'auto'][mode]}): {n / tm / 1e9:.2f} ({tm * 1e3:.3f}){'' if ok else ' MISMATCH'}")
    print("; ".join(res))


if __name__ == "__main__":
    main()
