#!/usr/bin/env python3
"""Would a tiled lane-to-ray mapping help camera rays?  The same primary rays in row-major order against the buffer permuted so
that every 64 consecutive rays are a WxH pixel tile (the permutation is not timed): kernel ms and steps per ray.
    python scripts/tiled_order_probe.py [--scene S1M] [--side 4096]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="S1M")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--bounce", action="store_true")
    args = ap.parse_args()
    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, RAY

    dev = torch.device("cuda", 0)
    eng = va.Engine(0)
    eng.set_timing(True)
    scene = va.Scene(eng, va.HostScene(va.HostBvh(va.tris_setup(W.make_scene(args.scene)), nthreads=16)))
    S = args.side
    n = S * S
    stream = tp.current_stream_handle(dev)
    d_rays = tp.empty_records(n, RAY, dev)
    eng.gen_primary_dev(S, S, d_rays.data_ptr(), stream=stream)
    if args.bounce:
        d_h = tp.trace_closest(scene, d_rays, n)
        d_a = tp.hit_attrs(scene, d_rays, d_h, n)
        d_b = tp.empty_records(n, RAY, dev)
        eng.gen_bounce_dev(d_a.data_ptr(), n, W.SEED + 3, d_b.data_ptr(), stream=stream)
        d_rays = d_b
    torch.cuda.synchronize()
    rays2d = d_rays.view(torch.uint8).view(S, S, RAY.itemsize)

    def timed(buf, tag):
        d_hits = tp.empty_records(n, HIT, dev)
        ms = []
        for _ in range(args.reps):
            tp.trace_closest(scene, buf, n, d_hits)
            torch.cuda.synchronize()
            ms.append(eng.last_kernel_ms())
        print(f"{tag}: median {np.median(ms):.4f} ms  ({n / np.median(ms) / 1e3:.0f} Mrays/s)", flush=True)

    timed(d_rays, "row-major")
    for tw, th in ((8, 8), (16, 4), (32, 2), (4, 16)):
        t = rays2d.view(S // th, th, S // tw, tw, RAY.itemsize).permute(0, 2, 1, 3, 4).contiguous().view(-1)
        timed(t, f"{tw}x{th} tiles per 64 rays")
        del t


if __name__ == "__main__":
    main()
