#!/usr/bin/env python3
"""Independent batches on two streams: does the next launch's start hide this launch's drain?  (profiles/r3/notes.md section 6)
    python scripts/overlap_launches.py [--side 4096] [--launches 100]
K launches of the headline batch back to back on ONE stream against the same K launches alternating between TWO streams (each
with its own hit buffer).  Both orders produce the same hit records."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="S1M")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--launches", type=int, default=100)
    ap.add_argument("--streams", type=int, default=2)
    args = ap.parse_args()
    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, RAY

    dev = torch.device("cuda", 0)
    eng = va.Engine(0)
    scene = va.Scene(eng, va.HostScene(va.HostBvh(va.tris_setup(W.make_scene(args.scene)), nthreads=16)))
    n = args.side * args.side
    stream = tp.current_stream_handle(dev)
    d_prim = tp.empty_records(n, RAY, dev)
    eng.gen_primary_dev(args.side, args.side, d_prim.data_ptr(), stream=stream)
    d_h = tp.trace_closest(scene, d_prim, n)
    d_a = tp.hit_attrs(scene, d_prim, d_h, n)
    d_rays = tp.empty_records(n, RAY, dev)
    eng.gen_bounce_dev(d_a.data_ptr(), n, W.SEED + 3, d_rays.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    S = max(2, args.streams)
    hits = [tp.empty_records(n, HIT, dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

    def run(two):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.launches):
            s = streams[k % S] if two else streams[0]
            scene.trace_closest_dev(d_rays.data_ptr(), n, hits[k % S].data_ptr(), s.cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.launches * 1e3

    for _ in range(2):
        one, two = run(False), run(True)
        same = bool((hits[0].view(torch.int64) == hits[1].view(torch.int64)).all())
        print(f"{n} rays x {args.launches} launches: one stream {one:.3f} ms per launch ({n / one / 1e3:.0f} Mrays/s), {S} streams "
              f"{two:.3f} ms ({n / two / 1e3:.0f} Mrays/s, {100 * (one / two - 1):+.1f} %); hit records equal: {same}")


if __name__ == "__main__":
    main()
