#!/usr/bin/env python3
"""Distribution of the per-ray walk length (steps + tests = record fetches = wave iterations a ray occupies its lane) on the
headline workload, from the STATS kernel:  python scripts/ray_lifetimes.py [--scene S1M] [--side 4096]
The longest rays bound the drain at the end of a launch (profiles/r3/notes.md, "fixed cost of a launch")."""
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
    args = ap.parse_args()
    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import RAY

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
    _, d_stats = tp.trace_stats(scene, d_rays, n)
    torch.cuda.synchronize()
    st = d_stats.cpu().numpy().view(np.uint32).reshape(n, 2)
    life = st[:, 0].astype(np.int64) + st[:, 1]
    print(f"{args.scene} bounce, {n} rays: mean {life.mean():.2f}  median {np.median(life):.0f}  "
          + "  ".join(f"p{p}: {np.percentile(life, p):.0f}" for p in (90, 99, 99.9, 99.99)) + f"  max {life.max()}")
    # the longest ray of each consecutive run of 393 216 rays (what the grid holds at the end of a launch)
    runs = life[: n // 393216 * 393216].reshape(-1, 393216)
    print(f"longest ray per 393 216-ray window: mean {runs.max(1).mean():.0f}, min {runs.max(1).min()}, max {runs.max(1).max()}")
    # per 64-ray wave load: the longest ray of each wave-sized group
    w = life[: n // 64 * 64].reshape(-1, 64)
    print(f"per 64 consecutive rays: mean of max {w.max(1).mean():.1f}, mean of sum/64 {w.mean(1).mean():.1f}")


if __name__ == "__main__":
    main()
