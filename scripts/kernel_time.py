#!/usr/bin/env python3
"""Kernel time of the library named by VISTRACE_HIP_LIB on a few workloads (dev tool for A/B runs).

    VISTRACE_HIP_LIB=vistrace_amd/lib/variants/libvistrace_hip_x.so python scripts/kernel_time.py --work S1M:bounce,S1M:primary

Prints one line per workload: median / min kernel ms over --reps launches (HIP events around each launch) and a
64-bit checksum of the hit records (equal checksums across variants = same results)."""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="S1M:bounce")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--opt", action="append", default=[], help="key=value engine options")
    ap.add_argument("--tag", default="")
    ap.add_argument("--builder", default="sah")
    args = ap.parse_args()

    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, RAY

    dev = torch.device("cuda", 0)
    scenes = {}
    for item in args.work.split(","):
        name, kind = item.split(":")
        if name not in scenes:
            tris = va.tris_setup(W.make_scene(name))
            eng = va.Engine(0)
            for o in args.opt:
                k, v = o.split("=")
                eng.set_option(k, int(v))
            scenes[name] = (eng, va.Scene(eng, va.HostScene(va.HostBvh(tris, nthreads=16, builder=args.builder))))
        eng, scene = scenes[name]
        n = args.side * args.side
        stream = tp.current_stream_handle(dev)
        d_prim = tp.empty_records(n, RAY, dev)
        eng.gen_primary_dev(args.side, args.side, d_prim.data_ptr(), stream=stream)
        if kind == "primary":
            d_rays = d_prim
        else:
            d_h = tp.trace_closest(scene, d_prim, n)
            d_a = tp.hit_attrs(scene, d_prim, d_h, n)
            d_rays = tp.empty_records(n, RAY, dev)
            eng.gen_bounce_dev(d_a.data_ptr(), n, W.SEED + 3, d_rays.data_ptr(), stream=stream)
            del d_h, d_a
        d_hits = tp.empty_records(n, HIT, dev)
        eng.set_timing(True)
        ms = []
        for _ in range(args.reps + 2):
            if kind == "any":
                tp.trace_any(scene, d_rays, n)
            else:
                tp.trace_closest(scene, d_rays, n, d_hits)
            ms.append(eng.last_kernel_ms())
        ms = ms[2:]
        torch.cuda.synchronize()
        chk = int(d_hits.view(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF
        if kind != "any":
            d_s = tp.trace_stats(scene, d_rays, n)[1]
            st = d_s.view(torch.int32).view(n, 2).sum(dim=0, dtype=torch.int64).cpu().numpy() / n
            extra = f" steps {st[0]:.2f} tests {st[1]:.2f} depth {scene.host_scene.max_depth}"
            del d_s
        else:
            extra = ""
        print(f"{args.tag or os.path.basename(os.environ.get('VISTRACE_HIP_LIB', 'default'))} {item}: median {np.median(ms):.4f} min {min(ms):.4f} ms  chk {chk:016x}{extra}", flush=True)
        del d_rays, d_hits, d_prim


if __name__ == "__main__":
    main()
