#!/usr/bin/env python3
"""Measure the BASELINE.json configs 2-5 on one GPU (dev/reporting tool; bench.py is the contract).

Prints one JSON line per config: rays, kernel ms (HIP events), Mrays/s, steps/tests per ray,
algorithmic GB/s.  Config 5 (S10M, 128 Mi primary rays = 128 camera tiles): ONE rank's contiguous shard
(16 tiles = 16 Mi rays) in one launch here; the 8-GPU form shards the tiles across ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--tiles", type=int, default=16, help="camera tiles of config 5 in one launch (16 = one of 8 ranks' shard)")
    args = ap.parse_args()
    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, HIT_ATTRS, RAY_STATS
    dev = torch.device("cuda", 0)
    engine = va.Engine(0)
    engine.set_timing(True)
    scenes = {}

    def scene(name):
        if name not in scenes:
            t0 = time.time()
            tris = va.tris_setup(W.make_scene(name))
            t1 = time.time()
            bvh = va.HostBvh(tris)
            t2 = time.time()
            hs = va.HostScene(bvh)
            scenes[name] = (va.Scene(engine, hs), hs, t2 - t1)
            print(f"# {name}: {len(tris)} tris, build {t2 - t1:.2f}s, setup {t1 - t0:.2f}s, depth {hs.max_depth}, "
                  f"{scenes[name][0].device_bytes / 1e6:.0f} MB", file=sys.stderr, flush=True)
        return scenes[name]

    def measure(cfg, name, d_rays, n, any_hit=False, reps=3):
        sc, hs, build_s = scene(name)
        _, d_stats = tp.trace_stats(sc, d_rays, n)
        torch.cuda.synchronize()
        st = tp.to_host(d_stats, RAY_STATS)
        steps, tests = int(st["steps"].sum(dtype=np.uint64)), int(st["tests"].sum(dtype=np.uint64))
        del d_stats, st
        out = torch.empty(n if any_hit else n * 16, dtype=torch.uint8, device=dev)
        ms = []
        warm = 4                                        # the clocks need a few launches after the idle CPU build phase
        for _ in range(reps + warm):
            if any_hit:
                tp.trace_any(sc, d_rays, n, out)
            else:
                tp.trace_closest(sc, d_rays, n, out)
            ms.append(engine.last_kernel_ms())
        k = float(np.median(ms[warm:]))
        alg = n * (32 + (1 if any_hit else 16)) + 64 * (steps + tests)   # closest-hit counters as the yardstick
        return {"config": cfg, "scene": name, "rays": n, "query": "any-hit" if any_hit else "closest-hit",
                "kernel_ms": round(k, 3), "mrays_s": round(n / k / 1e3, 1), "steps_per_ray": round(steps / n, 2),
                "tests_per_ray": round(tests / n, 2), "alg_gb_s": round(alg / k / 1e6, 1), "bvh_build_s": round(build_s, 2)}

    def bounce_from(name, side, seed, shadow=False):
        sc, _, _ = scene(name)
        n = side * side
        d_prim = tp.to_device(W.primary_rays(side, side), dev)
        d_h = tp.trace_closest(sc, d_prim, n)
        attrs = tp.to_host(tp.hit_attrs(sc, d_prim, d_h, n), HIT_ATTRS)
        del d_prim, d_h
        if shadow:
            return W.shadow_rays(attrs, W.light_positions(name), seed, per_hit=4)
        return W.bounce_rays(attrs, seed)

    for cfg in args.configs.split(","):
        if cfg == "2":
            rays = W.primary_rays(1024, 1024)
            print(json.dumps(measure(2, "S100k", tp.to_device(rays, dev), len(rays))), flush=True)
        elif cfg == "3":
            rays = bounce_from("S1M", 4096, W.SEED + 3)
            print(json.dumps(measure(3, "S1M", tp.to_device(rays, dev), len(rays))), flush=True)
        elif cfg == "4":
            rays = bounce_from("S1M", 4096, W.SEED + 4, shadow=True)     # 64 Mi shadow rays
            d = tp.to_device(rays, dev)
            n = len(rays)
            del rays
            print(json.dumps(measure(4, "S1M", d, n, any_hit=True)), flush=True)
            del d
        elif cfg == "5":
            # one rank's contiguous shard of the 128-tile ray array (N/G with G = 8 -> 16 tiles), one launch
            tile = 1024 * 1024
            d_rays = tp.empty_records(args.tiles * tile, va.RAY, dev)
            for t in range(args.tiles):
                pos, fwd = W.camera_pose("S10M", t)
                engine.gen_primary_dev(1024, 1024, d_rays.data_ptr() + t * tile * va.RAY.itemsize, pos=tuple(float(x) for x in pos),
                                       forward=tuple(float(x) for x in fwd), stream=tp.current_stream_handle(dev))
            torch.cuda.synchronize()
            r = measure(5, "S10M", d_rays, args.tiles * tile, reps=3)
            r["tiles"] = args.tiles
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
