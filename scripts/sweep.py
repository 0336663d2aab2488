#!/usr/bin/env python3
"""Interleaved A/B timing of launch options on one device, one process (dev tool).

    python scripts/sweep.py --scene S1M --side 4096 --kind bounce \
        --opt persistent=0 --opt persistent=1,block_rays=128,refill_threshold=16 ...

Each --opt is one variant (comma-separated key=value pairs for vt_engine_set_option).
Variants are timed round-robin for --rounds rounds; prints median/min kernel ms and Mrays/s.
"""
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
    ap.add_argument("--scene", default="S1M")
    ap.add_argument("--side", type=int, default=4096)
    ap.add_argument("--kind", default="bounce", choices=["bounce", "primary", "shadow"])
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--any", action="store_true")
    ap.add_argument("--alpha", type=float, default=0.0, help="fraction of alpha-tested triangles (ALPHA kernel variants)")
    args = ap.parse_args()

    import torch

    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, HIT_ATTRS, RAY_STATS

    device = torch.device("cuda", 0)
    verts = W.make_scene(args.scene)
    rig = W.alpha_test_rig(len(verts), alpha_fraction=args.alpha) if args.alpha > 0 else None
    tris = va.tris_setup(verts, rig[0] if rig else None)
    bvh = va.HostBvh(tris)
    hs = va.HostScene(bvh)
    engine = va.Engine(0)
    scene = va.Scene(engine, hs)
    if rig:
        scene.set_tri_attribs(rig[1].view(va.TRI_ATTRIBS))
        scene.set_alpha(rig[2].view(va.ALPHA_MATERIAL), rig[3])
    side = args.side
    n = side * side
    prim = W.primary_rays(side, side)
    d_rays = tp.to_device(prim, device)
    if args.kind != "primary":
        d_h = tp.trace_closest(scene, d_rays, n)
        attrs = tp.to_host(tp.hit_attrs(scene, d_rays, d_h, n), HIT_ATTRS)
        if args.kind == "bounce":
            rays = W.bounce_rays(attrs, W.SEED + 3)
        else:
            rays = W.shadow_rays(attrs, W.light_positions(args.scene), W.SEED + 4, per_hit=1)
        d_rays = tp.to_device(rays, device)
        del attrs
    _, d_stats = tp.trace_stats(scene, d_rays, n)
    st = tp.to_host(d_stats, RAY_STATS)
    steps, tests = int(st["steps"].sum(dtype=np.uint64)), int(st["tests"].sum(dtype=np.uint64))
    alg = n * 48 + 64 * (steps + tests)
    print(f"# {args.scene} {args.kind} n={n} steps/ray {steps / n:.2f} tests/ray {tests / n:.2f} alg B/ray {alg / n:.0f}"
          f" depth {hs.max_depth}", flush=True)
    del d_stats
    d_hits = tp.empty_records(n, HIT, device)
    d_occ = torch.empty(n, dtype=torch.uint8, device=device)
    engine.set_timing(True)
    variants = args.opt or ["persistent=2"]
    keys = ["persistent", "fetch_dma", "lds_entries", "blocks_per_cu", "block_rays", "refill_threshold", "tri_threshold",
            "auto_static_factor", "static_overflow_mb", "coherent_detect", "xcd_cursors", "max_claim"]
    defaults = {}
    for k in keys:                                       # older builds (A/B runs) may not know every key
        try:
            defaults[k] = engine.get_option(k)
        except Exception:
            pass
    times = {v: [] for v in variants}
    infos = {}
    ref = None
    for r in range(args.rounds + 1):
        for v in variants:
            for k, val in defaults.items():      # every variant starts from the defaults
                engine.set_option(k, val)
            for kv in v.split(","):
                k, val = kv.split("=")
                engine.set_option(k, int(val))
            if args.any:
                tp.trace_any(scene, d_rays, n, d_occ)
            else:
                tp.trace_closest(scene, d_rays, n, d_hits)
            ms = engine.last_kernel_ms()
            infos[v] = engine.launch_info()
            if r > 0:
                times[v].append(ms)
            elif not args.any:
                h = d_hits.clone()
                if ref is None:
                    ref = h
                else:
                    assert torch.equal(ref, h), f"variant {v} changed the results"
    for v in variants:
        t = np.array(times[v])
        med = float(np.median(t))
        print(f"{v:60s} median {med:8.3f} ms  min {t.min():8.3f} ms  {n / med / 1e3:9.1f} Mrays/s  "
              f"alg {alg / med / 1e6:8.1f} GB/s  info {infos.get(v)}", flush=True)


if __name__ == "__main__":
    main()
