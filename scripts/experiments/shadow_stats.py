#!/usr/bin/env python3
"""Where the steps of the any-hit (shadow) walk go on S1M: occluded against unoccluded rays (STATS any-hit kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    dev = torch.device("cuda", 0)
    eng = va.Engine(0)
    name = "S1M"
    scene = va.Scene(eng, va.HostScene(va.HostBvh(va.tris_setup(W.make_scene(name)), nthreads=16)))
    prim = W.primary_rays(1024, 1024)
    d_prim = tp.to_device(prim, dev)
    d_h = tp.trace_closest(scene, d_prim, len(prim))
    attrs = tp.to_host(tp.hit_attrs(scene, d_prim, d_h, len(prim)), va.HIT_ATTRS)
    rays = W.shadow_rays(attrs, W.light_positions(name), W.SEED + 4)
    d_rays = tp.to_device(rays, dev)
    d_occ, d_st = tp.trace_any_stats(scene, d_rays, len(rays))
    torch.cuda.synchronize()
    occ = d_occ.cpu().numpy() != 0
    st = tp.to_host(d_st, va.RAY_STATS)
    life = st["steps"].astype(np.int64) + st["tests"]
    print(f"{len(rays)} shadow rays: {100 * occ.mean():.1f} % occluded; fetches per ray: all {life.mean():.1f}, "
          f"occluded {life[occ].mean():.1f}, unoccluded {life[~occ].mean():.1f}; share of all fetches spent on occluded rays: "
          f"{100 * life[occ].sum() / life.sum():.1f} %")
    # closest-hit walk of the same rays for comparison
    _, d_st2 = tp.trace_stats(scene, d_rays, len(rays))
    torch.cuda.synchronize()
    st2 = tp.to_host(d_st2, va.RAY_STATS)
    l2 = st2["steps"].astype(np.int64) + st2["tests"]
    print(f"closest-hit walk of the same rays: {l2.mean():.1f} fetches per ray (occluded {l2[occ].mean():.1f})")


if __name__ == "__main__":
    main()
