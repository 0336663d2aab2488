#!/usr/bin/env python3
"""Inputs of scripts/sim_waves.c (CPU only): the v1 tree the product builds for a scene, its triangles in the oracle's layout and
a batch of bounce rays made the way bench.py makes them (primary image -> hits -> cosine-hemisphere rays), as raw arrays.

    python scripts/sim_inputs.py /tmp/sim_s1m [--scene S1M] [--side 512] [--kind bounce|primary]
    gcc -O2 -fopenmp -ffp-contract=off scripts/sim_waves.c -o /tmp/sim_waves -lm && /tmp/sim_waves /tmp/sim_s1m 0 4 8
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scene", default="S1M")
    ap.add_argument("--side", type=int, default=512)
    ap.add_argument("--kind", default="bounce", choices=["bounce", "primary"])
    ap.add_argument("--builder", default="sah")
    args = ap.parse_args()
    import vistrace_amd as va
    from oracle import binding as O
    from vistrace_amd import workloads as W
    os.makedirs(args.out, exist_ok=True)
    tris = va.tris_setup(W.make_scene(args.scene))
    bvh = va.HostBvh(tris, nthreads=8, builder=args.builder)
    nodes, pidx, otris = bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris)
    rays = W.primary_rays(args.side, args.side)
    if args.kind == "bounce":
        hits = O.traverse_batch(nodes, pidx, otris, rays)[0]
        oa = O.hit_attrs(otris, rays, hits)
        attrs = np.zeros(len(rays), va.HIT_ATTRS)
        for k in ("pos", "ngeo", "uvw", "wo"):
            attrs[k] = oa[k]
        attrs["front"], attrs["hit"], attrs["prim"], attrs["t"] = oa["front"], hits["prim"] != O.MISS, hits["prim"], hits["t"]
        rays = W.bounce_rays(attrs, W.SEED + 3)
    nodes.tofile(os.path.join(args.out, "nodes.bin"))
    pidx.astype(np.uint32).tofile(os.path.join(args.out, "pidx.bin"))
    otris.tofile(os.path.join(args.out, "tris.bin"))
    np.ascontiguousarray(rays).tofile(os.path.join(args.out, "rays.bin"))
    print(f"{args.out}: {len(nodes)} nodes, {len(otris)} triangles, {len(rays)} {args.kind} rays")


if __name__ == "__main__":
    main()
