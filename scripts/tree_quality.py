#!/usr/bin/env python3
"""Steps / tests per ray of every builder on a scene, counted by the CPU oracle (no GPU needed):

    python scripts/tree_quality.py HALL100k S100k [--rays 65536]

Rays: a 256x256 camera-0 image (coherent) and sphere rays from 16 seeded interior origins (incoherent).  Kernel time is
proportional to steps + tests per ray (one record fetch each), so this is the figure of merit of a tree."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="+")
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--builders", default="ploc,sah,sah_refined")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    import vistrace_amd as va
    from oracle import binding as O
    from vistrace_amd import workloads as W
    for name in args.scenes:
        verts = W.make_scene(name)
        tris = va.tris_setup(verts)
        otris = O.tris_from_tri64(tris)
        lo, hi = verts.reshape(-1, 3).min(0), verts.reshape(-1, 3).max(0)
        prim = W.primary_rays(256, 256)
        u = W.uniform01(W.SEED + 900, 0, 48).reshape(16, 3)
        per = args.rays // 16
        inc = np.concatenate([W.sphere_rays(per, 100 + k, origin=tuple((lo + (0.1 + 0.8 * u[k]) * (hi - lo)).astype(np.float64))) for k in range(16)])
        for builder in args.builders.split(","):
            t0 = time.time()
            bvh = va.HostBvh(tris, nthreads=args.threads, builder=builder)
            dt = time.time() - t0
            nodes, pidx = bvh.nodes().view(O.NODE), bvh.prim_indices()
            out = []
            for rays in (prim, inc):
                _, _, steps, tests, _ = O.traverse_batch(nodes, pidx, otris, rays, want_stats=True, nthreads=args.threads)
                out.append((steps / len(rays), tests / len(rays)))
            print(f"{name:9s} {builder:12s} build {dt:5.2f}s nodes {len(nodes):8d}  primary {out[0][0]:6.2f} steps {out[0][1]:5.2f} tests   "
                  f"incoherent {out[1][0]:6.2f} steps {out[1][1]:5.2f} tests  (steps+tests {sum(out[1]):6.2f})", flush=True)


if __name__ == "__main__":
    main()
