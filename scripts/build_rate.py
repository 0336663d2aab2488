#!/usr/bin/env python3
"""CPU BVH build time (vt_bvh_build: PLOC + leaf collapse) against the OpenMP thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vistrace_amd as va
from vistrace_amd import workloads as W
name = sys.argv[1] if len(sys.argv) > 1 else "S1M"
tris = va.tris_setup(W.make_scene(name))
for t in (0, 1, 4, 8, 16, 32, 64, 128, 256):
    if t > len(os.sched_getaffinity(0)):
        break
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); bvh = va.HostBvh(tris, nthreads=t); best = min(best, time.perf_counter() - t0)
    print(f"{name}: nthreads {t:3d} (0 = default): build {best * 1e3:7.0f} ms", flush=True)
