#!/usr/bin/env python3
"""vt_scene_upload_tree on S1M / S10M a few times: wall time and the library's own breakdown (dev tool; run under
rocprofv3 --kernel-trace --stats for the kernels' share)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vistrace_amd as va
from vistrace_amd import workloads as W

eng = va.Engine(0)
for name in sys.argv[1:] or ["S1M"]:
    tris = va.tris_setup(W.make_scene(name))
    bvh = va.HostBvh(tris, nthreads=16)
    for k in range(5):
        t0 = time.perf_counter()
        sc = va.Scene.from_tree(eng, bvh)
        ms = (time.perf_counter() - t0) * 1e3
        print(name, k, f"{ms:.2f} ms", sc.upload_stats(), flush=True)
        sc.free()
