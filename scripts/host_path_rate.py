#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point vt_trace_closest (rays and hits in pageable host memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vistrace_amd as va
from vistrace_amd import workloads as W

eng = va.Engine(0)
scene = va.build_scene(eng, W.make_scene("S1M"))
rays = W.primary_rays(4096, 4096)
hits = np.zeros(len(rays), va.HIT)          # caller-owned, already touched (no first-touch page faults in the timing)
scene.trace_closest(rays[:1 << 20], out=hits[:1 << 20])
for n in (1 << 20, 1 << 21, 1 << 22, 1 << 24):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        scene.trace_closest(rays[:n], out=hits[:n])
        best = min(best, time.perf_counter() - t0)
    print(f"vt_trace_closest host buffers: {n} rays in {best * 1e3:.1f} ms = {n / best / 1e6:.0f} Mrays/s (H2D 32 B + D2H 16 B per ray included)")
import torch
from vistrace_amd import torch_plumbing as tp
d = tp.trace_closest(scene, tp.to_device(rays, torch.device("cuda", 0)), len(rays))
torch.cuda.synchronize()
assert (tp.to_host(d, va.HIT).view(np.uint8) == hits.view(np.uint8)).all(), "host path and device path disagree"
print("host-buffer results identical to the device-resident path")
