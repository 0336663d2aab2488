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
scene.trace_closest(rays[:1 << 20])
for n in (1 << 20, 1 << 22, 1 << 24):
    t0 = time.perf_counter()
    hits = scene.trace_closest(rays[:n])
    dt = time.perf_counter() - t0
    print(f"vt_trace_closest host buffers: {n} rays in {dt * 1e3:.1f} ms = {n / dt / 1e6:.0f} Mrays/s (H2D 32 B + D2H 16 B per ray included)")
