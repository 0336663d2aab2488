#!/usr/bin/env python3
"""BASELINE config 1 on the device: single-ray Traverse-equivalent calls through the C ABI
(vt_trace_closest with n = 1, host buffers) on S10k; prints us/call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vistrace_amd as va
from vistrace_amd import workloads as W

eng = va.Engine(0)
scene = va.build_scene(eng, W.make_scene("S10k"))
rays = W.sphere_rays(10000, W.SEED + 1)
for mode, spin in ((1, 1), (0, 1), (0, 0), (2, 1)):
    eng.set_option("persistent", mode)
    eng.set_option("spin_wait", spin)
    for i in range(200):
        scene.trace_closest(rays[i:i + 1])
    t0 = time.perf_counter()
    for i in range(2000):
        scene.trace_closest(rays[i:i + 1])
    dt = (time.perf_counter() - t0) / 2000
    print(f"single-ray vt_trace_closest, persistent={mode} spin_wait={spin}: {dt * 1e6:.1f} us/call (S10k, includes H2D/D2H + python ctypes)")
