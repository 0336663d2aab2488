#!/usr/bin/env python3
"""Kernel time for degenerate 16 Mi-ray batches on S1M: nothing should be pathologically slow."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vistrace_amd as va
from vistrace_amd import torch_plumbing as tp, workloads as W
dev = torch.device("cuda", 0)
eng = va.Engine(0); eng.set_timing(True)
scene = va.build_scene(eng, W.make_scene("S1M"))
n = 1 << 24
rng = np.random.default_rng(1)
base = W.sphere_rays(n, 5, origin=(10.0, 20.0, 30.0))
cases = {}
cases["random directions from one point"] = base
r = base.copy(); r["org"] = (5000.0, 5000.0, 5000.0); r["dir"] = np.abs(r["dir"]); cases["all miss the scene box"] = r
r = base.copy(); r[:] = base[12345]; cases["16 Mi copies of one ray"] = r
r = base.copy(); r["tmax"] = 1e-30; cases["null rays (tmax 1e-30)"] = r
r = base.copy(); r["dir"][:, 1] = 0.0; r["dir"][:, 2] = 0.0; r["dir"][:, 0] = 1.0; cases["parallel rays, one origin"] = r
r = base.copy(); r["org"] = rng.uniform(-900, 900, (n, 3)).astype(np.float32); cases["random origins and directions"] = r
r = r.copy(); r["dir"] = (0.3, -0.5, 0.8); cases["one direction, random origins"] = r
r = base.copy(); r["dir"][::2] = np.nan; cases["every other ray NaN"] = r
d_hits = tp.empty_records(n, va.HIT, dev)
for name, rays in cases.items():
    d_r = tp.to_device(rays, dev)
    ms = []
    for _ in range(5):
        tp.trace_closest(scene, d_r, n, d_hits); ms.append(eng.last_kernel_ms())
    h = tp.to_host(d_hits, va.HIT)
    print(f"{name:36s} {np.median(ms[1:]):7.3f} ms  {n / np.median(ms[1:]) / 1e3:8.1f} Mrays/s  hit fraction {np.mean(h['prim'] != 0xFFFFFFFF):.2f}", flush=True)
    del d_r
