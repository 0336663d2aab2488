#!/usr/bin/env python3
"""Upper bound for ray re-ordering: how fast is the 16 Mi bounce batch if its rays arrive sorted (host-side sort,
not timed) by direction octant and/or Morton code of the origin?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vistrace_amd as va
from vistrace_amd import torch_plumbing as tp, workloads as W
from vistrace_amd._lib import HIT_ATTRS

def part1by2(x):
    x = x.astype(np.uint64) & 0x3FF
    x = (x | (x << 16)) & 0x30000FF
    x = (x | (x << 8)) & 0x300F00F
    x = (x | (x << 4)) & 0x30C30C3
    x = (x | (x << 2)) & 0x9249249
    return x

dev = torch.device("cuda", 0)
eng = va.Engine(0); eng.set_timing(True)
scene = va.build_scene(eng, W.make_scene(sys.argv[1] if len(sys.argv) > 1 else "S1M"))
side = 4096; n = side * side
d_prim = tp.to_device(W.primary_rays(side, side), dev)
d_h = tp.trace_closest(scene, d_prim, n)
attrs = tp.to_host(tp.hit_attrs(scene, d_prim, d_h, n), HIT_ATTRS)
rays = W.bounce_rays(attrs, W.SEED + 3)
del attrs
org = rays["org"]; lo = org.min(axis=0); hi = org.max(axis=0)
q = np.clip(((org - lo) / (hi - lo + 1e-6) * 1023.0), 0, 1023).astype(np.uint32)
morton = (part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)).astype(np.uint64)
octant = ((rays["dir"][:, 0] < 0).astype(np.uint64) | ((rays["dir"][:, 1] < 0).astype(np.uint64) << 1) | ((rays["dir"][:, 2] < 0).astype(np.uint64) << 2))
orders = {"as generated": None, "octant only": np.argsort(octant, kind="stable"), "morton(origin, 30 bit)": np.argsort(morton, kind="stable"),
          "octant, then morton": np.argsort((octant << np.uint64(30)) | morton, kind="stable"),
          "morton >> 12 (coarse cells), then octant": np.argsort(((morton >> np.uint64(12)) << np.uint64(3)) | octant, kind="stable")}
d_hits = tp.empty_records(n, va.HIT, dev)
for name, order in orders.items():
    r = rays if order is None else rays[order]
    d_r = tp.to_device(r, dev)
    ms = []
    for _ in range(6):
        tp.trace_closest(scene, d_r, n, d_hits); ms.append(eng.last_kernel_ms())
    print(f"{name:44s} kernel {np.median(ms[2:]):6.3f} ms  {n / np.median(ms[2:]) / 1e3:7.1f} Mrays/s", flush=True)
    del d_r
