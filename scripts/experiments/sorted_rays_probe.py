#!/usr/bin/env python3
"""What would sorting an incoherent batch buy?  (experiment, round 6: measured before anything is built)

The headline batch (16 Mi cosine-hemisphere bounce rays, S1M) is traced as it comes and again after a permutation that brings
rays with similar origins / directions together (keys made with torch on the device, torch.sort; the sort itself is NOT timed here:
the question is the kernel's time on a coherent ordering, i.e. the most a sorting pass could win).  Per-ray results are unchanged
by construction (checked: the hit records, un-permuted, are byte-equal)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import vistrace_amd as va
from vistrace_amd import torch_plumbing as tp
from vistrace_amd import workloads as W
from vistrace_amd._lib import HIT, RAY

scene_name = sys.argv[1] if len(sys.argv) > 1 else "S1M"
side = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
eng = va.Engine(0)
tris = va.tris_setup(W.make_scene(scene_name))
scene = va.Scene.from_tree(eng, va.HostBvh(tris, nthreads=16))
n = side * side
stream = tp.current_stream_handle(dev)
d_prim = tp.empty_records(n, RAY, dev)
eng.gen_primary_dev(side, side, d_prim.data_ptr(), stream=stream)
d_h = tp.trace_closest(scene, d_prim, n)
d_a = tp.hit_attrs(scene, d_prim, d_h, n)
d_rays = tp.empty_records(n, RAY, dev)
eng.gen_bounce_dev(d_a.data_ptr(), n, W.SEED + 3, d_rays.data_ptr(), stream=stream)
del d_h, d_a, d_prim
rays = d_rays.view(torch.float32).view(n, 8)


def spread3(v):                      # 10 bits -> every third bit
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton(p, lo, hi, bits):
    q = ((p - lo) / (hi - lo) * (1 << bits)).clamp(0, (1 << bits) - 1).to(torch.int64)
    return spread3(q[:, 0]) | (spread3(q[:, 1]) << 1) | (spread3(q[:, 2]) << 2)


org, d = rays[:, 0:3], rays[:, 3:6]
lo, hi = org.min(0).values, org.max(0).values
octant = ((d[:, 0] < 0).to(torch.int64) | ((d[:, 1] < 0).to(torch.int64) << 1) | ((d[:, 2] < 0).to(torch.int64) << 2))
dn = d / d.norm(dim=1, keepdim=True)
keys = {
    "as generated (image order of the primary hits)": None,
    "random permutation": torch.randperm(n, device=dev),
    "origin Morton 30 bits": morton(org, lo, hi, 10),
    "octant, then origin Morton 30 bits": (octant << 30) | morton(org, lo, hi, 10),
    "origin cell 4 bits/axis, then direction Morton 5 bits/axis, then origin": (morton(org, lo, hi, 4) << 45) | (morton(dn, -1.0, 1.0, 5) << 30) | morton(org, lo, hi, 10),
    "origin cell 5 bits/axis, octant (18-bit key: what a cheap binning pass could do)": (morton(org, lo, hi, 5) << 3) | octant,
    "origin cell 3 bits/axis, direction Morton 4 bits/axis (21-bit key)": (morton(org, lo, hi, 3) << 12) | morton(dn, -1.0, 1.0, 4),
}
d_hits = tp.empty_records(n, HIT, dev)
eng.set_timing(True)
ref = None
for name, key in keys.items():
    if key is None:
        perm, sorted_rays = None, d_rays
    else:
        perm = key if name.startswith("random") else torch.sort(key, stable=True).indices
        sorted_rays = rays[perm].contiguous().view(torch.uint8).view(-1)
    ms = []
    for _ in range(12):
        tp.trace_closest(scene, sorted_rays, n, d_hits)
        ms.append(eng.last_kernel_ms())
    torch.cuda.synchronize()
    hits = d_hits.view(torch.int32).view(n, 4)
    if perm is not None:
        back = torch.empty_like(hits)
        back[perm] = hits
        hits = back
    if ref is None:
        ref = hits.clone()
    same = bool(torch.equal(ref, hits))
    print(f"{name:86s} kernel median {np.median(ms[2:]):6.3f} ms  min {min(ms[2:]):6.3f}  results equal: {same}", flush=True)
    del sorted_rays
