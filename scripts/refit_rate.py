#!/usr/bin/env python3
"""Time a device-side refit of S1M against a full CPU Rebuild (build + linearise + upload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vistrace_amd as va
from vistrace_amd import workloads as W

eng = va.Engine(0)
verts = W.make_scene("S1M")
t0 = time.perf_counter(); tris = va.tris_setup(verts); bvh = va.HostBvh(tris); hs = va.HostScene(bvh); scene = va.Scene(eng, hs); t1 = time.perf_counter()
print(f"full Rebuild (setup + PLOC + collapse + linearise + upload): {(t1 - t0) * 1e3:.0f} ms")
moved = (verts + np.float32(0.25)).astype(np.float32)
scene.refit(moved)
t0 = time.perf_counter()
for _ in range(5):
    scene.refit(moved)
t1 = time.perf_counter()
print(f"vt_scene_refit (H2D of 36 MB vertices + {hs.max_depth} level launches): {(t1 - t0) / 5 * 1e3:.1f} ms")

# skinning on the device: per frame only the matrices cross the bus
skin, base, nmat = W.skinned_rig(len(verts), nents=64, bones_per_ent=32)
scene.set_skin(verts, skin, base)
bones, binds = W.rig_pose(nmat, 0)
for f in range(10):
    scene.skin_refit(bones, binds)
t0 = time.perf_counter()
for f in range(20):
    scene.skin_refit(bones, binds)
t1 = time.perf_counter()
print(f"vt_scene_skin_refit ({nmat} matrix pairs = {nmat * 128 / 1024:.0f} KiB H2D, skin + records + refit on device): "
      f"{(t1 - t0) / 20 * 1e3:.2f} ms per frame")
