#!/usr/bin/env python3
"""Per-frame host time of a group-wide scene update (vt_scene_skin_refit, vt_scene_refit) with 1 / 2 / 8 members.
Members share device 0 (test hooks + the RCCL test double; scripts in profiles/r5/notes.md): the DEVICE work of N members then
runs on one GPU one after the other, so the figure shows the host side of the phases -- with one device per member the device
work overlaps as well.  Usage: VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1 python3 scripts/group_update_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vistrace_amd as va
from vistrace_amd import workloads as W

verts = W.make_scene("S1M")
tris = va.tris_setup(verts)
bvh = va.HostBvh(tris)
hs = va.HostScene(bvh)
skin, base, nmat = W.skinned_rig(len(verts), nents=64, bones_per_ent=32)
bones, binds = W.rig_pose(nmat, 0)
moved = (verts + np.float32(0.25)).astype(np.float32)
for members in (1, 2, 8):
    eng = va.Engine([0] * members) if members > 1 else va.Engine(0)
    va.Scene.from_tree(eng, bvh).free()                      # first upload of an engine: staging allocations
    up = []
    for _ in range(5):
        t0 = time.perf_counter()
        sc = va.Scene.from_tree(eng, bvh)                    # Rebuild's upload step: vt_scene_upload_tree on the root, replicas by device-to-device copies
        up.append((time.perf_counter() - t0) * 1e3)
        rep_us = (eng.get_option("last_update_enqueue_us"), eng.get_option("last_update_wait_us")) if members > 1 else (0, 0)
        sc.free()
    print(f"members {members}: vt_scene_upload_tree {np.median(up):.2f} ms (replicas: prepare + enqueue {rep_us[0]} us, waits {rep_us[1]} us)", flush=True)
    scene = va.Scene(eng, hs)
    scene.set_skin(verts, skin, base)
    for _ in range(5):
        scene.skin_refit(bones, binds)
    t0 = time.perf_counter()
    enq, wait = [], []
    for _ in range(20):
        scene.skin_refit(bones, binds)
        enq.append(eng.get_option("last_update_enqueue_us")); wait.append(eng.get_option("last_update_wait_us"))
    skin_ms = (time.perf_counter() - t0) / 20 * 1e3
    scene.refit(moved)
    t0 = time.perf_counter()
    r_enq, r_wait = [], []
    for _ in range(5):
        scene.refit(moved)
        r_enq.append(eng.get_option("last_update_enqueue_us")); r_wait.append(eng.get_option("last_update_wait_us"))
    refit_ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"members {members}: vt_scene_skin_refit {skin_ms:.3f} ms per frame ({skin_ms / members:.3f} per member), vt_scene_refit {refit_ms:.2f} ms "
          f"({refit_ms / members:.2f} per member); early waits {eng.get_option('last_update_early_waits')} of {eng.get_option('last_update_members')} members; "
          f"skin refit host time: prepare + enqueue of all members {np.median(enq):.0f} us, waits {np.median(wait):.0f} us; "
          f"refit host time: prepare + enqueue of all members {np.median(r_enq):.0f} us ({np.median(r_enq) / members:.0f} per member; includes staging "
          f"the 36 MB of vertices once), waits {np.median(r_wait):.0f} us (the members' device work, serialised here on ONE GPU)", flush=True)
    scene.free()
    eng.close()
