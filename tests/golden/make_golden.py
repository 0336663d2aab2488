#!/usr/bin/env python3
"""Generates tests/golden/s1k_golden.npz, terrain_golden.npz and shading_frame_golden.npz.

The reference has no golden vectors for this path and cannot be built here (SURVEY.md 0.2,
0.4), so these vectors are produced by the CPU oracle (oracle/vt_oracle.c) after it has been
cross-checked against the independent brute-force intersector on the same rays: hit t/u/v
bit-identical, primitive index inside the min-t set.  They pin (a) the oracle against
accidental change, (b) the host BVH builder's determinism, (c) the HIP path.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import vistrace_amd as va  # noqa: E402
from oracle import binding as O  # noqa: E402
from vistrace_amd import workloads as W  # noqa: E402


def main():
    verts = W.make_scene("S1k")
    tris = va.tris_setup(verts)
    bvh = va.HostBvh(tris, builder="ploc")      # the fixtures pin the reference-algorithm tree
    nodes, pidx = bvh.nodes(), bvh.prim_indices()
    otris = O.tris_from_tri64(tris)

    prim = W.primary_rays(64, 32)
    sph = W.sphere_rays(1024, W.SEED + 11, origin=(300.0, -200.0, 100.0))
    h0, _, _, _, _ = O.traverse_batch(nodes.view(O.NODE), pidx, otris, prim)
    a0 = O.hit_attrs(otris, prim, h0)
    attrs = np.zeros(len(a0), va.HIT_ATTRS)
    for k in ("pos", "ngeo", "uvw", "wo", "front"):
        attrs[k] = a0[k]
    attrs["hit"] = h0["prim"] != O.MISS
    bounce = W.bounce_rays(attrs[:1024], W.SEED + 12)
    rays = np.concatenate([prim, sph, bounce])
    # a few windowed rays (tmin/tmax inside the scene) to exercise the range tests
    win = rays[2048:2304].copy()
    win["tmin"], win["tmax"] = 400.0, 1100.0
    rays = np.concatenate([rays, win])

    hits, stats, _, _, _ = O.traverse_batch(nodes.view(O.NODE), pidx, otris, rays, want_stats=True)
    brute = O.trace_brute(otris, rays)
    assert (brute["t"].view(np.uint32) == hits["t"].view(np.uint32)).all()
    ties = 0
    for i in np.nonzero(brute["prim"] != hits["prim"])[0]:
        _, ids, n = O.min_t_set(otris, rays[i:i + 1])
        assert hits["prim"][i] in ids[:n]
        ties += 1
    any_hits, _, _, _, _ = O.traverse_batch(nodes.view(O.NODE), pidx, otris, rays, any_hit=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s1k_golden.npz")
    np.savez_compressed(out, verts=verts, nodes=nodes, prim_indices=pidx, rays=rays, hits=hits, stats=stats,
                        occluded=(any_hits["prim"] != O.MISS).astype(np.uint8))
    # ---- second fixture: one-sided heightfield (cull flag set), rays from above and below, windows ----
    tverts, tflags = W.make_terrain(16)
    ttris = va.tris_setup(tverts, tflags)
    tbvh = va.HostBvh(ttris, builder="ploc")
    tnodes, tpidx, totris = tbvh.nodes(), tbvh.prim_indices(), O.tris_from_tri64(ttris)
    trays = np.concatenate([W.sphere_rays(1024, W.SEED + 21, origin=(0.0, 0.0, 60.0)),
                            W.sphere_rays(1024, W.SEED + 22, origin=(3.0, -4.0, -30.0))])
    twin = trays[:256].copy()
    twin["tmin"], twin["tmax"] = 30.0, 70.0
    trays = np.concatenate([trays, twin])
    thits, tstats, _, _, _ = O.traverse_batch(tnodes.view(O.NODE), tpidx, totris, trays, want_stats=True)
    tbrute = O.trace_brute(totris, trays)
    assert (tbrute["t"].view(np.uint32) == thits["t"].view(np.uint32)).all()
    assert ((tbrute["prim"] == O.MISS) == (thits["prim"] == O.MISS)).all()
    tout = os.path.join(os.path.dirname(os.path.abspath(__file__)), "terrain_golden.npz")
    np.savez_compressed(tout, verts=tverts, flags=tflags, nodes=tnodes, prim_indices=tpidx, rays=trays, hits=thits, stats=tstats)
    # ---- third fixture: the shading frame of the s1k fixture's hits (TraceResult::CalcTBN without a normal map +
    # CalcFootprint) from seeded vertex frames and uvs, cone on; and those frames skinned by one seeded pose ----
    frames = W.vertex_frames(verts, W.SEED + 31)
    uv = np.random.default_rng(W.SEED + 32).uniform(-2, 2, (len(verts), 3, 2)).astype(np.float32)
    cone = (0.25, 0.004)
    sel = np.arange(0, len(rays), 4)                         # every fourth ray of the s1k fixture keeps the file small
    tbn = O.hit_tbn(otris, rays[sel], hits[sel], frames.view(np.float32).reshape(-1, 18), uv.reshape(-1, 6), cone[0], cone[1])
    tbn_off = O.hit_tbn(otris, rays[sel], hits[sel], frames.view(np.float32).reshape(-1, 18), uv.reshape(-1, 6))
    skin, base, nmat = W.skinned_rig(len(verts), nents=4, bones_per_ent=6, seed=W.SEED + 33)
    bones, binds = W.rig_pose(nmat, 2, seed=W.SEED + 34)
    skinned = O.skin_frames(frames.view(np.float32).reshape(-1, 18), skin, base, O.skin_matrices(bones, binds))
    fout = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shading_frame_golden.npz")
    np.savez_compressed(fout, ray_index=sel, frames=frames.view(np.float32).reshape(-1, 18), uv=uv, cone=np.array(cone, np.float32), tbn=tbn, tbn_cone_off=tbn_off,
                        skin=skin, matrix_base=base, bones=bones, binds=binds, skinned_frames=skinned)
    print(f"wrote {fout}: {int((hits['prim'][sel] != O.MISS).sum())} frames of hits, {len(verts)} skinned triangle frames, {os.path.getsize(fout) / 1024:.0f} KiB")
    print(f"wrote {tout}: {len(trays)} rays, {int((thits['prim'] != O.MISS).sum())} hits, {os.path.getsize(tout) / 1024:.0f} KiB")
    print(f"wrote {out}: {len(rays)} rays, {int((hits['prim'] != O.MISS).sum())} hits, {ties} tie-broken indices, "
          f"{os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
