#!/usr/bin/env python3
"""How exposed is the "parity unpinned" oracle to its recalled details?  (CPU only; writes tests/golden/recall_sensitivity.json)

libs/bvh is an empty, unpinned submodule of the reference (.gitmodules:4-6), so the walk the oracle restates
(source/objects/AccelStruct.h:23-31, AccelStruct.cpp:818) follows madmann91/bvh v1 AS RECALLED in SURVEY.md section 3.2,
whose confidence table names the details that could differ upstream.  For each of them oracle/vt_oracle.c has a
-DVTO_ALT_<X> switch that reads the detail the other plausible way.  This script runs the shipped reading and every
variant over the same rays and counts, per workload, the rays whose result would change:

  miss_flip   hit <-> miss differs            prim   primitive index differs (both hit)
  t / uv      t or (u, v) bits differ (both hit)     counters_only   same hit record, different step / test counters

Workloads: the first 1 Mi rays of the headline batch (S1M, cosine-hemisphere bounce rays off camera 0, host generation,
default tree), BASELINE configs[1] (S100k, 1024 x 1024 primary), the two committed fixtures on their pinned PLOC trees,
the four seeded soups and the "weird rays" set of tests/test_gpu_parity.py.  A ninth row is not a switch: the same rays on
the reference's builder algorithm (PLOC) against the default tree (binned SAH) -- tree shape as a recalled detail.

Usage:  python scripts/recall_sensitivity.py [--quick]      (--quick: 64 Ki rays of the two big workloads; used by the test)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import vistrace_amd as va  # noqa: E402
from oracle import binding as O  # noqa: E402
from vistrace_amd import workloads as W  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

READINGS = {
    "PLAIN_INVERSE": ("safe_inverse clamp |x| <= FLT_EPSILON", "plain 1/x (early v1: +-inf)"),
    "SWAP_GE": ("near/far swap on dL.first > dR.first", "swap on >="),
    "FMA": ("fast_multiply_add unfused (a*b, then +c)", "fused fmaf (FP_FAST_FMAF builds)"),
    "RETEST_RIGHT": ("both children slab-tested before either leaf", "right child tested after the left leaf (shrunk tmax)"),
    "LEAF_DESC": ("leaf slots in ascending order", "descending order"),
    "ACCEPT_LT": ("node accepted on first <= second", "first < second"),
    "PUSH_NODE_CULL": ("far child's first-child index pushed, no test on pop", "far node + entry distance pushed, dropped on pop if entry > tmax"),
    "FMINMAX": ("robust_max/min (a > b ? a : b)", "fmaxf / fminf (NaN-ignoring)"),
}


def tree(tris, builder=None):
    bvh = va.HostBvh(tris, builder=builder) if builder else va.HostBvh(tris)
    return bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris)


def diff(ref, ref_st, got, got_st):
    rh, gh = ref["prim"] != O.MISS, got["prim"] != O.MISS
    both = rh & gh
    flip = rh != gh
    prim = both & (ref["prim"] != got["prim"])
    t = both & (ref["t"].view(np.uint32) != got["t"].view(np.uint32))
    uv = both & ((ref["u"].view(np.uint32) != got["u"].view(np.uint32)) | (ref["v"].view(np.uint32) != got["v"].view(np.uint32)))
    same_hit = ~(flip | prim | t | uv)
    ctr = same_hit & ((ref_st[:, 0] != got_st[:, 0]) | (ref_st[:, 1] != got_st[:, 1]))
    return {"miss_flip": int(flip.sum()), "prim": int(prim.sum()), "t": int(t.sum()), "uv": int(uv.sum()),
            "counters_only": int(ctr.sum()),
            "steps_ratio": round(float(got_st[:, 0].sum()) / max(1.0, float(ref_st[:, 0].sum())), 6),
            "tests_ratio": round(float(got_st[:, 1].sum()) / max(1.0, float(ref_st[:, 1].sum())), 6)}


def big_workloads(quick: bool):
    """Yields (name, nodes, prim_indices, otris, rays, tris64): the two BASELINE-sized workloads."""
    n_big = 1 << (16 if quick else 20)
    # headline: rows of the 4096 x 4096 camera-0 image whose bounce rays are the first n_big rays of S1M_bounce16777216
    verts = W.make_scene("S1M")
    tris = va.tris_setup(verts)
    nodes, pidx, otris = tree(tris)
    cam = W.camera_positions("S1M")[0]
    side = 4096
    prim = W.primary_rays(side, side, pos=cam)[:n_big]
    h0, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, prim)
    a0 = O.hit_attrs(otris, prim, h0)                                   # TraceResult.cpp:45-86, 255-262 (oracle)
    attrs = np.zeros(len(a0), va.HIT_ATTRS)
    for k in ("pos", "ngeo", "uvw", "wo", "front"):
        attrs[k] = a0[k]
    attrs["hit"] = h0["prim"] != O.MISS
    bounce = W.bounce_rays(attrs, W.SEED + 3)
    yield f"headline S1M bounce (first {n_big} of 16777216, default tree)", nodes, pidx, otris, bounce, tris
    del verts
    # configs[1]
    verts = W.make_scene("S100k")
    tris = va.tris_setup(verts)
    nodes, pidx, otris = tree(tris)
    side = 256 if quick else 1024
    yield f"configs[1] S100k primary {side}x{side}", nodes, pidx, otris, W.primary_rays(side, side, pos=W.camera_positions("S100k")[0]), tris


def small_workloads():
    """Yields (name, nodes, prim_indices, otris, rays, tris64): fixtures, soups, weird rays (seconds; re-run by the test)."""
    # committed fixtures on their pinned trees
    for fx in ("s1k_golden.npz", "terrain_golden.npz"):
        g = np.load(os.path.join(GOLD, fx))
        tris = va.tris_setup(g["verts"], g["flags"] if "flags" in g.files else None)
        yield f"fixture {fx}", g["nodes"].view(O.NODE).reshape(-1), g["prim_indices"], O.tris_from_tri64(tris), g["rays"].view(O.RAY).reshape(-1), tris
    # seeded soups (degenerate + duplicate triangles, windows, zero direction components)
    for seed in (1, 2, 3, 4):
        verts, flags, org, d, tmin, tmax = W.random_soup(seed)
        tris = va.tris_setup(verts, flags)
        nodes, pidx, otris = tree(tris)
        yield f"soup seed {seed}", nodes, pidx, otris, va.make_rays(org, d, tmin, tmax), tris
    # zero / tiny / non-finite components on the S1k scene (tests/test_gpu_parity.py::test_weird_rays, first part)
    tris = va.tris_setup(W.make_scene("S1k"))
    nodes, pidx, otris = tree(tris)
    rays = np.concatenate([W.sphere_rays(512, 9, origin=(5.0, 6.0, 7.0))] * 8)
    d = rays["dir"]
    d[0:512, 0] = 0.0
    d[512:1024, 1] = -0.0
    d[1024:1536, 2] = 1e-9
    d[1536:2048, :2] = 0.0
    rays["tmin"][2048:2560] = np.nan
    rays["tmax"][2560:3072] = np.nan
    d[3072:3328, 0] = np.nan
    d[3328:3584, 1] = np.inf
    rays["org"][3584:3840, 2] = np.nan
    rays["tmax"][3840:] = 1e-30
    yield "weird rays (zero / -0 / 1e-9 / NaN / inf components) on S1k", nodes, pidx, otris, rays, tris


def measure(name, nodes, pidx, otris, rays, tris64) -> dict:
    ref, ref_st, _, _, _ = O.traverse_batch(nodes, pidx, otris, rays, want_stats=True)
    row = {"workload": name, "rays": int(len(rays)), "hits": int((ref["prim"] != O.MISS).sum()), "variants": {}}
    for alt in O.ALT_NAMES:
        got, got_st, _, _, _ = O.traverse_batch(nodes, pidx, otris, rays, want_stats=True, L=O.alt_lib(alt))
        row["variants"][alt] = diff(ref, ref_st, got, got_st)
    if not name.startswith("fixture"):
        pn, pp, po = tree(tris64, "ploc")
        got, got_st, _, _, _ = O.traverse_batch(pn, pp, po, rays, want_stats=True)
        row["variants"]["TREE_PLOC"] = diff(ref, ref_st, got, got_st)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default=os.path.join(GOLD, "recall_sensitivity.json"))
    args = ap.parse_args()
    assert O.lib().vto_alt_mask() == 0
    t0 = time.time()
    rows = []
    totals = {k: {"rays": 0, "miss_flip": 0, "prim": 0, "t": 0, "uv": 0, "counters_only": 0} for k in list(O.ALT_NAMES) + ["TREE_PLOC"]}
    import itertools
    for wl in itertools.chain(big_workloads(args.quick), small_workloads()):
        name = wl[0]
        row = measure(*wl)
        for k, dct in row["variants"].items():
            totals[k]["rays"] += row["rays"]
            for f in ("miss_flip", "prim", "t", "uv", "counters_only"):
                totals[k][f] += dct[f]
        rows.append(row)
        print(f"[{time.time() - t0:6.1f}s] {name}: {row['rays']} rays, {row['hits']} hits", flush=True)
        for k, dct in row["variants"].items():
            print(f"    {k:15s} flip {dct['miss_flip']:6d}  prim {dct['prim']:6d}  t {dct['t']:6d}  uv {dct['uv']:6d}  counters-only {dct['counters_only']:8d}"
                  f"  steps x{dct['steps_ratio']:.4f}  tests x{dct['tests_ratio']:.4f}", flush=True)
    out = {"generator": "scripts/recall_sensitivity.py" + (" --quick" if args.quick else ""),
           "what": "rays whose result changes when ONE recalled bvh-v1 detail of oracle/vt_oracle.c is read the other way (VTO_ALT_<X>); "
                   "TREE_PLOC = same rays, reference's builder algorithm instead of the default binned SAH",
           "shipped_vs_alternative": {k: {"shipped": v[0], "alternative": v[1]} for k, v in READINGS.items()},
           "totals": totals, "workloads": rows}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(f"wrote {args.out} in {time.time() - t0:.0f} s")
    print("\n| if upstream differs in | rays | hit<->miss | index | t | u,v | counters only |\n|---|---|---|---|---|---|---|")
    for k, tt in totals.items():
        print(f"| {k} | {tt['rays']} | {tt['miss_flip']} | {tt['prim']} | {tt['t']} | {tt['uv']} | {tt['counters_only']} |")


if __name__ == "__main__":
    main()
