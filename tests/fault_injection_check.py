#!/usr/bin/env python3
"""Child process of tests/test_gpu_fault_injection.py: every device / pinned-host allocation of the library fails in turn.

Environment (set by the test): VT_ENABLE_TEST_HOOKS=1 (arms vt_test_fail_alloc), VT_TEST_ALLOW_DEVICE_ALIASES=1 and
VT_RCCL_LIB = the RCCL test double (the group cases run their three members on device 0).

For every operation below: a clean run counts its allocations (vt_test_alloc_count); then, for k = 1 .. count, the same operation
runs on a FRESH engine with its k-th allocation failing.  Required of every k:
  * the call returns a non-zero status with vt_last_error set (a VisTraceError here) -- or, where the library has a designed
    retry (the shared overflow block of the launch slots, the pageable-copy probe), completes with correct results;
  * nothing crashes; the engine that saw the failure still traces the committed golden scene bit for bit, and an object the
    failed call was working on (scene, batch) is either gone or still answers as before;
  * after the engine is closed the device's free memory is back at its level (hipMemGetInfo).
The reference's convention on its own failure paths is delete-before-throw (source/VisTrace.cpp:782-785,
source/objects/AccelStruct.cpp:186-203, :780).  Prints one line per operation and "fault injection: ok", or raises.

`fault_injection_check.py hip` runs the same operations against the SECOND hook (vt_test_fail_hip): every other HIP call the library
checks -- copies, event / stream calls, launch checks, synchronisations: the VT_HIP sites -- reports a failure in turn instead of
being made.  Same requirements, with one difference the header states: an object that was being updated IN PLACE when the call
failed (refit) is unspecified until the same call succeeds on it, so the check repeats the call and then requires the completed
state.  A failure may also be absorbed where the library has a second way (the group's device-to-device replication falls back to
per-member uploads)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

assert os.environ.get("VT_ENABLE_TEST_HOOKS") == "1"
import vistrace_amd as va
from vistrace_amd import torch_plumbing as tp
from vistrace_amd import workloads as W

L = va._lib.lib
Err = va._lib.VisTraceError
HIP_MODE = len(sys.argv) > 1 and sys.argv[1] == "hip"
arm = L.vt_test_fail_hip if HIP_MODE else L.vt_test_fail_alloc
passed_count = L.vt_test_hip_count if HIP_MODE else L.vt_test_alloc_count
WHAT = "checked HIP calls" if HIP_MODE else "allocations"
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s1k_golden.npz"))
g_tris = va.tris_setup(GOLD["verts"])
g_bvh = va.HostBvh(g_tris, builder="ploc")                      # the fixture's tree (pinned by tests/test_host_build.py)
g_rays = GOLD["rays"].view(va.RAY).reshape(-1)
g_hits = GOLD["hits"].view(va.HIT).reshape(-1)

verts10k = np.ascontiguousarray(W.make_scene("S10k"), np.float32)
tris10k = va.tris_setup(verts10k)
bvh10k = va.HostBvh(tris10k)
a_flags, a_attribs, a_mats, a_texels = W.alpha_test_rig(len(verts10k))
a_tris = va.tris_setup(verts10k, a_flags)
a_bvh = va.HostBvh(a_tris)
skin, skin_base, nmat = W.skinned_rig(len(verts10k))
bones, binds = W.rig_pose(nmat, 1)
rays_small = W.sphere_rays(5000, 3, origin=(1.0, 2.0, 3.0))
rays_big = W.sphere_rays((2 << 20) + 999, 4, origin=(-1.0, 2.0, 3.0))    # above two pipeline chunks: the staged host path
frames10k = W.vertex_frames(verts10k, W.SEED + 31).view(va.TRI_FRAME).reshape(-1)


def free_bytes() -> int:
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(dev)[0]


def golden_ok(eng) -> bool:
    sc = va.Scene.from_tree(eng, g_bvh)
    try:
        got = sc.trace_closest(g_rays)
        return got.tobytes() == g_hits.tobytes()
    finally:
        sc.free()


def same_hits(a, b) -> bool:
    return a.tobytes() == b.tobytes()


class Case:
    """setup(eng) -> ctx (not counted, never fails); run(eng, ctx) -> result (the injected call); check(eng, ctx, result or None)."""

    def __init__(self, name, run, setup=None, check=None, members=1, may_absorb=False):
        # may_absorb: the call has a designed retry on its path (the launch slots first ask for ONE overflow block for all idle
        # slots and fall back to a block of their own), so an injected failure may be absorbed -- with correct results
        self.name, self.run, self.setup, self.check, self.members, self.may_absorb = name, run, setup, check, members, may_absorb


def open_engine(members):
    return va.Engine([0] * members) if members > 1 else va.Engine(0)


cases = []

# ---- engine open ------------------------------------------------------------------------------------------------------------------
cases.append(Case("vt_engine_open", run=None))          # special-cased below: the engine itself is the injected call

# ---- Rebuild's upload: device re-pack (single, and a 3-member group: replicas by device-to-device copies) and the host-linearised path
cases.append(Case("vt_scene_upload_tree", run=lambda e, c: va.Scene.from_tree(e, bvh10k),
                  check=lambda e, c, r: r is None or (same_hits(r.trace_closest(rays_small), c["ref"]), r.free())[0],
                  setup=lambda e: {"ref": REF_SMALL}))
cases.append(Case("vt_scene_upload_tree (3 members)", members=3, run=lambda e, c: va.Scene.from_tree(e, bvh10k),
                  check=lambda e, c, r: r is None or (same_hits(r.trace_closest(rays_small), c["ref"]), r.free())[0],
                  setup=lambda e: {"ref": REF_SMALL}))
cases.append(Case("vt_scene_upload (host-linearised)", run=lambda e, c: va.Scene(e, c["hs"]),
                  check=lambda e, c, r: r is None or (same_hits(r.trace_closest(rays_small), c["ref"]), r.free())[0],
                  setup=lambda e: {"ref": REF_SMALL, "hs": va.HostScene(bvh10k)}))


# ---- alpha tables: a failed call leaves the scene answering as before ---------------------------------------------------------------
def alpha_setup(e):
    sc = va.Scene.from_tree(e, a_bvh)
    return {"scene": sc}


def alpha_run(e, c):
    c["scene"].set_tri_attribs(a_attribs.view(va.TRI_ATTRIBS))
    c["scene"].set_alpha(a_mats.view(va.ALPHA_MATERIAL), a_texels)
    return True


def alpha_check(e, c, r):
    if r is None:                          # failed somewhere: the tables may be incomplete -- completing them must work now
        c["scene"].set_tri_attribs(a_attribs.view(va.TRI_ATTRIBS))
        c["scene"].set_alpha(a_mats.view(va.ALPHA_MATERIAL), a_texels)
    ok = same_hits(c["scene"].trace_closest(rays_small), REF_ALPHA)
    c["scene"].free()
    return ok


cases.append(Case("vt_scene_set_tri_attribs + vt_scene_set_alpha", setup=alpha_setup, run=alpha_run, check=alpha_check))


def alpha_again_setup(e):
    c = alpha_setup(e)
    alpha_run(e, c)
    return c


def alpha_again_run(e, c):                 # replacing existing tables: old ones must survive a failure
    c["scene"].set_alpha(a_mats.view(va.ALPHA_MATERIAL), a_texels)
    return True


def alpha_again_check(e, c, r):
    if r is None and HIP_MODE:             # the swap may have happened and the AlphaRecs not: the scene refuses to trace (loudly) until
        try:                               # the same call succeeds
            c["scene"].trace_closest(rays_small[:10])
        except Err as exc:
            assert "vt_scene_set_alpha" in str(exc), exc
        alpha_again_run(e, c)
    ok = same_hits(c["scene"].trace_closest(rays_small), REF_ALPHA)      # with the old tables (failure) or the new, equal ones
    c["scene"].free()
    return ok


cases.append(Case("vt_scene_set_alpha (replacing tables)", setup=alpha_again_setup, run=alpha_again_run, check=alpha_again_check))


# ---- skinning + refit -------------------------------------------------------------------------------------------------------------
def skin_setup(e):
    return {"scene": va.Scene.from_tree(e, bvh10k)}


def skin_run(e, c):
    c["scene"].set_skin(verts10k, skin, skin_base)
    c["scene"].set_tri_frames(frames10k)
    c["scene"].skin_refit(bones, binds)
    return True


def skin_check(e, c, r):
    if r is None:
        ok = True
        try:                               # whatever was left half-done: the full sequence must work now
            skin_run(e, c)
        except Err as exc:
            print("    skin sequence after a failure:", exc, flush=True)
            ok = False
    else:
        ok = True
    ok = ok and same_hits(c["scene"].trace_closest(rays_small), REF_SKIN)
    c["scene"].free()
    return ok


cases.append(Case("vt_scene_set_skin + set_tri_frames + vt_scene_skin_refit", setup=skin_setup, run=skin_run, check=skin_check))
cases.append(Case("vt_scene_set_skin + set_tri_frames + vt_scene_skin_refit (3 members)", members=3, setup=skin_setup, run=skin_run, check=skin_check))


def refit_run(e, c):
    c["scene"].refit(MOVED)
    return True


def refit_check(e, c, r):
    if r is None and HIP_MODE:             # failed somewhere inside the update: unspecified until the same call succeeds
        r = refit_run(e, c)
    got = c["scene"].trace_closest(rays_small)
    # a refit that could not prepare changed nothing (every member still has the old geometry); a completed one moved all of them
    ok = same_hits(got, REF_MOVED if r is not None else REF_SMALL)
    if len(e_members(e)) > 1:
        big = c["scene"].trace_closest(rays_big)          # sharded over the members: all of them hold the same geometry
        ok = ok and same_hits(big, REF_BIG_MOVED if r is not None else REF_BIG)
    c["scene"].free()
    return ok


def e_members(e):
    return range(e.device_count)


cases.append(Case("vt_scene_refit", setup=skin_setup, run=refit_run, check=refit_check))
cases.append(Case("vt_scene_refit (3 members)", members=3, setup=skin_setup, run=refit_run, check=refit_check))


# ---- host-pointer traces and batch objects --------------------------------------------------------------------------------------------
def trace_setup(e):
    return {"scene": va.Scene.from_tree(e, bvh10k)}


def trace_check_factory(ref_of):
    def chk(e, c, r):
        ok = r is None or same_hits(r, ref_of())
        ok = ok and same_hits(c["scene"].trace_closest(rays_small), REF_SMALL)
        c["scene"].free()
        return ok
    return chk


cases.append(Case("vt_trace_closest (5 000 host rays)", setup=trace_setup, run=lambda e, c: c["scene"].trace_closest(rays_small),
                  check=trace_check_factory(lambda: REF_SMALL), may_absorb=True))
cases.append(Case("vt_trace_closest (2 Mi host rays: staged pipeline)", setup=trace_setup, run=lambda e, c: c["scene"].trace_closest(rays_big),
                  check=trace_check_factory(lambda: REF_BIG), may_absorb=True))


def batch_run(e, c):
    b = c["scene"].trace_batch(rays_big[: (1 << 20) + 5], check_ranges=True, fetch_hits=True)
    hits = b.hits().copy()
    attrs = b.attrs().copy()
    b.free()
    return hits, attrs


def batch_check(e, c, r):
    ok = r is None or same_hits(r[0], REF_BIG[: (1 << 20) + 5])
    ok = ok and same_hits(c["scene"].trace_closest(rays_small), REF_SMALL)
    c["scene"].free()
    return ok


cases.append(Case("vt_batch_trace_closest_ex (+ hits, attrs)", setup=trace_setup, run=batch_run, check=batch_check, may_absorb=True))


def sets_run(e, c):
    bs = c["scene"].trace_batch_set([rays_small, rays_small[:1000], rays_small[2000:]], fetch_hits=True)
    out = [b.hits().copy() for b in bs]
    for b in bs:
        b.free()
    return out


def sets_check(e, c, r):
    ok = r is None or (same_hits(r[0], REF_SMALL) and same_hits(r[1], REF_SMALL[:1000]) and same_hits(r[2], REF_SMALL[2000:]))
    ok = ok and same_hits(c["scene"].trace_closest(rays_small), REF_SMALL)
    c["scene"].free()
    return ok


cases.append(Case("vt_batch_trace_closest_set (one merged launch)", setup=trace_setup, run=sets_run, check=sets_check, may_absorb=True))


# ---- device-resident rays: deep-stack launch scratch, bounce loop, the group's gather ------------------------------------------------
def dev_setup(e):
    e.set_option("lds_entries", 2)                        # most of the stack spills: the launch slots' overflow blocks are allocated
    e.set_option("persistent", 1)
    sc = va.Scene.from_tree(e, bvh10k)
    return {"scene": sc, "d_rays": tp.to_device(rays_small, dev), "d_hits": tp.empty_records(len(rays_small) * 3, va.HIT, dev)}


def dev_run(e, c):
    c["scene"].trace_closest_dev(c["d_rays"].data_ptr(), len(rays_small), c["d_hits"].data_ptr())
    e.synchronize()
    return tp.to_host(c["d_hits"], va.HIT)[: len(rays_small)].copy()


cases.append(Case("vt_trace_closest_dev (stack overflow area)", setup=dev_setup, run=dev_run, check=trace_check_factory(lambda: REF_SMALL),
                  may_absorb=True))


def loop_run(e, c):
    live = c["scene"].bounce_loop_dev(c["d_rays"].data_ptr(), len(rays_small), 3, 77, c["d_hits"].data_ptr())
    e.synchronize()
    return tp.to_host(c["d_hits"], va.HIT).copy(), live


def loop_check(e, c, r):
    ok = r is None or (same_hits(r[0], REF_LOOP[0]) and list(r[1]) == list(REF_LOOP[1]))
    ok = ok and same_hits(c["scene"].trace_closest(rays_small), REF_SMALL)
    c["scene"].free()
    return ok


cases.append(Case("vt_bounce_loop_dev", setup=dev_setup, run=loop_run, check=loop_check, may_absorb=True))


def gather_setup(e):
    sc = va.Scene.from_tree(e, bvh10k)
    n, nd = len(rays_small), e.device_count
    shards = []
    for g in range(nd):
        lo, hi = va.shard_bounds(n, nd, g)
        shards.append(tp.to_device(rays_small[lo:hi], dev))
    cap = va.shard_capacity(n, nd)
    return {"scene": sc, "shards": shards, "out": torch.zeros(nd * cap * 16, dtype=torch.uint8, device=dev)}


def gather_run(e, c):
    c["scene"].trace_closest_gather_dev([s.data_ptr() for s in c["shards"]], len(rays_small), c["out"].data_ptr())
    e.synchronize()
    return tp.to_host(c["out"][: len(rays_small) * 16], va.HIT).copy()


cases.append(Case("vt_trace_closest_gather_dev (3 members)", members=3, setup=gather_setup, run=gather_run,
                  check=trace_check_factory(lambda: REF_SMALL), may_absorb=True))

cases.append(Case("vt_engine_set_option reserved_cus", setup=trace_setup, run=lambda e, c: (e.set_option("reserved_cus", 16), None)[1],
                  check=trace_check_factory(lambda: REF_SMALL)))

# ---- references (no injection) ------------------------------------------------------------------------------------------------------
assert L.vt_test_fail_alloc(0) == 0 and L.vt_test_fail_hip(0) == 0
_e = va.Engine(0)
assert golden_ok(_e), "the product does not reproduce the golden fixture before any injection"
_s = va.Scene.from_tree(_e, bvh10k)
REF_SMALL = _s.trace_closest(rays_small)
REF_BIG = _s.trace_closest(rays_big)
MOVED = (verts10k + np.float32(0.375)).astype(np.float32)
_s.refit(MOVED)
REF_MOVED = _s.trace_closest(rays_small)
REF_BIG_MOVED = _s.trace_closest(rays_big)
_s.free()
_s = va.Scene.from_tree(_e, a_bvh)
alpha_run(_e, {"scene": _s})
REF_ALPHA = _s.trace_closest(rays_small)
_s.free()
_s = va.Scene.from_tree(_e, bvh10k)
skin_run(_e, {"scene": _s})
REF_SKIN = _s.trace_closest(rays_small)
_s.free()
_c = dev_setup(_e)
REF_LOOP = loop_run(_e, _c)
_c["scene"].free()
_e.close()
del _e, _s, _c

total_k = total_failed = 0
for case in cases:
    # warm-up + clean count on a fresh engine
    if case.run is None:
        arm(0)
        va.Engine(0).close()
        count = passed_count()
    else:
        e = open_engine(case.members)
        ctx = case.setup(e) if case.setup else {}
        arm(0)
        r = case.run(e, ctx)
        count = passed_count()
        assert case.check is None or case.check(e, ctx, r), f"{case.name}: the clean run is wrong"
        e.close()
        del e, ctx, r
    level = free_bytes()            # (after the clean run: torch's caching allocator holds what the case's device tensors need)
    failed = survived = 0

    def inject(k):
        """one run with the k-th step failing; returns (reported as an error, absorbed)"""
        if case.run is None:
            arm(k)
            try:
                e = va.Engine(0)
                raise AssertionError(f"vt_engine_open with its {k}-th step failing must fail")
            except Err as exc:
                assert exc.code != 0 and str(exc), "no message"
            arm(0)
            e = va.Engine(0)                       # the next open works
            assert golden_ok(e)
            e.close()
            return 1, 0
        e = open_engine(case.members)
        ctx = case.setup(e) if case.setup else {}
        arm(k)
        r, rep, absorbed = None, 0, 0
        try:
            r = case.run(e, ctx)
            absorbed = 1
            assert case.may_absorb or HIP_MODE, f"{case.name}: allocation {k} failed but the call reported success"
        except Err as exc:
            assert exc.code != 0 and len(str(exc)) > 8, f"{case.name} k={k}: no message"
            rep = 1
        arm(0)
        assert case.check is None or case.check(e, ctx, r), f"{case.name} k={k}: wrong results after the injected failure"
        assert golden_ok(e), f"{case.name} k={k}: the engine no longer reproduces the golden fixture"
        e.close()
        return rep, absorbed

    for k in range(1, count + 1):
        rep, absorbed = inject(k)
        failed += rep
        survived += absorbed
        now = free_bytes()
        if abs(now - level) > (2 << 20):
            # A leak of the library repeats; a block the HIP runtime keeps for itself (staging chunks, signal pools: seen once in
            # a while, a few MB) does not.  The same injection once more: the level may not move again.
            first = now
            inject(k)
            now = free_bytes()
            assert abs(now - first) <= (1 << 20), (f"{case.name} k={k}: device memory not back at its level "
                                                   f"({(level - first) / 1e6:.1f} MB missing, {(first - now) / 1e6:.1f} MB more on the repeat)")
            print(f"    {case.name} k={k}: the free-memory level moved by {(level - first) / 1e6:.1f} MB once and not again on the repeat "
                  f"(the runtime's own blocks); new level", flush=True)
            level = now
    total_k += count
    total_failed += failed
    assert failed + survived == count and (case.may_absorb or HIP_MODE or failed == count), f"{case.name}: {failed} of {count} injected failures surfaced"
    print(f"{case.name}: {count} {WHAT}, {failed} injected failures reported as errors, {survived} absorbed by a designed retry; memory level kept", flush=True)

print(f"fault injection: ok, {total_k} injected failures over {len(cases)} operations, {total_failed} surfaced as a status + message")
