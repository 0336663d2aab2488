#!/usr/bin/env python3
"""Child process of tests/test_gpu_fake_group.py: the N > 1 control flow of the product on ONE GPU.

Environment (set by the test): VT_RCCL_LIB = tests/cpp/_build/libfake_rccl.so (a test double: transfers between ranks are
stream-ordered device copies, see tests/cpp/fake_rccl.cpp) and VT_TEST_ALLOW_DEVICE_ALIASES=1 (device 0 stands for every member
of a group).  Everything else is the shipped library: vt_engine_open_multi, scene replication, the per-device host threads,
vt_trace_closest_gather_dev's schedule (double-buffered send buffers, K pieces per batch), and the one-process-per-GPU form
(vt_engine_comm_init_rank + vt_gather_hits[_part]_dev) with one thread per rank as bench.py --gpus N runs one process per rank.
Every result is compared with the CPU oracle bit for bit.  Prints "fake group: ok ..." and exits 0, or raises."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

assert os.environ.get("VT_RCCL_LIB", "").endswith("libfake_rccl.so") and os.environ.get("VT_TEST_ALLOW_DEVICE_ALIASES") == "1"
import vistrace_amd as va
from oracle import binding as O
from vistrace_amd import torch_plumbing as tp
from vistrace_amd import workloads as W

dev = torch.device("cuda", 0)
verts = np.ascontiguousarray(W.make_scene("S10k"), np.float32)
tris = va.tris_setup(verts)
bvh = va.HostBvh(tris)
host_scene = va.HostScene(bvh)
otris = O.tris_from_tri64(tris)


def oracle(rays):
    return O.traverse_batch(bvh.nodes().view(O.NODE), bvh.prim_indices(), otris, rays)[0]


def same(got, ref):
    eq = (got.view(np.uint8).reshape(-1, 16) == ref.view(np.uint8).reshape(-1, 16)).all(axis=1)
    if not eq.all():                           # where: runs of differing records
        bad = np.nonzero(~eq)[0]
        cuts = np.nonzero(np.diff(bad) > 1)[0]
        runs = [(int(a), int(b) + 1) for a, b in zip(np.concatenate([[bad[0]], bad[cuts + 1]]), np.concatenate([bad[cuts], [bad[-1]]]))]
        print(f"mismatch: {len(bad)} of {len(eq)} records differ, runs {runs[:12]}{' ...' if len(runs) > 12 else ''}; first got {got[bad[0]]} expected {ref[bad[0]]}", flush=True)
    return bool(eq.all())


checks = 0
ROUND = int(os.environ.get("FAKE_GROUP_ROUND", "0"))      # scripts/fake_group_soak.sh: other sizes and rays per round
rng = np.random.default_rng(1000 + ROUND)

# ---- one process, N devices: vt_engine_open_multi + vt_trace_closest_gather_dev -------------------------------------------------
n = 300_007 if ROUND == 0 else int(rng.integers(5_000, 700_000))   # ragged: shards of different sizes, the last one short of its capacity
NB = 5                                        # batches in flight back to back, each with its OWN rays: a send buffer that is
batch_rays = [W.sphere_rays(n, 5 + j + 10 * ROUND, origin=(4.0 - j, 5.0, 6.0 + j)) for j in range(NB)]   # reused too early shows as a mismatch
batch_ref = [oracle(r) for r in batch_rays]
rays, ref = batch_rays[0], batch_ref[0]
for ndev in (2, 3, 4, 8):
    eng = va.Engine([0] * ndev)
    assert eng.device_count == ndev
    eng.set_option("persistent", 1 if ndev in (3, 8) else 2)      # persistent waves on the small shards too
    # replicated to every member: re-packed on each member's device (vt_scene_upload_tree) or uploaded from the host lineariser's output
    scene = va.Scene.from_tree(eng, bvh) if ndev in (3, 8) else va.Scene(eng, host_scene)
    # ... by device-to-device copies of the root's finished records, in phases: every member's copies were enqueued before the
    # first member was waited for (engine.hip: scene_replicate)
    assert eng.get_option("last_update_members") == ndev and eng.get_option("last_update_early_waits") == 0
    cap = va.shard_capacity(n, ndev)
    shards, ptrs = [], []
    for j in range(NB):
        per_dev = []
        for g in range(ndev):
            lo, hi = va.shard_bounds(n, ndev, g)
            per_dev.append(tp.to_device(batch_rays[j][lo:hi], dev) if hi > lo else None)
        shards.append(per_dev)
        ptrs.append([s.data_ptr() if s is not None else 0 for s in per_dev])
    outs = [torch.zeros(ndev * cap * 16, dtype=torch.uint8, device=dev) for _ in range(NB)]
    for K in (1, 2, 4, 3):
        eng.set_option("gather_chunks", K)
        for o in outs:
            o.fill_(0xEE)
        torch.cuda.synchronize()               # the fills run on torch's stream, the group on the engines' own streams
        for j, o in enumerate(outs):           # five batches back to back: both send buffers of every peer are reused
            scene.trace_closest_gather_dev(ptrs[j], n, o.data_ptr())
        eng.synchronize()
        for j, o in enumerate(outs):
            if not same(tp.to_host(o[: n * 16], va.HIT), batch_ref[j]):
                print(f"n {n} ndev {ndev} cap {cap} K {K} batch {j}; shards {[va.shard_bounds(n, ndev, g) for g in range(ndev)]}; pieces {[va.gather_chunk_bounds(cap, K, c) for c in range(K)]}", flush=True)
                raise AssertionError(f"gather_dev ndev {ndev} K {K} batch {j}")
            checks += 1
    # with room reserved for a collective's kernels beside the persistent grids
    eng.set_option("gather_chunks", 2)
    eng.set_option("reserved_cus", 32)
    outs[0].fill_(0xEE)
    torch.cuda.synchronize()
    scene.trace_closest_gather_dev(ptrs[0], n, outs[0].data_ptr())
    eng.synchronize()
    assert same(tp.to_host(outs[0][: n * 16], va.HIT), ref), f"gather_dev reserved CUs ndev {ndev}"
    eng.set_option("reserved_cus", 0)
    checks += 1
    # host rays: one staging pipeline per member, side by side on their own threads (needs >= 1 Mi rays)
    if ndev in (2, 4):
        big = W.sphere_rays((1 << 20) + 4099, 9, origin=(-3.0, 2.0, 1.0))
        big_ref = oracle(big)
        assert same(scene.trace_closest(big), big_ref), f"host rays ndev {ndev}"
        assert (scene.trace_any(big) == (big_ref["prim"] != O.MISS)).all()
        checks += 2
    scene.free()
    eng.close()

# ---- calls that rewrite a scene reach every member: host rays (>= 1 Mi: sharded over the members) see the same scene everywhere --
eng = va.Engine([0, 0, 0])
flags, attribs, mats, texels = W.alpha_test_rig(len(verts))
atris = va.tris_setup(verts, flags)
abvh = va.HostBvh(atris)
scene = va.Scene(eng, va.HostScene(abvh))
scene.set_tri_attribs(attribs.view(va.TRI_ATTRIBS))
scene.set_alpha(mats.view(va.ALPHA_MATERIAL), texels)
big = W.sphere_rays((1 << 20) + 777, 21 + ROUND, origin=(2.0, -3.0, 4.0))


def alpha_oracle(tris64, rays_):
    ot = O.tris_from_tri64(tris64)
    try:
        O.set_alpha(ot, attribs["uv"].reshape(len(verts), 6), attribs["material"], mats.view(O.ALPHA_MATERIAL), texels)
        return O.traverse_batch(abvh.nodes().view(O.NODE), abvh.prim_indices(), ot, rays_)[0]
    finally:
        O.set_alpha()


assert same(scene.trace_closest(big), alpha_oracle(atris, big)), "alpha tables on every member"
moved = (verts + np.float32(0.375)).astype(np.float32)
scene.refit(moved, flags)                                        # vt_scene_refit on every member
mtris = va.tris_setup(moved, flags)
abvh.refit(mtris)
assert same(scene.trace_closest(big), alpha_oracle(mtris, big)), "refit on every member"
# ... in phases: no member was waited for before the LAST member's work had been enqueued (engine.hip: update_every_member)
assert eng.get_option("last_update_members") == 3 and eng.get_option("last_update_early_waits") == 0
# a refit that is refused (non-finite vertices) leaves EVERY member refusing to trace, not just the first one asked
bad = moved.copy(); bad[5, 1, 1] = np.nan
try:
    scene.refit(bad, flags)
    raise AssertionError("a refit with a NaN vertex must fail")
except va._lib.VisTraceError:
    pass
for rays_ in (big, big[:1000]):                                  # through the members, and through the root alone
    try:
        scene.trace_closest(rays_)
        raise AssertionError("a poisoned group must refuse to trace")
    except va._lib.VisTraceError:
        pass
scene.refit(moved, flags)                                        # a clean refit heals every member
assert same(scene.trace_closest(big), alpha_oracle(mtris, big)), "clean refit after a refused one"
checks += 5
scene.free()
eng.close()
# skinning: matrices to every member, positions and vertex frames follow
eng = va.Engine([0, 0])
pbvh = va.HostBvh(tris)
scene = va.Scene(eng, va.HostScene(pbvh))
skin, base, nmat = W.skinned_rig(len(verts))
scene.set_skin(verts, skin, base)
bones, binds = W.rig_pose(nmat, 1 + ROUND)
scene.skin_refit(bones, binds)
posed = O.skin_verts(verts.reshape(len(verts), 9), skin, base, O.skin_matrices(bones, binds)).reshape(len(verts), 3, 3)
ptris = va.tris_setup(posed)
pbvh.refit(ptris)
pref = O.traverse_batch(pbvh.nodes().view(O.NODE), pbvh.prim_indices(), O.tris_from_tri64(ptris), big)[0]
assert same(scene.trace_closest(big), pref), "skin refit on every member"
assert eng.get_option("last_update_members") == 2 and eng.get_option("last_update_early_waits") == 0
checks += 1
scene.free()
eng.close()

# ---- two host threads on one group: Rebuilds of one scene beside refits of another (advisor, round 5: vt_scene_upload_tree held the
# root's lock while it asked its peers, the group-wide refits hold every member's lock, peers first -- opposite orders; the upload
# now lets go of the root before it goes to the peers).  Both threads must get through, with right answers. ----------------------
import time
eng = va.Engine([0, 0, 0])
keep = va.Scene.from_tree(eng, bvh)
small = big[:20000]
ref_small = oracle(small)
moved_v = (verts + np.float32(0.25)).astype(np.float32)
moved_b = va.HostBvh(tris)
moved_b.refit(va.tris_setup(moved_v))
ref_moved = O.traverse_batch(moved_b.nodes().view(O.NODE), moved_b.prim_indices(), O.tris_from_tri64(va.tris_setup(moved_v)), small)[0]
stop_at = time.time() + 3.0
errors, rounds = [], {"rebuild": 0, "refit": 0}


def rebuilds():
    try:
        while time.time() < stop_at and not errors:
            sc = va.Scene.from_tree(eng, bvh)                      # vt_scene_upload_tree on the root, replicas on the peers
            if not same(sc.trace_closest(small), ref_small):
                errors.append("a scene rebuilt beside a refit answers wrongly")
            sc.free()
            rounds["rebuild"] += 1
    except Exception as exc:                                        # noqa: BLE001
        errors.append(f"rebuild thread: {exc}")


def refits():
    try:
        k = 0
        while time.time() < stop_at and not errors:
            keep.refit(moved_v if k % 2 == 0 else verts)           # every member's lock at once
            if not same(keep.trace_closest(small), ref_moved if k % 2 == 0 else ref_small):
                errors.append("a scene refitted beside a Rebuild answers wrongly")
            k += 1
            rounds["refit"] += 1
    except Exception as exc:                                        # noqa: BLE001
        errors.append(f"refit thread: {exc}")


pair = [threading.Thread(target=rebuilds, daemon=True), threading.Thread(target=refits, daemon=True)]      # (daemon: a deadlocked pair must not keep the process)
for t in pair:
    t.start()
for t in pair:
    t.join(60)
    if t.is_alive():
        print("Rebuild beside refit on a group: a thread hangs (lock order)", flush=True)
        os._exit(3)
assert not errors and rounds["rebuild"] >= 3 and rounds["refit"] >= 3, (errors, rounds)
checks += 2
keep.free()
eng.close()

# ---- one engine per rank, one thread per rank: vt_engine_comm_init_rank + vt_gather_hits[_part]_dev -------------------------------
for nranks in (2, 4):
    uid = va.comm_unique_id()
    cap = va.shard_capacity(n, nranks)
    recv = [torch.zeros(nranks * cap * 16, dtype=torch.uint8, device=dev) for _ in range(4)]    # one receive buffer per batch, two send buffers per rank
    errors = []
    barrier = threading.Barrier(nranks)

    def rank_main(rank):
        try:
            eng = va.Engine(0)
            eng.comm_init_rank(nranks, rank, uid)
            eng.set_option("reserved_cus", 32)
            scene = va.Scene(eng, host_scene)
            lo, hi = va.shard_bounds(n, nranks, rank)
            d_batch = [tp.to_device(batch_rays[j][lo:hi], dev) for j in range(4)]
            stream = torch.cuda.Stream(dev)
            sh = stream.cuda_stream
            send = [torch.zeros(cap * 16, dtype=torch.uint8, device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            for K in (1, 4, 2):
                for batch in range(4):
                    k = batch % 2
                    d_rays = d_batch[batch]
                    eng.gather_wait(1, sh)                                 # the gather of two batches ago has read send[k]
                    # 4 ranks: the root traces straight into its slice of the result, as bench.py's NativeGather does (in place:
                    # d_send == d_recv_root + root * count records, no local copy); 2 ranks: from a send buffer of its own
                    mine = recv[batch] if (rank == 0 and nranks == 4) else send[k]
                    for c in range(K):
                        clo, chi = va.gather_chunk_bounds(cap, K, c)
                        m = min(chi, hi - lo) - clo
                        if m > 0:
                            scene.trace_closest_dev(d_rays.data_ptr() + 32 * clo, m, mine.data_ptr() + 16 * clo, sh)
                        if K == 1:
                            eng.gather_hits_dev(mine.data_ptr(), cap, recv[batch].data_ptr() if rank == 0 else 0, 0, sh)
                        else:
                            eng.gather_hits_part_dev(mine.data_ptr(), cap, c, K, recv[batch].data_ptr() if rank == 0 else 0, 0, sh)
                eng.gather_wait(0)
                torch.cuda.synchronize()
                barrier.wait()
                if rank == 0:
                    for j in range(4):
                        if not same(tp.to_host(recv[j][: n * 16], va.HIT), batch_ref[j]):
                            errors.append(f"per-rank gather nranks {nranks} K {K} batch {j}")
                        recv[j].fill_(0xEE)
                    torch.cuda.synchronize()
                barrier.wait()
            scene.free()
            eng.close()
        except Exception as exc:                                            # noqa: BLE001
            errors.append(f"rank {rank}: {exc!r}")
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
        assert not t.is_alive(), "a rank thread hangs"
    assert not errors, errors
    checks += 12

print(f"fake group: ok, {checks} checks (round {ROUND}, {n} rays per batch)")
