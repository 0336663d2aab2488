"""world_size-2 CPU test (gloo) of the multi-GPU layer: contiguous ray shards, one gather of
hit records to the root.  The per-rank tracer here is the CPU oracle (test infrastructure);
on GPUs the same helpers run over RCCL with the HIP tracer (bench.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_exactly():
    from vistrace_amd.distributed import shard_bounds, shard_capacity
    for n in (0, 1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            cap = shard_capacity(n, world)
            assert cap % 64 == 0 and max(hi - lo for lo, hi in spans) <= cap
            # one rule for the torch path and the native path: the C ABI's (shard g starts at g * capacity)
            assert all(lo == min(n, r * cap) for r, (lo, hi) in enumerate(spans))


def test_python_and_abi_shard_rules_are_the_same_function():
    import vistrace_amd as va
    from vistrace_amd import distributed as D
    for n in (0, 65, 1000, (1 << 24) + 5):
        for world in (1, 2, 3, 8):
            assert D.shard_capacity(n, world) == va.shard_capacity(n, world)
            assert [D.shard_bounds(n, world, r) for r in range(world)] == [va.shard_bounds(n, world, r) for r in range(world)]


def _worker(rank, world, port, n, result_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vistrace_amd as va
        from oracle import binding as O
        from vistrace_amd import workloads as W
        from vistrace_amd.distributed import trace_sharded
        tris = va.tris_setup(W.make_scene("S1k"))           # every rank builds its own replica
        bvh = va.HostBvh(tris)
        nodes, pidx, otris = bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris)
        rays = W.sphere_rays(n, 99, origin=(1.0, 2.0, 3.0))
        rays_t = torch.from_numpy(rays.view(np.uint8).reshape(-1).copy())

        def trace_fn(shard, count):
            r = shard.numpy().view(O.RAY)
            assert len(r) == count
            hits, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, r, nthreads=1)
            return torch.from_numpy(hits.view(np.uint8).reshape(-1).copy())

        got = trace_sharded(trace_fn, rays_t, n)
        if rank == 0:
            full, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, rays, nthreads=1)
            ok = got is not None and got.numpy().tobytes() == full.tobytes()
            open(result_path, "w").write("ok" if ok else "mismatch")
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def _worker_weak(rank, world, port, n, result_path):
    """bench.py's N > 1 step: every rank traces ITS OWN batch in chunks, async gather to rank 0."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import vistrace_amd as va
        from oracle import binding as O
        from vistrace_amd import workloads as W
        from vistrace_amd.distributed import HitGatherPipeline, chunk_bounds, pipelined_trace_gather
        tris = va.tris_setup(W.make_scene("S1k"))
        bvh = va.HostBvh(tris, nthreads=1)
        nodes, pidx, otris = bvh.nodes().view(O.NODE), bvh.prim_indices(), O.tris_from_tri64(tris)

        def rays_of(r):
            return W.sphere_rays(n, 500 + r, origin=(10.0 * r, -5.0, 2.0))
        rays = rays_of(rank)
        hits_local = torch.zeros(n * 16, dtype=torch.uint8)
        recv = [torch.zeros(n * 16, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None

        def trace_chunk(lo, hi):
            h, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, rays[lo:hi], nthreads=1)
            hits_local[lo * 16: hi * 16] = torch.from_numpy(h.view(np.uint8).reshape(-1).copy())

        for _ in range(2):   # buffers are reused across steps
            pipelined_trace_gather(trace_chunk, n, hits_local, recv, nchunks=3)
        assert chunk_bounds(n, 3)[0][0] == 0 and chunk_bounds(n, 3)[-1][1] == n
        # the double-buffered pipeline bench.py uses: three batches, results of the last two checked
        pipe = HitGatherPipeline(n, torch.device("cpu"), nchunks=2)
        batch_rays = [W.sphere_rays(n, 900 + 10 * k + rank, origin=(1.0 * k, 2.0, 3.0)) for k in range(3)]
        used = []
        for k in range(3):
            def tc(buf, lo, hi, k=k):
                h, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, batch_rays[k][lo:hi], nthreads=1)
                buf[lo * 16: hi * 16] = torch.from_numpy(h.view(np.uint8).reshape(-1).copy())
            used.append(pipe.submit(tc))
        pipe.drain()
        assert used == [0, 1, 0]
        if rank == 0:
            ok = True
            for k in (1, 2):
                for r in range(world):
                    full, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, W.sphere_rays(n, 900 + 10 * k + r, origin=(1.0 * k, 2.0, 3.0)), nthreads=1)
                    ok = ok and pipe.recv[used[k]][r].numpy().tobytes() == full.tobytes()
            for r in range(world):
                full, _, _, _, _ = O.traverse_batch(nodes, pidx, otris, rays_of(r), nthreads=1)
                ok = ok and recv[r].numpy().tobytes() == full.tobytes()
            open(result_path, "w").write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_two_rank_weak_scaling_step(tmp_path):
    result = tmp_path / "result.txt"
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker_weak, args=(2, port, 1000, str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"


@pytest.mark.parametrize("n", [1001, 4096])
def test_two_rank_shard_and_gather(tmp_path, n):
    result = tmp_path / "result.txt"
    port = 29500 + (os.getpid() + n) % 2000
    mp.spawn(_worker, args=(2, port, n, str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"


def test_abi_shard_bounds_cover_and_align(va):
    """vt_shard_bounds / vt_shard_capacity (what vt_trace_closest and vt_trace_closest_gather_dev shard by): contiguous,
    disjoint, 64-ray aligned starts, shard g starts at g * capacity -- so gathered hit records land in ray order."""
    for n in (0, 1, 63, 64, 65, 1000, 1 << 20, (1 << 24) + 5, 1 << 27):
        for ndev in (1, 2, 3, 4, 8):
            cap = va.shard_capacity(n, ndev)
            assert cap % 64 == 0 and cap * ndev >= n
            expect = 0
            for g in range(ndev):
                lo, hi = va.shard_bounds(n, ndev, g)
                assert lo == expect == min(n, g * cap) and lo <= hi <= lo + cap
                expect = hi
            assert expect == n


def test_gather_schedule_on_a_simulated_group():
    """The order of traces, event waits and gathers that multi_gpu.hip executes (vistrace_amd/csrc/gather_schedule.h),
    run on a simulated group of 1, 2, 4 and 8 devices with randomly interleaved streams: no send buffer is re-written
    under a gather, no gather starts before its trace, overlap happens (and does not with gather_overlap = 0)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "schedule"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_build", "test_gather_schedule")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


def test_gather_chunk_bounds_tile_a_shard():
    """vt_gather_chunk_bounds (the pieces a shard is traced and gathered in, round 4): contiguous, in order, multiples of 64 records
    except the last, covering the shard exactly -- for every rank the same, since all ranks pass the same count."""
    import vistrace_amd as va
    for count in (0, 1, 63, 64, 65, 1000, 1 << 20, (1 << 24) + 77, va.shard_capacity(134217728, 8)):
        for K in (1, 2, 3, 4, 8, 16):
            covered = 0
            for c in range(K):
                lo, hi = va.gather_chunk_bounds(count, K, c)
                assert lo == covered and lo <= hi <= count
                assert lo % 64 == 0 or lo == count
                covered = hi
            assert covered == count
        assert va.gather_chunk_bounds(count, 4, 4) == (0, 0) and va.gather_chunk_bounds(count, 4, -1) == (0, 0)   # out of range: empty
    # configs[4] over 8 ranks in 4 pieces: every piece 4 Mi rays
    cap = va.shard_capacity(134217728, 8)
    assert [va.gather_chunk_bounds(cap, 4, c) for c in range(4)] == [(k << 22, (k + 1) << 22) for k in range(4)]
