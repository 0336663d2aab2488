#!/usr/bin/env python3
"""bench.py -- Mrays/s closest-hit on the 1M-triangle scene (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload = "S1M_bounce16M", BASELINE.json configs[2]): scene S1M
(1 000 300 triangles), 16 777 216 incoherent cosine-hemisphere bounce rays generated from
the 4096x4096 primary hits of one camera, closest hit.  One "step" = one pass of the hot
path (vt_trace_closest_dev) over the whole ray batch, rays and hits resident in HBM.
With N > 1 ranks the BVH is replicated, every rank traces its own 16 Mi-ray batch (camera =
rank; weak scaling) and the hit records are gathered to rank 0 over RCCL inside the step.

The printed JSON line also carries
  roofline     -- algorithmic bytes (32+16+64*steps+64*tests per ray, counters from the
                  device stats kernel, cross-checked with the CPU oracle on the sample)
                  / mean kernel time (HIP events on the launch stream) vs the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (oracle/, a port: the reference itself cannot be built
                  here) on all host cores over a bounded sample of the same rays.
Data is synthetic (seeded generator, vistrace_amd/workloads.py); nothing reads /root/reference.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="S1M")
    ap.add_argument("--side", type=int, default=4096, help="primary image side; rays per GPU = side*side")
    ap.add_argument("--kind", default="bounce", choices=["bounce", "primary"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--builder", default="ploc", choices=["ploc", "sah"],
                    help="ploc = the reference's build pipeline (default); sah = opt-in binned SAH (not the headline)")
    ap.add_argument("--gen", default="device", choices=["device", "host"], help="where the synthetic rays are generated")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one GPU per rank); gloo = test mode: ranks may share a GPU, hits gathered via host")
    ap.add_argument("--force-dist", action="store_true",
                    help="dev: run the N > 1 control flow (process group, gather pipeline, barriers) even with one rank")
    ap.add_argument("--reserve-cus", type=int, default=32,
                    help="N > 1: CUs (one per shader engine of every XCD) on which the persistent trace grid leaves room "
                         "for the RCCL gather's kernels, so that the transfer of batch b overlaps the trace of batch b+1; "
                         "0 = off (the gather then only starts when the resident grid drains)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="N > 1: launches per batch in the trace/gather pipeline (1 = whole batch per launch: every extra "
                         "launch costs ~0.27 ms of ramp-up and end-of-queue tail; batches are double-buffered either way)")
    ap.add_argument("--mode", default=None, choices=[None, "persistent", "static"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd.distributed import HitGatherPipeline
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, HIT_ATTRS, RAY, RAY_STATS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the traversal has no CPU path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    # ---- scene: CPU build once, upload once (Rebuild) -----------------------------------
    t0 = time.time()
    verts = W.make_scene(args.scene)
    tris = va.tris_setup(verts)
    t1 = time.time()
    host_threads = max(1, len(os.sched_getaffinity(0)) // max(1, world))   # explicit: launchers may export OMP_NUM_THREADS=1
    bvh = va.HostBvh(tris, nthreads=min(16, host_threads), builder=args.builder)   # ranks build side by side
    t2 = time.time()
    host_scene = va.HostScene(bvh)
    engine = va.Engine(dev_index)
    if args.mode is not None:
        engine.set_option("persistent", 1 if args.mode == "persistent" else 0)
    scene = va.Scene(engine, host_scene)
    t3 = time.time()
    log(f"[bench] scene {args.scene}: {len(tris)} tris, {host_scene.pair_count} pairs, depth {host_scene.max_depth}, "
        f"{scene.device_bytes / 1e6:.1f} MB on device; gen {t1 - t0:.2f}s build {t2 - t1:.2f}s upload {t3 - t2:.2f}s")

    # ---- rays: primary pass on the GPU, bounce rays from its hit records ----------------
    side = args.side
    n = side * side
    cams = W.camera_positions(args.scene)
    cam = cams[rank % len(cams)]
    stream0 = tp.current_stream_handle(device)
    rays_host = None
    if args.gen == "host":
        prim_rays = W.primary_rays(side, side, pos=cam)
        d_prim = tp.to_device(prim_rays, device)
    else:   # rays are generated on the device (vt_gen_primary_dev / vt_gen_bounce_dev)
        d_prim = tp.empty_records(n, RAY, device)
        engine.gen_primary_dev(side, side, d_prim.data_ptr(), pos=tuple(float(x) for x in cam), stream=stream0)
    if args.kind == "primary":
        d_rays = d_prim
        if args.gen == "host":
            rays_host = prim_rays
    else:
        d_hits0 = tp.trace_closest(scene, d_prim, n)
        d_attrs = tp.hit_attrs(scene, d_prim, d_hits0, n)
        miss = n - int(d_attrs.view(torch.int32).view(n, 16)[:, 15].sum().item())
        seed = W.SEED + 3 + 1000 * rank
        if args.gen == "host":
            rays_host = W.bounce_rays(tp.to_host(d_attrs, HIT_ATTRS), seed)
            d_rays = tp.to_device(rays_host, device)
        else:
            d_rays = tp.empty_records(n, RAY, device)
            engine.gen_bounce_dev(d_attrs.data_ptr(), n, seed, d_rays.data_ptr(), stream=stream0)
        del d_attrs, d_hits0, d_prim
        log(f"[bench] bounce rays: {n} from {side}x{side} primary hits ({miss} primary misses"
            f"{' re-filled' if args.gen == 'host' else ' -> null rays'}), generated on the {args.gen}")
    d_hits = tp.empty_records(n, HIT, device)
    pipe = HitGatherPipeline(n, device, nchunks=args.chunks, via_host=args.backend == "gloo") if dist_on else None
    t4 = time.time()
    log(f"[bench] ray set-up {t4 - t3:.2f}s")

    # ---- algorithmic bytes: exact counters from the stats kernel (untimed) --------------
    _, d_stats = tp.trace_stats(scene, d_rays, n)
    torch.cuda.synchronize(device)
    stats = tp.to_host(d_stats, RAY_STATS)
    tot_steps = int(stats["steps"].sum(dtype=np.uint64))
    tot_tests = int(stats["tests"].sum(dtype=np.uint64))
    del d_stats
    alg_bytes = n * (32 + 16) + 64 * (tot_steps + tot_tests)
    log(f"[bench] steps/ray {tot_steps / n:.2f} tests/ray {tot_tests / n:.2f} -> {alg_bytes / n:.0f} B/ray algorithmic")

    # ---- timed region -------------------------------------------------------------------
    engine.set_timing(True)
    if dist_on and args.reserve_cus > 0:
        # the gather of batch b runs while batch b+1 is traced: its kernels need somewhere to run
        try:
            engine.set_option("reserved_cus", args.reserve_cus)
        except Exception as exc:   # never lose the run over an optimisation: trace without the reservation
            log(f"[bench] reserved_cus not available ({exc}); the gather will not overlap the resident trace grid")
        log(f"[bench] {engine.get_option('reserved_cus')} CUs keep room for the collective "
            f"({engine.get_option('reserved_limit')} trace blocks each instead of {engine.launch_info()['blocks'] // max(1, engine.get_option('cu_count'))})")

    stream = tp.current_stream_handle(device)

    def trace_chunk(hits_buf, lo, hi):
        scene.trace_closest_dev(d_rays.data_ptr() + lo * RAY.itemsize, hi - lo, hits_buf.data_ptr() + lo * HIT.itemsize, stream)

    def step():
        if not dist_on:
            tp.trace_closest(scene, d_rays, n, d_hits)
        else:
            # the single exchange of the path: hit records -> rank 0 (RCCL gather over xGMI).  Chunked and
            # double-buffered: the gather of one chunk/batch overlaps the tracing of the next
            pipe.submit(trace_chunk)

    # dominant-kernel time for the roofline: whole-batch launches bracketed by HIP events
    pre_ms = []
    for _ in range(2):
        tp.trace_closest(scene, d_rays, n, d_hits)
        pre_ms.append(engine.last_kernel_ms())
    for _ in range(args.warmup):
        step()
    if pipe is not None:
        pipe.drain()
    kernel_ms = []
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    start = time.perf_counter()
    for _ in range(args.steps):
        step()
        if not dist_on:
            kernel_ms.append(engine.last_kernel_ms())  # HIP events on the launch stream
    if pipe is not None:
        pipe.drain()                                   # every hit record has reached rank 0
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - start
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    engine.set_timing(False)

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n * args.steps / elapsed / 1e6
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float(pre_ms[-1])
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    traffic = None
    try:   # HBM-side bytes per launch from the committed rocprofv3 PMC pass of this same command
        with open(os.path.join(ROOT, "profiles", "r1", "traffic.json")) as f:
            tj = json.load(f)
        if tj.get("workload") == f"{args.scene}_{args.kind}{n}" and args.builder == "ploc":
            traffic = tj.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass

    result = {
        "metric": "Mrays/s closest-hit, 1M-triangle scene",
        "value": round(value, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded; rays generated on the %s)" % args.gen,
        "config": {
            "workload": f"{args.scene}_{args.kind}{n}",
            "scene_triangles": int(len(tris)),
            "bvh_builder": "PLOC r=14 + SAH leaf collapse (reference pipeline)" if args.builder == "ploc" else "binned SAH (opt-in)",
            "rays_per_gpu": n,
            "query": "closest-hit",
            "ray_kind": "cosine-hemisphere bounce (incoherent)" if args.kind == "bounce" else "pinhole primary",
            "parallelism": f"rays sharded x{world}, BVH replicated" + (f", RCCL gather of hits to rank 0 ({args.chunks} chunks per batch, double-buffered, overlapped with tracing; {engine.get_option('reserved_cus')} CUs keep room for its kernels)" if world > 1 else ""),
            "kernel_mode": ("persistent" + ("+lds-dma-fetch" if engine.get_option("last_fetch_dma") else "")) if engine.get_option("last_persistent") else "static",
            "launch_options": {k: engine.get_option(k) for k in ("lds_entries", "blocks_per_cu", "block_rays", "refill_threshold", "tri_threshold", "reserved_cus", "reserved_limit")},
            "launch": engine.launch_info(),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": "profiles/r1/traffic.json: FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc pass of this command (factor calibrated: profiles/r1/calib_fetch.txt)" if traffic else None,
            "kernel": "vt::trace_kernel<false,false,%s,%s,false>" % (   # <ANY_HIT, STATS, PERSISTENT, FETCH_DMA, ALPHA>
                "true" if engine.get_option("last_persistent") else "false",
                "true" if engine.get_option("last_fetch_dma") else "false"),
            "kernel_ms": round(k_ms, 4),
            "alg_bytes_per_ray": round(alg_bytes / n, 1),
            "steps_per_ray": round(tot_steps / n, 2),
            "tests_per_ray": round(tot_tests / n, 2),
        },
    }

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only) -------------------
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import binding as O
        if rays_host is None:
            rays_host = tp.to_host(d_rays, RAY)
        nodes = bvh.nodes().view(O.NODE)
        pidx = bvh.prim_indices()
        otris = O.tris_from_tri64(tris)
        pilot = min(n, 1 << 17)
        # the CPU gets its best thread count: all logical CPUs or one per physical core (SMT often hurts this walk)
        rate, cpu_threads = 0.0, host_threads
        for cand in sorted({host_threads, max(1, host_threads // 2)}, reverse=True):
            O.traverse_batch(nodes, pidx, otris, rays_host[:pilot], nthreads=cand)           # warm-up
            tp0 = time.perf_counter()
            O.traverse_batch(nodes, pidx, otris, rays_host[:pilot], nthreads=cand)
            r = pilot / (time.perf_counter() - tp0)
            if r > rate:
                rate, cpu_threads = r, cand
        sample = int(min(n, max(pilot, rate * args.cpu_seconds)))
        sample = max(4096, (sample // 4096) * 4096) if n >= 4096 else n
        tc0 = time.perf_counter()
        ref, ref_stats, s_steps, s_tests, threads = O.traverse_batch(nodes, pidx, otris, rays_host[:sample], want_stats=True,
                                                                    nthreads=cpu_threads)
        cpu_s = time.perf_counter() - tc0
        # one host thread: the closest analogue of what a GLua script gets today (one ray per call, serial)
        t10 = time.perf_counter()
        O.traverse_batch(nodes, pidx, otris, rays_host[:pilot], nthreads=1)
        one_thread = pilot / (time.perf_counter() - t10) / 1e6
        gpu = tp.to_host(d_hits[: sample * HIT.itemsize], HIT)
        same_prim = bool((gpu["prim"] == ref["prim"]).all())
        same_tuv = all(bool((gpu[k].view(np.uint32) == ref[k].view(np.uint32)).all()) for k in ("t", "u", "v"))
        same_stats = bool((stats[:sample]["steps"] == ref_stats[:, 0]).all() and (stats[:sample]["tests"] == ref_stats[:, 1]).all())
        cpu_model = ""
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        result["cpu_baseline"] = {
            "value": round(sample / cpu_s / 1e6, 4),
            "unit": "Mrays/s",
            "cores": threads,
            "kind": "port",
            "sample": f"first {sample} rays of the same batch, same tree, OpenMP schedule(dynamic,4096), {cpu_s:.1f}s",
            "cpu": cpu_model,
            "one_thread_value": round(one_thread, 4),
        }
        result["parity_sample"] = {"rays": sample, "prim_bit_exact": same_prim, "tuv_bit_exact": same_tuv,
                                   "counters_equal": same_stats}
        if not (same_prim and same_tuv):
            log("[bench] PARITY FAILURE on the sample")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
