"""The secondary figures of a bench.py line (never `value`): what the other builder's tree is worth, two streams, merged launches,
the transfer-inclusive rate of SURVEY.md 8(d), the scene beyond every cache, what a Rebuild costs.

bench.py's main() calls run(ctx) behind its timed region, with its own names in `ctx` (a SimpleNamespace of main's locals and
bench.py's helpers).  Each leg reads what it needs from there, adds one key to ctx.result, and never costs the headline line: a
failure is logged, not raised.  Moved out of bench.py in round 6 so that the timed region can be found without reading through
250 lines of legs; the bodies are unchanged.  `--legs all | host | off` and `--rebuild-leg` / `--alt-builder` select them as before."""
import argparse
import time

import numpy as np

def alt_builder(c) -> None:
    (BUILDER_NAMES, HIT, W, any_hit, apply_image_hint, args, build_scene, dev_index, device, dist_on, log, make_rays,
    n, rank, result, torch, tp, va, world) = (
        c.BUILDER_NAMES, c.HIT, c.W, c.any_hit, c.apply_image_hint, c.args, c.build_scene, c.dev_index, c.device,
        c.dist_on, c.log, c.make_rays, c.n, c.rank, c.result, c.torch, c.tp, c.va, c.world)
    # ---- the same workload on the other builder's tree (N = 1): what the tree is worth ------------------------------------
    if rank == 0 and world == 1 and not dist_on and args.alt_builder not in ("none", args.builder) and n > 0 and not any_hit and args.alpha_frac == 0:
        try:
            alt_args = argparse.Namespace(**vars(args))
            alt_args.builder = args.alt_builder
            a_tris, a_bvh, a_hs, a_engine, a_scene, _ = build_scene(alt_args, va, W, dev_index, world)
            a_rays, a_n, _, _, _ = make_rays(alt_args, rank, world, va, W, tp, a_engine, a_scene, device)
            apply_image_hint(alt_args, a_engine)
            a_hits = tp.empty_records(a_n, HIT, device)
            _, a_stats = tp.trace_stats(a_scene, a_rays, a_n)
            a_st = a_stats.view(torch.int32).view(a_n, 2).sum(dim=0, dtype=torch.int64).cpu().numpy() / a_n
            del a_stats
            for _ in range(3):
                tp.trace_closest(a_scene, a_rays, a_n, a_hits)
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            alt_steps = max(10, min(args.steps, 50))
            e0.record()
            for _ in range(alt_steps):
                tp.trace_closest(a_scene, a_rays, a_n, a_hits)
            e1.record()
            torch.cuda.synchronize(device)
            a_ms = e0.elapsed_time(e1) / alt_steps
            result["alt_builder"] = {
                "bvh_builder": BUILDER_NAMES[args.alt_builder],
                "value": round(a_n / (a_ms * 1e-3) / 1e6, 2), "unit": "Mrays/s", "kernel_ms": round(a_ms, 4),
                "steps_per_ray": round(float(a_st[0]), 2), "tests_per_ray": round(float(a_st[1]), 2),
                "note": "same rays procedure, same kernel; kernel time is proportional to steps per ray (profiles/r2/notes.md)",
            }
            del a_rays, a_hits, a_scene, a_engine
        except Exception as exc:   # a secondary figure must never cost the headline line
            log(f"[bench] alt_builder leg failed: {exc}")


def two_streams(c) -> None:
    (any_hit, args, d_hits, d_rays, device, dist_on, log, n, rank, result, scene, torch, under_profiler, world) = (
        c.any_hit, c.args, c.d_hits, c.d_rays, c.device, c.dist_on, c.log, c.n, c.rank, c.result, c.scene, c.torch,
        c.under_profiler, c.world)
    # ---- independent batches on two streams (rank 0, N = 1): an extra figure, never `value` ---------------------------------
    # A launch ends with ~0.3 ms of drain (profiles/r3/notes.md section 6); a caller whose batches are independent can hide it by
    # alternating between two streams -- the next grid's blocks move in as this one's leave.  Reported beside the serial figure.
    if rank == 0 and world == 1 and not dist_on and n > 0 and not under_profiler():
        try:
            two = [torch.cuda.Stream(device=device) for _ in range(2)]
            buf2 = [d_hits, torch.empty_like(d_hits)]

            def launch_on(k):
                if any_hit:
                    scene.trace_any_dev(d_rays.data_ptr(), n, buf2[k % 2].data_ptr(), two[k % 2].cuda_stream)
                else:
                    scene.trace_closest_dev(d_rays.data_ptr(), n, buf2[k % 2].data_ptr(), two[k % 2].cuda_stream)
            k2 = max(10, min(args.steps, 200))
            for k in range(4):
                launch_on(k)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for k in range(k2):
                launch_on(k)
            torch.cuda.synchronize(device)
            ms2 = (time.perf_counter() - t0) / k2 * 1e3
            same = bool((buf2[0].view(torch.uint8) == buf2[1].view(torch.uint8)).all())
            result["two_streams"] = {
                "value": round(n / ms2 / 1e3, 2), "unit": result["unit"], "ms_per_step": round(ms2, 4), "steps": k2,
                "results_equal": same,
                "note": "the same step alternating between two HIP streams (two result buffers): the start of one launch hides the drain of "
                        "the other.  Only for callers whose consecutive batches are independent; `value` above is the one-stream figure",
            }
            del buf2
        except Exception as exc:   # a secondary figure must never cost the headline line
            log(f"[bench] two-stream leg failed: {exc}")


def merged_launch(c) -> None:
    (HIT, RAY, any_hit, args, d_hits, d_rays, device, dist_on, image_width, log, ms_per_step, n, rank, result, scene,
    stream, torch, under_profiler, world) = (
        c.HIT, c.RAY, c.any_hit, c.args, c.d_hits, c.d_rays, c.device, c.dist_on, c.image_width, c.log, c.ms_per_step,
        c.n, c.rank, c.result, c.scene, c.stream, c.torch, c.under_profiler, c.world)
    # ---- many small batches (rank 0, N = 1): the same rays as 16 sets, 16 launches against ONE merged launch ----------------------
    # What a caller with many small ray sets per frame pays: a launch costs ~0.3 ms beyond its rays (grid start + drain), 16 sets pay
    # it 16 times unless they share a launch (vt_trace_closest_multi_dev: one cursor over all sets, one drain).  Never `value`.
    if rank == 0 and world == 1 and not dist_on and n >= 16 * 4096 and not under_profiler() and args.legs != "off":
        try:
            sets = 16
            per = (n // sets) // (16 * image_width) * (16 * image_width) if image_width else (n // sets) // 64 * 64
            if per > 0:
                esz = 1 if any_hit else HIT.itemsize
                out_a, out_b = torch.zeros_like(d_hits), torch.zeros_like(d_hits)
                descs = [(d_rays.data_ptr() + k * per * RAY.itemsize, out_b.data_ptr() + k * per * esz, per, image_width) for k in range(sets)]

                def separate():
                    for k in range(sets):
                        if any_hit:
                            scene.trace_any_dev(d_rays.data_ptr() + k * per * RAY.itemsize, per, out_a.data_ptr() + k * per * esz, stream)
                        else:
                            scene.trace_closest_dev(d_rays.data_ptr() + k * per * RAY.itemsize, per, out_a.data_ptr() + k * per * esz, stream)

                def merged():
                    scene.trace_multi_dev(descs, stream, any_hit=any_hit)

                times = {}
                for name, fn in (("separate", separate), ("merged", merged)):
                    for _ in range(3):
                        fn()
                    torch.cuda.synchronize(device)
                    reps = max(3, min(50, int(0.25 / max(ms_per_step * 1e-3, 1e-5))))
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    torch.cuda.synchronize(device)
                    times[name] = (time.perf_counter() - t0) / reps * 1e3
                same = bool(torch.equal(out_a[: sets * per * esz], out_b[: sets * per * esz]))
                result["merged_launch"] = {
                    "sets": sets, "rays_per_set": per,
                    "separate_launches_ms": round(times["separate"], 4), "separate_launches_value": round(sets * per / times["separate"] / 1e3, 2),
                    "one_merged_launch_ms": round(times["merged"], 4), "one_merged_launch_value": round(sets * per / times["merged"] / 1e3, 2),
                    "unit": result["unit"], "results_equal": same,
                    "note": "the workload's rays cut into 16 equal sets (whole bands of 16 image rows for camera rays): 16 launches on one stream "
                            "against vt_trace_*_multi_dev (one grid start, one drain); the results must be byte-equal",
                }
                del out_a, out_b
        except Exception as exc:   # a secondary figure must never cost the headline line
            log(f"[bench] merged-launch leg failed: {exc}")


def host_inclusive(c) -> None:
    (HIT, RAY, any_hit, apply_image_hint, args, d_hits, d_rays, dist_on, engine, log, n, out_bytes, rank, rays_host,
    result, scene, tp, under_profiler, va, world) = (
        c.HIT, c.RAY, c.any_hit, c.apply_image_hint, c.args, c.d_hits, c.d_rays, c.dist_on, c.engine, c.log, c.n,
        c.out_bytes, c.rank, c.rays_host, c.result, c.scene, c.tp, c.under_profiler, c.va, c.world)
    # ---- the transfer-inclusive figure of SURVEY 8(d) (rank 0, N = 1): never `value` ------------------------------------------
    # vt_trace_closest on HOST buffers: the same rays from pageable caller memory, every copy inside the call (the engine's chunked
    # pinned pipeline: upload of chunk c + 1, trace of chunk c and download of chunk c - 1 overlap)
    if rank == 0 and world == 1 and not dist_on and n > 0 and not under_profiler() and args.legs != "off":
        try:
            h_rays = rays_host if rays_host is not None else tp.to_host(d_rays, RAY)
            rays_host = c.rays_host = h_rays          # the CPU baseline leg of bench.py reuses the host copy
            nh = len(h_rays)
            h_out = np.empty(nh, dtype=np.uint8) if any_hit else np.empty(nh, dtype=HIT)
            fn = (lambda: va._lib.check(va._lib.lib.vt_trace_any(scene._h, va._lib.ptr(h_rays), nh, va._lib.ptr(h_out)))) if any_hit else \
                 (lambda: scene.trace_closest(h_rays, h_out))
            engine.set_option("ray_image_width", 0)
            fn()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            apply_image_hint(args, engine)
            ref_dev = d_hits[:nh].cpu().numpy() if any_hit else tp.to_host(d_hits[: nh * HIT.itemsize], HIT)
            same_p = bool((h_out.view(np.uint8) == ref_dev.view(np.uint8)).all())
            # the same call on arrays the caller has page-locked once (vt_host_register): no staging copies, copy engines only
            locked, ts_l, same_l = None, [], None
            try:
                va.host_register(h_rays); va.host_register(h_out)
                locked = True
                h_out[...] = 0 if any_hit else np.zeros(1, HIT)[0]
                fn()
                for _ in range(3):
                    t0 = time.perf_counter()
                    fn()
                    ts_l.append(time.perf_counter() - t0)
                same_l = bool((h_out.view(np.uint8) == ref_dev.view(np.uint8)).all())
            except Exception as exc:
                log(f"[bench] host-inclusive leg, page-locked arrays: {exc}")
            finally:
                if locked:
                    va.host_unregister(h_rays); va.host_unregister(h_out)
            result["host_inclusive"] = {
                "value": round(nh / min(ts) / 1e6, 2), "unit": result["unit"], "ms_per_call": round(min(ts) * 1e3, 3), "rays": nh,
                "bytes_over_pcie_per_ray": 32 + out_bytes,
                "results_equal_device_resident": same_p,
                "page_locked_arrays": {"value": round(nh / min(ts_l) / 1e6, 2), "ms_per_call": round(min(ts_l) * 1e3, 3), "results_equal_device_resident": same_l,
                                       "note": "the caller's arrays page-locked once with vt_host_register (not timed: ~70 us per MB): the copy engines "
                                               "read and write them in place, no staging copies"} if ts_l else None,
                "note": "vt_trace_closest / vt_trace_any on pageable host arrays, best of 3 calls: staging copies, H2D, trace, D2H and copy-out "
                        "all inside the call.  The transfer-inclusive second figure of SURVEY 8(d); `value` above has rays and hits resident in HBM",
            }
            del h_out
        except Exception as exc:   # a secondary figure must never cost the headline line
            log(f"[bench] host-inclusive leg failed: {exc}")


def beyond_cache(c) -> None:
    (HBM_PEAK_GBS, HIT, W, any_hit, args, build_scene, collect_pmc_live, committed_pmc, dev_index, device, dist_on,
    log, make_rays, n, rank, result, sha, torch, tp, under_profiler, va, world) = (
        c.HBM_PEAK_GBS, c.HIT, c.W, c.any_hit, c.args, c.build_scene, c.collect_pmc_live, c.committed_pmc,
        c.dev_index, c.device, c.dist_on, c.log, c.make_rays, c.n, c.rank, c.result, c.sha, c.torch, c.tp,
        c.under_profiler, c.va, c.world)
    # ---- beyond every cache (rank 0, N = 1): the same ray kind into S10M, where HBM CAN bind ------------------------------------
    # The headline scene (104 MB of records) lives in L2 / Infinity Cache, so its HBM fraction says little about the kernel.  S10M is
    # 1.04 GB of records -- beyond the 256 MiB Infinity Cache: this leg reports its rate, its algorithmic bytes and, from the FETCH /
    # WRITE counters (live with --beyond-cache-pmc, else the committed pass of exactly these kernel sources), its HBM fraction.
    if rank == 0 and world == 1 and not dist_on and n > 0 and not under_profiler() and args.legs == "all" and args.scene != "S10M" \
            and not any_hit and args.alpha_frac == 0 and args.scaling == "weak":
        try:
            b_args = argparse.Namespace(**vars(args))
            b_args.scene, b_args.kind = "S10M", "bounce"
            _, _, _, b_engine, b_scene, _ = build_scene(b_args, va, W, dev_index, world)
            b_rays, b_n, _, b_workload, _ = make_rays(b_args, rank, world, va, W, tp, b_engine, b_scene, device)
            b_engine.set_option("ray_image_width", 0)
            b_hits = tp.empty_records(b_n, HIT, device)
            _, b_stats = tp.trace_stats(b_scene, b_rays, b_n)
            b_st = b_stats.view(torch.int32).view(b_n, 2).sum(dim=0, dtype=torch.int64).cpu().numpy()
            del b_stats
            for _ in range(3):
                tp.trace_closest(b_scene, b_rays, b_n, b_hits)
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b_steps = 20
            e0.record()
            for _ in range(b_steps):
                tp.trace_closest(b_scene, b_rays, b_n, b_hits)
            e1.record()
            torch.cuda.synchronize(device)
            b_ms = e0.elapsed_time(e1) / b_steps
            b_alg = float(b_n) * 48.0 + 64.0 * float(b_st[0] + b_st[1])
            b_pmc, b_src = {}, None
            if args.beyond_cache_pmc:
                b_pmc = collect_pmc_live(b_args, ["FETCH_SIZE", "WRITE_SIZE"])
                b_src = "live rocprofv3 --pmc passes of this launch (child processes of this run)" if "FETCH_SIZE" in b_pmc else None
            if "FETCH_SIZE" not in b_pmc:
                b_pmc = committed_pmc(b_workload, args.builder, sha)
                b_src = b_pmc.pop("_source") + " (kernel sources unchanged since that pass)" if "FETCH_SIZE" in b_pmc else None
            b_traffic = int(b_pmc["FETCH_SIZE"] * 1024 + b_pmc.get("WRITE_SIZE", 0.0) * 1024) if "FETCH_SIZE" in b_pmc else None
            result["beyond_cache"] = {
                "workload": b_workload, "scene_record_bytes": int(b_scene.device_bytes),
                "value": round(b_n / (b_ms * 1e-3) / 1e6, 2), "unit": "Mrays/s", "kernel_ms": round(b_ms, 4), "steps": b_steps,
                "steps_per_ray": round(float(b_st[0]) / b_n, 2), "tests_per_ray": round(float(b_st[1]) / b_n, 2),
                "alg_achieved_gb_s": round(b_alg / (b_ms * 1e-3) / 1e9, 1), "alg_over_peak": round(b_alg / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": b_traffic, "traffic_source": b_src,
                "achieved_gb_s": round(b_traffic / (b_ms * 1e-3) / 1e9, 1) if b_traffic else None,
                "frac": round(b_traffic / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if b_traffic else None,
                "note": "16 Mi bounce rays into 10 M triangles (1.04 GB of records: beyond L2 and the Infinity Cache); frac = (FETCH_SIZE + "
                        "WRITE_SIZE) x 1024 B / kernel time / 8 TB/s, as roofline.frac",
            }
            del b_rays, b_hits, b_scene, b_engine
        except Exception as exc:   # a secondary figure must never cost the headline line
            log(f"[bench] beyond-cache leg failed: {exc}")


def rebuild(c) -> None:
    (W, args, dist_on, engine, host_threads, log, rank, rebuild_leg, result, under_profiler, va, world) = (
        c.W, c.args, c.dist_on, c.engine, c.host_threads, c.log, c.rank, c.rebuild_leg, c.result, c.under_profiler,
        c.va, c.world)
    # ---- what a Rebuild costs (rank 0, N = 1): never `value` -------------------------------------------------------------------
    if rank == 0 and world == 1 and not dist_on and not under_profiler() and (args.rebuild_leg == "on" or (args.rebuild_leg == "auto" and args.legs == "all")):
        try:
            result["rebuild"] = rebuild_leg(args, va, W, engine, host_threads)
        except Exception as exc:   # a secondary figure must never cost the headline line
            log(f"[bench] rebuild leg failed: {exc}")


def flat_scalars(c) -> None:
    (rank, result) = (
        c.rank, c.result)
    # the legs' figures flat in `roofline` as well (scalars only: what a parser that drops nested objects still keeps)
    if rank == 0:
        hi_, bc_, ts_ = result.get("host_inclusive") or {}, result.get("beyond_cache") or {}, result.get("two_streams") or {}
        result["roofline"].update({
            "l1_gather_reference_loop_lo": 66, "l1_gather_reference_loop_hi": 73,
            "host_inclusive_mrays_s": hi_.get("value"), "host_inclusive_ms": hi_.get("ms_per_call"),
            "host_inclusive_page_locked_mrays_s": (hi_.get("page_locked_arrays") or {}).get("value"),
            "host_inclusive_page_locked_ms": (hi_.get("page_locked_arrays") or {}).get("ms_per_call"),
            "beyond_cache_workload": bc_.get("workload"), "beyond_cache_mrays_s": bc_.get("value"), "beyond_cache_kernel_ms": bc_.get("kernel_ms"),
            "beyond_cache_frac": bc_.get("frac"), "beyond_cache_alg_over_peak": bc_.get("alg_over_peak"),
            "two_streams_mrays_s": ts_.get("value"),
            "sets16_separate_launches_mrays_s": (result.get("merged_launch") or {}).get("separate_launches_value"),
            "sets16_one_merged_launch_mrays_s": (result.get("merged_launch") or {}).get("one_merged_launch_value"),
        })


def run(c) -> None:
    """Every leg in bench.py's former order; each checks its own conditions (rank 0, N = 1, --legs, not under a profiler)."""
    for leg in (alt_builder, two_streams, merged_launch, host_inclusive, beyond_cache, rebuild, flat_scalars):
        leg(c)
