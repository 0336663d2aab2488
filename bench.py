#!/usr/bin/env python3
"""bench.py -- Mrays/s closest-hit on the 1M-triangle scene (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Default workload (config.workload = "S1M_bounce16777216", BASELINE.json configs[2]): scene S1M (1 000 300
triangles; tree from the product's default builder, binned SAH -- `alt_builder` repeats the measurement on the
reference-algorithm PLOC tree), 16 777 216 incoherent cosine-hemisphere bounce rays generated from the 4096x4096
primary hits of one camera, closest hit.  One "step" = one pass of the hot path (vt_trace_closest_dev) over the whole ray batch, rays and
hits resident in HBM.  With N > 1 ranks the BVH is replicated and
  --scaling weak   (default) every rank traces its own 16 Mi-ray batch (same camera, the rank's own bounce seed);
  --scaling strong BASELINE configs[4] verbatim with `--scene S10M`: 128 tiles of 1024x1024 primary rays from 128
                   seeded camera poses = 134 217 728 rays in total, split contiguously over the ranks;
either way the hit records are gathered to rank 0 over RCCL (xGMI) inside the step -- the single exchange of the path.

The printed JSON line also carries
  roofline     -- the dominant kernel against the HBM roofline: `traffic` = fabric-side bytes per launch from
                  rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected live by re-running this workload in a child
                  process under the profiler, or read from profiles/r<N>/ when the kernel sources are unchanged),
                  `achieved` = traffic / mean launch duration (HIP events on the launch stream over the timed region),
                  `frac` = achieved / 8 TB/s.  The algorithmic figure of SURVEY.md 8(d) (32 + 16 + 64 * steps + 64 *
                  tests bytes per ray, counters from the device stats kernel) is `alg_achieved` / `alg_over_peak`: it
                  exceeds the peak because records are served by L1 / L2 / Infinity Cache.  `bound_actual` says what
                  the kernel is bound by instead (VALU issue), from an SQ counter pass of the same launch;
  cpu_baseline -- the CPU oracle (oracle/, a port: the reference itself cannot be built here) on the host's cores over
                  a bounded sample of the same rays: -O3 x86-64-v3 build, threads pinned, one tree replica per NUMA node.
Data is synthetic (seeded generator, vistrace_amd/workloads.py); nothing reads /root/reference.
"""
from __future__ import annotations

import argparse
from types import SimpleNamespace
import csv
import glob
import hashlib
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
if int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # the cpu_baseline leg: pin the oracle's OpenMP threads (must be set before libgomp initialises)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured streaming copy)
ROUND = "r6"
KERNEL_SOURCES = ("vistrace_amd/csrc/trace_kernels.hip", "vistrace_amd/csrc/trace_kernels.h", "vistrace_amd/csrc/engine.hip",
                  "vistrace_amd/csrc/engine_internal.h", "vistrace_amd/csrc/Makefile")
# issue cost per wave-instruction and SIMD in cycles, measured on this part (scripts/ubench_valu.hip, profiles/r1/notes.md)
ISSUE_COST = {"MUL_F32": 2.4, "ADD_F32": 2.4, "FMA_F32": 4.2, "TRANS_F32": 8.2, "INT32": 3.2, "OTHER": 4.2}
BUILDER_NAMES = {"sah": "binned SAH, 16 bins, task-parallel, subtrees refined by re-insertion (VT_BUILDER_BINNED_SAH: the default of vt_bvh_build)",
                 "ploc": "PLOC r=14 + SAH leaf collapse (VT_BUILDER_PLOC: the reference's algorithm; the tree of round 1)",
                 "sah_refined": "the default tree + 2 re-insertion passes over the whole tree (VT_BUILDER_BINNED_SAH_REFINED: opt-in, +25 % build time)"}
SIMDS = 1024               # 256 CUs x 4
CLOCK_GHZ = 2.4            # nominal; main() replaces it by the device's own shader clock (hipDeviceAttributeClockRate)
CLOCK_SOURCE = "nominal 2.4 GHz (MI355X_MICROARCH.md)"


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def kernel_sources_sha() -> str:
    h = hashlib.sha1()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ---- workload -----------------------------------------------------------------------------------------------------
def build_scene(args, va, W, dev_index, world):
    t0 = time.time()
    verts = W.make_scene(args.scene)
    # --alpha-frac F: that share of the triangles carries VT_TRI_ALPHATEST (Primitives.h:196-208): the ALPHA kernel variants
    rig = W.alpha_test_rig(len(verts), alpha_fraction=args.alpha_frac) if args.alpha_frac > 0 else None
    tris = va.tris_setup(verts, rig[0] if rig else None)
    t1 = time.time()
    host_threads = max(1, len(os.sched_getaffinity(0)) // max(1, world))   # explicit: launchers may export OMP_NUM_THREADS=1
    bvh = va.HostBvh(tris, nthreads=min(16, host_threads), builder=args.builder)   # ranks build side by side
    t2 = time.time()
    engine = va.Engine(dev_index)
    if args.mode is not None:
        engine.set_option("persistent", 1 if args.mode == "persistent" else 0)
    for kv in args.engine_opt:
        k, v = kv.split("=")
        engine.set_option(k, int(v))
    t2b = time.time()
    # Rebuild's upload step: the tree and the triangle records go up as they are, the device re-packs them (vt_scene_upload_tree)
    scene = va.Scene.from_tree(engine, bvh)
    t3 = time.time()
    if rig:
        scene.set_tri_attribs(rig[1].view(va.TRI_ATTRIBS))
        scene.set_alpha(rig[2].view(va.ALPHA_MATERIAL), rig[3])
    scene.alpha_rig = rig
    st = scene.upload_stats()
    log(f"[bench] scene {args.scene}: {len(tris)} tris, {(len(bvh.nodes()) - 1) // 2} pairs, {scene.device_bytes / 1e6:.1f} MB on device; "
        f"gen {t1 - t0:.2f}s build {t2 - t1:.2f}s engine {t2b - t2:.3f}s upload + device re-pack {(t3 - t2b) * 1e3:.1f} ms "
        f"(copies issued {st['copy_ms']:.1f}, device {st['device_ms']:.1f})")
    return tris, bvh, None, engine, scene, host_threads


def make_rays(args, rank, world, va, W, tp, engine, scene, device):
    """This rank's device-resident rays.  Returns (d_rays, n_local, n_total, workload name, host copy or None)."""
    import torch
    from vistrace_amd._lib import HIT_ATTRS, RAY
    stream0 = tp.current_stream_handle(device)
    if args.scaling == "strong":
        # BASELINE configs[4]: `tiles` camera poses x 1024^2 pixel-centre rays, split contiguously over the ranks
        tile = 1024 * 1024
        per = (args.tiles + world - 1) // world
        lo_t, hi_t = min(args.tiles, rank * per), min(args.tiles, (rank + 1) * per)
        n_local = (hi_t - lo_t) * tile
        d_rays = tp.empty_records(max(n_local, 1), RAY, device)
        for t in range(lo_t, hi_t):
            pos, fwd = W.camera_pose(args.scene, t)
            engine.gen_primary_dev(1024, 1024, d_rays.data_ptr() + (t - lo_t) * tile * RAY.itemsize, pos=tuple(float(x) for x in pos),
                                   forward=tuple(float(x) for x in fwd), stream=stream0)
        torch.cuda.synchronize(device)
        log(f"[bench] strong scaling: {args.tiles} tiles of 1024x1024 primary rays, rank {rank} traces tiles [{lo_t}, {hi_t})")
        return d_rays, n_local, args.tiles * tile, f"{args.scene}_primary_{args.tiles}x1048576_tiles", None
    side = args.side
    n = side * side
    # weak scaling: every rank gets a batch of the SAME difficulty -- the camera of the N = 1 workload, its own bounce seed
    # (different seeded cameras see 10-20 % more or fewer steps per ray, which would show up as scaling loss or gain)
    cam = W.camera_positions(args.scene)[0]
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
    elif args.kind == "shadow":
        # BASELINE configs[3]: `per_hit` shadow rays per primary hit towards 16 seeded point lights, tmax = dist (1 - 1e-4)
        # (generated on the host: there is no device generator for them; 64 Mi rays take ~30 s of numpy)
        d_hits0 = tp.trace_closest(scene, d_prim, n)
        attrs = tp.to_host(tp.hit_attrs(scene, d_prim, d_hits0, n), HIT_ATTRS)
        del d_hits0, d_prim
        rays_host = W.shadow_rays(attrs, W.light_positions(args.scene), W.SEED + 4 + 1000 * rank, per_hit=args.shadow_per_hit)
        del attrs
        d_rays = tp.to_device(rays_host, device)
        n = len(rays_host)
        log(f"[bench] shadow rays: {n} = {args.shadow_per_hit} per primary hit of a {side}x{side} image, any-hit")
        if n > (1 << 24):
            rays_host = rays_host[: 1 << 24].copy()           # the CPU sample never needs more; keeps host memory bounded
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
    torch.cuda.synchronize(device)
    name = f"{args.scene}_{args.kind}{n}" + (f"_alpha{int(round(args.alpha_frac * 100))}" if args.alpha_frac > 0 else "")
    return d_rays, n, n * world, name, rays_host


# ---- PMC: fabric-side traffic and SQ counters of the dominant kernel ------------------------------------------------
PMC_PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq": ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
           "SQ_ACTIVE_INST_ANY", "SQ_WAVES"],
    "mix": ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32",
            "SQ_INSTS_VALU_INT32", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"],
    # the vector L1 (TCP) and the L2 (TCC): what the record gather costs (profiles/r3/notes.md)
    "tcp": ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
    "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "GRBM_GUI_ACTIVE"],
    # round 6: how busy the vector ALUs are, COUNTED (SQ_ACTIVE_INST_VALU: quad-cycles waves spend executing vector instructions;
    # rocprofiler-sdk's VALUBusy = that / CUs / GRBM_GUI_ACTIVE) -- with the cycle count of the same pass beside it
    "busy": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
             "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"],
}


def apply_image_hint(args, engine) -> int:
    """Camera-ray workloads are images in row-major order: tell the engine their row length (engine option "ray_image_width":
    a wave takes a 4 x 16 pixel tile instead of 64 neighbours of one row; scheduling only, arrays and results unchanged)."""
    width = 0
    if args.image_hint == "on" and (args.kind == "primary" or args.scaling == "strong"):
        width = 1024 if args.scaling == "strong" else args.side
    engine.set_option("ray_image_width", width)
    return width


def under_profiler() -> bool:
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")) or \
        "ROCPROFILER_LIBRARY_CTOR" in os.environ


def child_workload_args(args) -> list:
    a = ["--scene", args.scene, "--side", str(args.side), "--kind", args.kind, "--builder", args.builder, "--gen", "device",
         "--scaling", args.scaling, "--tiles", str(args.tiles), "--alpha-frac", str(args.alpha_frac),
         "--shadow-per-hit", str(args.shadow_per_hit)]
    if args.mode is not None:
        a += ["--mode", args.mode]
    for kv in args.engine_opt:
        a += ["--engine-opt", kv]
    return a


def kernel_row_prefix(args) -> str:
    """How the dominant kernel's rows start in rocprofv3's CSVs (spaces removed): trace_kernel<ANY_HIT,STATS,..."""
    return "%s<%s,false," % ("trace_kernel_alpha" if args.alpha_frac > 0 else "trace_kernel", "true" if args.kind == "shadow" else "false")


def collect_pmc_live(args, passes) -> dict:
    """Re-run this workload's dominant launch in a child process under `rocprofv3 --pmc` (one pass per counter group;
    never combined with other trace domains) and return {counter: value of the last dispatch of the trace kernel}."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        log("[bench] rocprofv3 not found: no live PMC pass")
        return {}
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for name in passes:
        tmp = tempfile.mkdtemp(prefix=f"vt_pmc_{name}_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", *PMC_PASSES[name], "--output-format", "csv", "-d", tmp, "--",
               "python3", os.path.join(ROOT, "bench.py"), "--pmc-child", *child_workload_args(args)]
        t0 = time.time()
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=args.pmc_timeout)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                log(f"[bench] PMC pass {name}: timed out after {args.pmc_timeout}s")
                shutil.rmtree(tmp, ignore_errors=True)
                break
            got = {}
            for f in glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    kn = r.get("Kernel_Name", "")
                    if kernel_row_prefix(args) in kn.replace(" ", ""):
                        got[r["Counter_Name"]] = float(r["Counter_Value"])     # rows are in dispatch order: keep the last
            log(f"[bench] PMC pass {name}: rc {rc}, {len(got)} counters, {time.time() - t0:.1f}s")
            out.update(got)
        except OSError as exc:
            log(f"[bench] PMC pass {name} failed: {exc}")
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def committed_pmc(workload: str, builder: str, sha: str) -> dict:
    """profiles/r<N>/pmc_<workload>_<builder>.json, only if it was taken from exactly these kernel sources."""
    path = os.path.join(ROOT, "profiles", ROUND, f"pmc_{workload}_{builder}.json")
    try:
        with open(path) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return {}
    if tj.get("workload") != workload or tj.get("builder") != builder or tj.get("kernel_sources_sha") != sha:
        log(f"[bench] {path} is stale (kernel sources changed): not used")
        return {}
    return dict(tj.get("counters", {}), _source=os.path.relpath(path, ROOT))


def kernel_cycles(pmc: dict, kernel_ms: float):
    """(shader cycles of the launch, clock in GHz, where the clock comes from): ONE figure for everything that is a share of the
    launch's cycles.  GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back); taken in a profiled pass,
    like every counter it is compared with."""
    if pmc.get("GRBM_GUI_ACTIVE"):
        cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0
        return cyc, cyc / (kernel_ms * 1e-3) / 1e9, "effective: GRBM_GUI_ACTIVE / 8 XCDs of the profiled launch (its duration taken as the un-profiled kernel_ms)"
    return kernel_ms * 1e-3 * CLOCK_GHZ * 1e9, CLOCK_GHZ, CLOCK_SOURCE


def bound_actual(pmc: dict, kernel_ms: float, cus: int = 0) -> dict | None:
    if "SQ_INSTS_VALU" not in pmc:
        return None
    valu = pmc["SQ_INSTS_VALU"]
    cyc, ghz, clock_src = kernel_cycles(pmc, kernel_ms)
    out = {"kind": "valu_issue", "sq_insts_valu": valu, "sq_insts_salu": pmc.get("SQ_INSTS_SALU"),
           "lane_utilisation": round(pmc["SQ_THREAD_CYCLES_VALU"] / (valu * 64.0), 4) if pmc.get("SQ_THREAD_CYCLES_VALU") else None,
           "kernel_cycles": round(cyc), "clock_ghz": round(ghz, 3), "clock_source": clock_src}
    if "SQ_WAVE_CYCLES" in pmc and pmc["SQ_WAVE_CYCLES"]:
        wc = pmc["SQ_WAVE_CYCLES"]
        out["wave_time_split"] = {k: round(pmc.get(c, 0.0) / wc, 3) for k, c in
                                  (("executing", "SQ_ACTIVE_INST_ANY"), ("waiting_memory_or_barrier", "SQ_WAIT_ANY"), ("waiting_to_issue", "SQ_WAIT_INST_ANY"))}
        if cus:
            out["resident_waves_per_simd"] = round(wc / cus / cyc, 2)      # SQ_WAVE_CYCLES (quad-cycles of wave residency) / CUs / cycles
    if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("GRBM_GUI_ACTIVE") and cus:
        # COUNTED: quad-cycles the waves spent executing vector instructions, over CUs x cycles of the same pass (rocprofiler-sdk's
        # VALUBusy formula for gfx9).  On gfx950 this is NOT a busy fraction of the vector pipe: the counter charges one quad-cycle
        # per instruction whatever the instruction costs the pipe (valu_active_per_inst below: ~1.0), and a SIMD retires a wave64
        # mul / add in 2.4 cycles while the issuing wave holds it for 4 (MI355X_MICROARCH.md: "2 cyc (SIMD-32); one wave alone: 4") --
        # the figure is the mean number of waves per SIMD that have a vector instruction in flight, between 1 x and 2 x the pipe's
        # busy fraction.  What the counters alone say about the pipe: busy >= (instructions x 2 cycles) / SIMD cycles.
        out["valu_active_waves_per_simd"] = round(pmc["SQ_ACTIVE_INST_VALU"] / cus / cyc, 3)
        out["valu_active_how"] = "SQ_ACTIVE_INST_VALU / CUs / (GRBM_GUI_ACTIVE / 8), one rocprofv3 --pmc pass"
        out["valu_active_quad_cycles_per_inst"] = round(pmc["SQ_ACTIVE_INST_VALU"] / valu, 3)
        out["valu_busy_bounds_from_counters"] = [round(valu * 2.0 / SIMDS / cyc, 3), round(min(1.0, valu * 4.0 / SIMDS / cyc), 3)]
        out["inst_active_waves_per_simd"] = {k: round(pmc[f"SQ_ACTIVE_INST_{k}"] / cus / cyc, 3) for k in ("SCA", "LDS", "VMEM", "MISC") if pmc.get(f"SQ_ACTIVE_INST_{k}")}
        if pmc.get("SQ_BUSY_CU_CYCLES"):
            out["cu_busy_frac"] = round(pmc["SQ_BUSY_CU_CYCLES"] / cus / cyc, 3)
    if "SQ_INSTS_VALU_MUL_F32" in pmc:
        named = {k: pmc.get(f"SQ_INSTS_VALU_{k}", 0.0) for k in ("MUL_F32", "ADD_F32", "FMA_F32", "TRANS_F32", "INT32")}
        other = max(0.0, valu - sum(named.values()))
        cycles = sum(named[k] * ISSUE_COST[k] for k in named) + other * ISSUE_COST["OTHER"]
        per_simd = cycles / SIMDS
        out["valu_issue_cycles_per_simd"] = round(per_simd)
        out["valu_busy_model"] = round(per_simd / cyc, 3)
        out["valu_busy_model_how"] = "instruction-class counts x issue costs of scripts/ubench_valu.hip (cycles per wave-instruction and SIMD: mul/add 2.4, int 3.2, fma/select/minmax 4.2, rcp 8.2) / kernel_cycles"
        out["valu_class_share_of_issue_cycles"] = {k: round(named[k] * ISSUE_COST[k] / cycles, 3) for k in named} | {"OTHER(select/minmax/cmp/mov/dpp)": round(other * ISSUE_COST["OTHER"] / cycles, 3)}
        out["branches"] = pmc.get("SQ_INSTS_BRANCH")
    # the figure DESIGN.md section 5 quotes: the MODEL (counted instructions x measured issue costs) -- gfx950 exposes no counter of
    # the vector pipe's busy cycles (SQ_INST_CYCLES_VALU is gfx12's); the counters bracket it (valu_busy_bounds_from_counters)
    out["valu_busy_frac"] = out.get("valu_busy_model")
    out["valu_busy_frac_is"] = "model, bracketed by counters" if "valu_busy_bounds_from_counters" in out and "valu_busy_model" in out else ("model" if "valu_busy_model" in out else None)
    return out


def gather_path(pmc: dict, kernel_ms: float, cus: int) -> dict | None:
    """What the record gather costs on the way in: the vector L1 (TCP) of every CU takes one access per 64-B record
    (quad-cooperative DMA) and returns them in order, so its throughput is (entries in flight) / (latency of the L2 reads
    among them).  MI355X_MICROARCH.md measures 66-73 GB/s per CU for L2-resident random gathers and 29-34 GB/s per CU
    from the Infinity Cache; the figures below are this launch against that ceiling (profiles/r3/notes.md)."""
    if "TCP_TOTAL_CACHE_ACCESSES_sum" not in pmc or not cus:
        return None
    acc, req = pmc["TCP_TOTAL_CACHE_ACCESSES_sum"], pmc.get("TCP_TCC_READ_REQ_sum", 0.0)
    out = {"tcp_accesses": acc, "tcp_l2_read_requests": req, "l1_hit_rate": round(1.0 - req / acc, 4) if acc else None,
           "l1_gb_s_per_cu": round(acc * 64 / cus / (kernel_ms * 1e-3) / 1e9, 1),
           "l2_to_l1_gb_s_per_cu": round(req * 64 / cus / (kernel_ms * 1e-3) / 1e9, 1),
           "guide_reference_gather_loop_gb_s_per_cu": {"l2_resident_gather": [66, 73], "infinity_cache_gather": [29, 34],
                                                       "note": "MI355X_MICROARCH.md measures these for ITS gather loop and labels them lower bounds: a reference point, not a ceiling"}}
    if pmc.get("TCP_TCC_READ_REQ_LATENCY_sum") and req:
        out["avg_l2_read_latency_cycles"] = round(pmc["TCP_TCC_READ_REQ_LATENCY_sum"] / req, 1)
    if "GRBM_GUI_ACTIVE" in pmc:
        cyc, ghz, _ = kernel_cycles(pmc, kernel_ms)                           # the one cycle count (roofline.bound_actual.kernel_cycles)
        out["effective_clock_ghz"] = round(ghz, 3)
        out["tcp_accesses_per_cycle_per_cu"] = round(acc / cus / cyc, 3)
        if "TCP_PENDING_STALL_CYCLES_sum" in pmc:
            out["tcp_pending_stall_frac"] = round(pmc["TCP_PENDING_STALL_CYCLES_sum"] / cus / cyc, 3)
    if "TCC_HIT_sum" in pmc and pmc.get("TCC_REQ_sum"):
        out["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc.get("TCC_MISS_sum", 0.0)), 4)
    return out


# ---- N > 1: gather of the hit records to rank 0 ---------------------------------------------------------------------
class NativeGather:
    """vt_gather_hits_dev (ncclGather from libvistrace_hip.so on its own communication stream), double-buffered: batch b
    traces into hits[b % 2] and lands in recv[b % 2] on rank 0 while batch b + 1 is traced."""

    def __init__(self, engine, n, world, rank, device, dist):
        import torch
        import vistrace_amd as va
        # every rank takes the same decision: a failure on one rank must not leave the others waiting in a collective
        ids = [None]
        if rank == 0:
            try:
                ids[0] = va.comm_unique_id()
            except Exception as exc:
                ids[0] = f"error: {exc}"
        dist.broadcast_object_list(ids, src=0)
        if isinstance(ids[0], str):
            raise RuntimeError(f"rank 0 could not create the RCCL id ({ids[0]})")
        err = None
        try:
            engine.comm_init_rank(world, rank, ids[0])
        except Exception as exc:
            err = exc
        bad = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(bad)
        if int(bad.item()) != 0:
            raise RuntimeError(f"vt_engine_comm_init_rank failed on {int(bad.item())} rank(s)" + (f": {err}" if err else ""))
        self.engine, self.n, self.rank = engine, n, rank
        self.recv = [torch.empty(world * n * 16, dtype=torch.uint8, device=device) if rank == 0 else None for _ in range(2)]
        # the root traces straight into its slice of the result (ncclGather in place: sendbuff == recvbuff + rank * count)
        self.hits = [self.recv[b][: n * 16] if rank == 0 else torch.empty(n * 16, dtype=torch.uint8, device=device) for b in range(2)]
        self.batch = 0

    def submit(self, trace, stream, chunks=1):
        """One batch.  chunks = K > 1: the shard is traced in K pieces and piece c is gathered (vt_gather_hits_part_dev) while piece
        c + 1 is traced -- what a one-shot batch needs; K = 1: one trace, one ncclGather."""
        import vistrace_amd as va
        b = self.batch % 2
        self.engine.gather_wait(1, stream)                      # the gather that read hits[b] two batches ago is over
        recv = self.recv[b].data_ptr() if self.rank == 0 else 0
        if chunks <= 1:
            trace(self.hits[b])
            self.engine.gather_hits_dev(self.hits[b].data_ptr(), self.n, recv, 0, stream)
        else:
            # all pieces on the caller's stream.  (Alternating two launch streams would hide each piece's drain behind the next
            # piece's start, but not with reserved CUs, which N > 1 needs for RCCL's kernels: the second grid's blocks are
            # dispatched to the only free room -- the reserved CUs -- and leave at once; measured 40 ms instead of 5.)
            for c in range(chunks):
                lo, hi = va.gather_chunk_bounds(self.n, chunks, c)
                trace(self.hits[b], lo, hi)
                self.engine.gather_hits_part_dev(self.hits[b].data_ptr(), self.n, c, chunks, recv, 0, stream)
        self.batch += 1
        return b

    def drain(self):
        self.engine.gather_wait(0)


def verify_gather(dist, rank, world, n, local_hits, recv, device, backend):
    """Every rank's hit shard must have arrived on rank 0 unchanged: 64-bit word sums per shard, compared on rank 0."""
    import torch
    mine = local_hits.view(torch.int64).sum().reshape(1)
    sums = [torch.zeros(1, dtype=torch.int64, device=mine.device) for _ in range(world)]
    dist.all_gather(sums, mine)
    ok = True
    if rank == 0:
        got = recv.view(torch.int64).view(world, -1).sum(dim=1)
        ok = bool((got == torch.cat(sums)).all())
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
    dist.broadcast(flag, src=0)
    return bool(flag.item())


def rebuild_leg(args, va, W, engine, host_threads) -> dict:
    """What accel:Rebuild costs (ref source/VisTrace.cpp:798-818 -> AccelStruct.cpp:762-775), step by step on this workload's
    scene, each step the best of 3: triangle set-up, the CPU build, then the part that is NOT the build -- re-packing the tree for
    the device and uploading it -- both ways: on the host (vt_scene_linearise + vt_scene_upload, rounds 1-4) and on the device
    (vt_scene_upload_tree, round 5; byte-equal records), against what the same bytes cost on the link.  Refit / skin refit (the
    per-frame alternative to a Rebuild) for contrast."""
    def best(fn, reps=3):
        ts, out = [], None
        for _ in range(reps):
            t0 = time.perf_counter()
            out = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts), out
    verts = W.make_scene(args.scene)
    threads = min(16, host_threads)
    setup_ms, tris = best(lambda: va.tris_setup(verts))
    build_ms, bvh = best(lambda: va.HostBvh(tris, nthreads=threads, builder=args.builder), reps=2)
    lin_ms, hs = best(lambda: va.HostScene(bvh))
    up_ms, sc_host = best(lambda: va.Scene(engine, hs), reps=1)
    ts = []
    for _ in range(3):                                     # fresh scene each time, the previous one freed outside the timed part
        sc_host.free()
        t0 = time.perf_counter()
        sc_host = va.Scene(engine, hs)
        ts.append((time.perf_counter() - t0) * 1e3)
    up_ms = min(ts)
    sc_dev, ts, stats = None, [], None
    for _ in range(4):
        if sc_dev is not None:
            sc_dev.free()
        t0 = time.perf_counter()
        sc_dev = va.Scene.from_tree(engine, bvh)
        ts.append((time.perf_counter() - t0) * 1e3)
        stats = sc_dev.upload_stats()
    tree_ms = min(ts[1:])                                  # the first call allocates the engine's staging block
    hp, ht = sc_host.read_records()
    t0 = time.perf_counter()
    sc_dev.download_host_scene()
    down_ms = (time.perf_counter() - t0) * 1e3
    dp, dt = sc_dev.read_records()
    equal = bool(hp.tobytes() == dp.tobytes() and ht.tobytes() == dt.tobytes())
    del hp, ht, dp, dt
    moved = (verts + np.float32(0.25)).astype(np.float32)
    sc_dev.refit(moved)
    refit_ms, _ = best(lambda: sc_dev.refit(moved))
    skin, base, nmat = W.skinned_rig(len(verts), nents=64, bones_per_ent=32)
    sc_dev.set_skin(verts, skin, base)
    bones, binds = W.rig_pose(nmat, 0)
    for _ in range(3):
        sc_dev.skin_refit(bones, binds)
    skin_ms, _ = best(lambda: sc_dev.skin_refit(bones, binds), reps=10)
    h2d = stats["bytes_h2d"]
    link_ms = h2d / 56e9 * 1e3                             # profiles/r4/upload_probe.txt: 56 GB/s host -> device on these boxes
    out = {
        "scene": args.scene, "triangles": int(len(tris)), "host_threads": threads,
        "tris_setup_ms": round(setup_ms, 2), "bvh_build_ms": round(build_ms, 1), "bvh_builder": args.builder,
        "host_repack": {"vt_scene_linearise_ms": round(lin_ms, 2), "vt_scene_upload_ms": round(up_ms, 2), "sum_ms": round(lin_ms + up_ms, 2)},
        "device_repack": {"vt_scene_upload_tree_ms": round(tree_ms, 2), "first_call_ms": round(ts[0], 2),
                          "copies_issued_ms": round(stats["copy_ms"], 2), "device_and_readback_ms": round(stats["device_ms"], 2),
                          "bytes_h2d": int(h2d), "link_time_ms_at_56GBs": round(link_ms, 2), "over_link_time": round(tree_ms / link_ms, 2)},
        "records_byte_equal": equal,
        "vt_host_scene_download_ms": round(down_ms, 2),
        "vt_scene_refit_ms": round(refit_ms, 2), "vt_scene_skin_refit_ms": round(skin_ms, 3),
        "note": "Rebuild = tris_setup + bvh_build (CPU, as north_star prescribes) + the re-pack / upload step; rounds 1-4 re-packed on the host "
                "(host_repack), round 5 on the device (device_repack: three plain copies + five kernels + one radix sort; the product path). "
                "vt_host_scene_download is paid only by callers that walk single rays on the host, on first use.",
    }
    sc_host.free(); sc_dev.free()
    return out


def launch_ranks(args) -> int:
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, exactly as the driver's
    N > 1 command does (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>`), as a CHILD process -- this process has not touched HIP (torch is not even imported yet) and
    never will: it relays rank 0's JSON line to stdout (everything else the ranks print on stdout goes to stderr, so the line
    is the only thing on stdout) and returns the child's exit code.  No fallback: a child that fails, or ends without a line,
    is a non-zero exit with a one-line reason on stderr."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.pop("OMP_PROC_BIND", None)          # set above for the N = 1 cpu_baseline leg only: N ranks must not all pin to the same cores
    env.pop("OMP_PLACES", None)
    log(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks as a child process ({' '.join(cmd[1:9])} bench.py ...)")
    try:
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    except OSError as exc:
        print(f"[bench] FATAL: could not start the ranks: {exc}", file=sys.stderr, flush=True)
        return 127

    def forward(signum, _frame):            # the driver's timeout reaches the ranks too
        try:
            os.killpg(child.pid, signum)
        except OSError:
            pass
    old = {sig: signal.signal(sig, forward) for sig in (signal.SIGTERM, signal.SIGINT)}
    lines = []
    try:
        for line in child.stdout:
            if line.startswith("{") and line.rstrip().endswith("}"):
                lines.append(line.rstrip("\n"))
            else:
                sys.stderr.write(line)
        rc = child.wait()
    finally:
        for sig, handler in old.items():
            signal.signal(sig, handler)
    if rc != 0:
        print(f"[bench] FATAL: the ranks exited with code {rc} (torch.distributed.run, {args.gpus} ranks); no result line", file=sys.stderr, flush=True)
        return rc if 0 < rc < 256 else 1
    if len(lines) != 1:
        print(f"[bench] FATAL: the ranks exited cleanly but printed {len(lines)} result lines instead of one", file=sys.stderr, flush=True)
        return 5
    print(lines[0], flush=True)
    return 0


def run_group(args) -> None:
    """--form group: ONE process drives all N devices -- vt_engine_open_multi, the scene replicated by vt_scene_upload, every member's
    shard of the rays resident on its device, one vt_trace_closest_gather_dev per step (traces on the members' streams, ONE gather of
    the hit records to the root over RCCL, double-buffered across steps).  This is the form a Lua state would use (one thread of one
    process: INTEGRATION.md 3b, ref call site source/objects/AccelStruct.cpp:818).  Same workloads, same line as the one-process-
    per-GPU form; `config.form` says which one ran, `config.simulated` whether members shared a device / RCCL was the test double."""
    import torch

    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, RAY, RAY_STATS

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the traversal has no CPU path)")
    if args.kind == "shadow":
        raise SystemExit("--kind shadow is a single-GPU workload (BASELINE configs[3])")
    ndev = args.gpus
    devices = [int(x) for x in args.group_devices.split(",")] if args.group_devices else list(range(ndev))
    if len(devices) != ndev:
        raise SystemExit(f"--group-devices names {len(devices)} devices, --gpus {ndev}")
    aliased = len(set(devices)) < ndev
    fake_rccl = "fake_rccl" in os.environ.get("VT_RCCL_LIB", "")
    simulated = None
    if aliased or fake_rccl:
        simulated = ("members share device(s) %s" % sorted(set(devices)) if aliased else "one device per member") + (
            "; RCCL replaced by the test double tests/cpp/fake_rccl.cpp (transfers = stream-ordered device copies)" if fake_rccl else "") + \
            ": the control flow and what ONE GPU needs for N members' work -- NOT a scaling number"
    if args.scaling == "strong" and args.tiles % ndev != 0:
        raise SystemExit("--form group --scaling strong needs --tiles divisible by --gpus")
    root_dev = torch.device("cuda", devices[0])
    torch.cuda.set_device(root_dev)

    # ---- scene: CPU build once, replicated to every member by vt_scene_upload ----------------------------------------------------
    t0 = time.time()
    verts = W.make_scene(args.scene)
    tris = va.tris_setup(verts)
    host_threads = max(1, len(os.sched_getaffinity(0)))
    bvh = va.HostBvh(tris, nthreads=min(16, host_threads), builder=args.builder)
    t1 = time.time()
    engine = va.Engine(devices)
    if args.mode is not None:
        engine.set_option("persistent", 1 if args.mode == "persistent" else 0)
    scene = va.Scene.from_tree(engine, bvh)              # every member re-packs its own copy (vt_scene_upload_tree)
    log(f"[bench] group of {ndev} members on devices {devices}: scene {args.scene} ({len(tris)} tris) built in {t1 - t0:.2f}s, "
        f"replicated in {time.time() - t1:.2f}s ({scene.device_bytes / 1e6:.1f} MB per member)")

    # ---- every member's rays on its own device (made with a single-device engine of that device; also the gather's reference) -----
    tile = 1024 * 1024
    singles = {}
    def single_for(d):
        if d not in singles:
            e1 = va.Engine(d)
            singles[d] = (e1, va.Scene.from_tree(e1, bvh))
        return singles[d]
    member_rays, member_n, ref_sums, tot_steps, tot_tests = [], [], [], 0, 0
    workload = None
    image_width = 0                 # camera-ray workloads are images in row-major order: the engines learn the row length
    if args.image_hint == "on" and (args.kind == "primary" or args.scaling == "strong"):
        image_width = 1024 if args.scaling == "strong" else args.side
    for g, d in enumerate(devices):
        dev = torch.device("cuda", d)
        with torch.cuda.device(dev):
            e1, s1 = single_for(d)
            r, n_g, n_total, workload, _ = make_rays(args, g, ndev, va, W, tp, e1, s1, dev)
            _, d_stats = tp.trace_stats(s1, r, n_g)                       # algorithmic bytes: exact counters, untimed
            st = d_stats.view(torch.int32).view(n_g, 2).sum(dim=0, dtype=torch.int64).cpu().numpy()
            tot_steps, tot_tests = tot_steps + int(st[0]), tot_tests + int(st[1])
            del d_stats
            e1.set_option("ray_image_width", image_width)
            ref = tp.trace_closest(s1, r, n_g)                                # the member's shard traced alone: what must arrive on the root
            torch.cuda.synchronize(dev)
            ref_sums.append(int(ref.view(torch.int64).sum().item()))
            del ref
        member_rays.append(r)
        member_n.append(n_g)
    for e1, s1 in singles.values():
        s1.free(); e1.close()
    singles.clear()
    if len(set(member_n)) != 1:
        raise SystemExit(f"members got shards of different sizes {member_n}: the group form needs equal shards")
    n = member_n[0]
    cap = va.shard_capacity(n_total, ndev)
    assert cap == n and n_total == n * ndev, (cap, n, n_total)
    engine.set_option("ray_image_width", image_width)
    alg_bytes = n_total * 48 + 64 * (tot_steps + tot_tests)
    log(f"[bench] workload {workload}: {n} rays per member, {n_total} in the job; steps/ray {tot_steps / n_total:.2f} tests/ray {tot_tests / n_total:.2f}")

    engine.set_timing(True)
    # room for RCCL's kernels beside the resident trace grids -- but not when members SHARE a device: several persistent grids on
    # one GPU with CUs reserved send each other's blocks to the reserved CUs (profiles/r4/notes.md section 3: 40 ms instead of 5),
    # and the test double's transfers are copies that need no CUs
    if args.reserve_cus > 0 and ndev > 1 and not aliased:
        engine.set_option("reserved_cus", args.reserve_cus)
    if args.overlap == "off":
        engine.set_option("gather_overlap", 0)
    engine.set_option("gather_chunks", max(1, args.chunks))
    ptrs = [r.data_ptr() for r in member_rays]
    outs = [torch.empty(ndev * cap * 16, dtype=torch.uint8, device=root_dev) for _ in range(2)]
    for o in outs:
        o.fill_(0xEE)

    def sync_all():
        for d in set(devices):
            torch.cuda.synchronize(torch.device("cuda", d))
        engine.synchronize()

    sync_all()
    batch = [0]

    def step():
        scene.trace_closest_gather_dev(ptrs, n_total, outs[batch[0] % 2].data_ptr())
        batch[0] += 1

    import threading

    def fire():
        print(f"[bench] FATAL: the group's warm-up (traces + gather) did not finish within {args.gather_timeout:.0f} s; no result line", file=sys.stderr, flush=True)
        os._exit(6)
    timer = threading.Timer(args.gather_timeout, fire) if args.gather_timeout > 0 else None
    if timer:
        timer.daemon = True
        timer.start()
    for _ in range(max(args.warmup, 2)):
        step()
    engine.synchronize()
    if timer:
        timer.cancel()
    # the proof that the gather delivered every member's records to the root unchanged (64-bit word sums per shard)
    last = outs[(batch[0] - 1) % 2].view(torch.int64).view(ndev, -1).sum(dim=1).cpu().numpy()
    gather_verified = bool(all(int(last[g]) == ref_sums[g] for g in range(ndev)))
    if not gather_verified:
        print("[bench] FATAL: the group's gather did NOT deliver every member's records to the root intact", file=sys.stderr, flush=True)
        sys.exit(4)

    sync_all()
    start = time.perf_counter()
    for _ in range(args.steps):
        step()
    engine.synchronize()                                   # every hit record of every step is on the root
    sync_all()
    elapsed = time.perf_counter() - start
    ms_per_step = elapsed / args.steps * 1e3

    # ---- where the step time goes (untimed, synchronised batches behind the timed region) ------------------------------------------
    tr, ga = [[] for _ in range(ndev)], []
    for _ in range(6):
        step()
        engine.synchronize()
        for g in range(ndev):
            tr[g].append(engine.member_kernel_ms(g))
        if ndev > 1:
            ga.append(engine.last_gather_ms())
    t_mean = [float(np.mean(x)) for x in tr]
    g_mean = float(np.mean(ga)) if ga else 0.0
    single = {}
    for K in (1, 2, 4, 8):
        engine.set_option("gather_chunks", K)
        ts = []
        for _ in range(4):
            sync_all()
            t0 = time.perf_counter()
            step()
            engine.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        single[str(K)] = round(float(np.mean(ts[1:])), 4)
    engine.set_option("gather_chunks", max(1, args.chunks))
    engine.set_timing(False)
    t_max = max(t_mean)
    hidden = max(0.0, t_max + g_mean - ms_per_step)
    value = n_total * args.steps / elapsed / 1e6
    k_ms = float(np.mean(t_mean))
    result = {
        "metric": ("Mrays/s closest-hit, 1M-triangle scene" if args.scene == "S1M" else "Mrays/s closest-hit, scene %s" % args.scene),
        "value": round(value, 2), "unit": "Mrays/s", "n_gpus": ndev, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded; rays generated on the device)",
        "config": {
            "workload": workload, "scene_triangles": int(len(tris)), "bvh_builder": BUILDER_NAMES[args.builder],
            "rays_per_gpu": n, "rays_total": n_total, "query": "closest-hit",
            "ray_kind": "pinhole primary" if (args.kind == "primary" or args.scaling == "strong") else "cosine-hemisphere bounce (incoherent)",
            "form": "group: ONE process, vt_engine_open_multi over %d members, vt_trace_closest_gather_dev per step" % ndev,
            "group_devices": devices,
            "simulated": simulated,
            "parallelism": f"rays sharded x{ndev}, BVH replicated, hits gathered to the root inside the step (ncclGather on the members' "
                           f"communication streams, double-buffered across steps); {engine.get_option('reserved_cus')} CUs keep room for its kernels",
            "gather_verified": gather_verified,
            "dist_breakdown": {
                "trace_ms_per_rank": {"min": round(min(t_mean), 4), "max": round(t_max, 4)},
                "gather_ms_per_rank": {"min": round(g_mean, 4), "max": round(g_mean, 4)} if ga else None,
                "gather_ms_how": "HIP events on the root's communication stream around one batch's gather (vt_engine_last_gather_ms); with members "
                                 "sharing a device the 'transfer' is device copies competing with the other members' traces",
                "step_ms": round(ms_per_step, 4), "chunks": max(1, args.chunks),
                "single_batch_ms": single,
                "single_batch_how": "one isolated batch (devices idle before, every record on the root after; wall clock) traced and gathered in K "
                                    "pieces (engine option gather_chunks), K = the keys",
                "overlap": args.overlap,
                "overlap_frac": round(min(1.0, hidden / min(t_max, g_mean)), 3) if ga and min(t_max, g_mean) > 0 else None,
                "overlap_note": "(trace + gather - step) / min(trace, gather): 1 = the shorter of the two is fully hidden, 0 = they run back to back",
                "gather_kind": "native ncclGather issued by vt_trace_closest_gather_dev (single-process group, ncclCommInitAll)",
                "bytes_into_root_per_step": int((ndev - 1) * cap * 16),
            },
            "launch_options": {k: engine.get_option(k) for k in ("lds_entries", "blocks_per_cu", "block_rays", "refill_threshold", "tri_threshold", "reserved_cus", "reserved_limit", "ray_image_width", "gather_chunks")},
        },
        "roofline": {
            "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "traffic_source": "PMC passes are taken at N = 1 only (the kernel is the same)",
            "alg_achieved": round(alg_bytes / ndev / (k_ms * 1e-3) / 1e9, 1),
            "alg_note": "algorithmic bytes (SURVEY 8(d)) of one member's shard / that member's mean launch duration",
            "kernel_ms": round(k_ms, 4), "kernel_ms_how": "mean over the members of vt_engine_last_kernel_ms (HIP events on each member's trace stream), 6 synchronised batches",
            "alg_bytes_per_ray": round(alg_bytes / n_total, 1), "steps_per_ray": round(tot_steps / n_total, 2), "tests_per_ray": round(tot_tests / n_total, 2),
            "kernel_sources_sha": kernel_sources_sha(),
        },
    }
    print(json.dumps(result), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="S1M")
    ap.add_argument("--side", type=int, default=4096, help="primary image side; rays per GPU = side*side (weak scaling)")
    ap.add_argument("--kind", default="bounce", choices=["bounce", "primary", "shadow"],
                    help="bounce = BASELINE configs[2] (the headline); primary with --scene S100k --side 1024 = configs[1]; "
                         "shadow = configs[3]: any-hit occlusion rays, --shadow-per-hit per primary hit (side 4096 x 4 = 64 Mi)")
    ap.add_argument("--shadow-per-hit", type=int, default=4)
    ap.add_argument("--alpha-frac", type=float, default=0.0,
                    help="share of the triangles that carry the alpha-test flag (Primitives.h:196-208): runs the ALPHA kernel variants")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank traces its own side*side batch; strong: BASELINE configs[4] -- `--tiles` camera tiles of "
                         "1024x1024 primary rays split over the ranks (use with --scene S10M)")
    ap.add_argument("--tiles", type=int, default=128, help="strong scaling: camera tiles (1024x1024 rays each) in the whole job")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the live rocprofv3 PMC passes (traffic then comes from profiles/ or is null)")
    ap.add_argument("--pmc-passes", default="fetch,write,sq,mix,tcp,l2,busy")
    ap.add_argument("--pmc-timeout", type=float, default=240.0)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--gather-timeout", type=float, default=180.0,
                    help="N > 1: seconds the warm-up (and, plus 0.25 s per step, the timed region) may take before the run is ended with a "
                         "reason and exit code 6 instead of hanging in a collective; 0 = no watchdog")
    ap.add_argument("--pieces-timeout", type=float, default=60.0,
                    help="N > 1: seconds the closing measurement of one batch in 2 / 4 / 8 pieces may take before it is abandoned")
    ap.add_argument("--legs", default=None, choices=["all", "host", "off"],
                    help="extra figures beside `value` at N = 1: host = host_inclusive (vt_trace_closest on host arrays: PCIe inside the call); "
                         "all = + beyond_cache (the same ray kind into S10M).  Default: all for the default workload, host otherwise")
    ap.add_argument("--rebuild-leg", default="auto", choices=["auto", "on", "off"],
                    help="N = 1: time accel:Rebuild's steps on this workload's scene and report them as `rebuild` (auto: beside the default workload)")
    ap.add_argument("--beyond-cache-pmc", action="store_true", help="collect the beyond_cache leg's FETCH / WRITE counters live (two more child passes)")
    ap.add_argument("--builder", default="sah", choices=["ploc", "sah", "sah_refined"],
                    help="sah = binned SAH, the product's default builder (vt_bvh_build); ploc = the reference's algorithm "
                         "(PLOC r=14 + leaf collapse); sah_refined = binned SAH + insertion-based optimisation (opt-in)")
    ap.add_argument("--alt-builder", default="ploc", choices=["ploc", "sah", "sah_refined", "none"],
                    help="N = 1: also time the same workload on the other builder's tree and report it as `alt_builder` "
                         "(default: the reference-algorithm PLOC tree, the one round 1 was measured on)")
    ap.add_argument("--gen", default="device", choices=["device", "host"], help="where the synthetic rays are generated")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one GPU per rank); gloo = test mode: ranks may share a GPU, hits gathered via host")
    ap.add_argument("--gather", default="native", choices=["native", "torch"],
                    help="N > 1: native = vt_gather_hits_dev (ncclGather issued by libvistrace_hip.so, verified in the warm-up; any "
                         "failure ends the run with a non-zero exit code -- a SCALE number is never silently a torch number); "
                         "torch = torch.distributed.gather, on request only")
    ap.add_argument("--overlap", default="on", choices=["on", "off"],
                    help="N > 1: off = every trace waits for the previous batch's gather (diagnostic: step = trace + gather)")
    ap.add_argument("--force-dist", action="store_true",
                    help="dev: run the N > 1 control flow (process group, gather pipeline, barriers) even with one rank")
    ap.add_argument("--reserve-cus", default="auto",
                    help="N > 1: CUs (one per shader engine of every XCD) on which the persistent trace grid leaves room "
                         "for the RCCL gather's kernels, so that the transfer of batch b overlaps the trace of batch b+1; "
                         "0 = off (the gather then only starts when the resident grid drains).  auto (default) = 32, and with more than one "
                         "rank a short probe behind the warm-up times 64 / 32 / 16 / 0 with the gather in flight and keeps the fastest "
                         "(what RCCL's kernels need beside the grid can only be measured on a multi-GPU node; results never depend on it); "
                         "probe = run that probe with one rank too (test)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="N > 1: pieces a rank's batch is traced and gathered in (piece c crosses the links while piece c + 1 is traced; "
                         "native: vt_gather_hits_part_dev, 1 .. 16).  1 = one trace + one ncclGather per step: in a stream of steps the gather "
                         "of step b already overlaps the trace of step b + 1; pieces pay for a one-shot batch (dist_breakdown.single_batch_ms)")
    ap.add_argument("--form", default="ranks", choices=["ranks", "group"],
                    help="N > 1: ranks = one process per GPU (vt_engine_comm_init_rank; what the driver launches, and what a bare "
                         "`bench.py --gpus N` starts by itself); group = ONE process drives all N devices (vt_engine_open_multi + "
                         "vt_trace_closest_gather_dev: the form a Lua state would use, INTEGRATION.md 3b)")
    ap.add_argument("--group-devices", default=None,
                    help="--form group: comma-separated device ids of the members (default 0 .. N-1).  A device listed twice needs "
                         "the test hooks (VT_ENABLE_TEST_HOOKS=1 VT_TEST_ALLOW_DEVICE_ALIASES=1) and the RCCL test double (VT_RCCL_LIB): "
                         "the line is then labelled simulated")
    ap.add_argument("--mode", default=None, choices=[None, "persistent", "static"])
    ap.add_argument("--engine-opt", action="append", default=[], metavar="KEY=VALUE",
                    help="dev: vt_engine_set_option on the workload's engine (e.g. alpha_threshold=4); recorded in config.launch_options")
    ap.add_argument("--image-hint", default="on", choices=["on", "off"],
                    help="camera-ray workloads (--kind primary, --scaling strong): pass the image's row length to the engine "
                         "(option ray_image_width); off = lanes take consecutive rays as for any other batch")
    args = ap.parse_args()
    args.reserve_probe = args.reserve_cus in ("auto", "probe")
    args.reserve_probe_always = args.reserve_cus == "probe"
    args.reserve_cus = 32 if args.reserve_probe else int(args.reserve_cus)
    if args.legs is None:   # the S10M leg only beside the default (headline) workload: the other configs are lines of their own
        args.legs = "all" if (args.scene == "S1M" and args.kind == "bounce" and args.side == 4096 and args.scaling == "weak" and args.alpha_frac == 0) else "host"

    if args.gpus > 1 and args.form == "ranks" and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        # before torch is imported and before anything touches HIP: the ranks are a child process, never an exec of this one
        sys.exit(launch_ranks(args))

    if args.form == "group":
        run_group(args)
        return

    import torch
    import torch.distributed as dist

    import vistrace_amd as va
    from vistrace_amd import torch_plumbing as tp
    from vistrace_amd.distributed import HitGatherPipeline
    from vistrace_amd import workloads as W
    from vistrace_amd._lib import HIT, RAY, RAY_STATS

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.form == "ranks" and args.gpus != world and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): launch with --nproc-per-node {args.gpus} "
                         f"(or run `python3 bench.py --gpus {args.gpus}` without a launcher: it starts its ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the traversal has no CPU path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(1, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    # ---- child of a PMC pass: the dominant launch a few times, nothing else ------------------------------------------
    if args.pmc_child:
        tris, bvh, host_scene, engine, scene, _ = build_scene(args, va, W, dev_index, 1)
        d_rays, n, _, _, _ = make_rays(args, 0, 1, va, W, tp, engine, scene, device)
        apply_image_hint(args, engine)
        if args.kind == "shadow":
            d_occ = torch.empty(n, dtype=torch.uint8, device=device)
            for _ in range(3):
                tp.trace_any(scene, d_rays, n, d_occ)
        else:
            d_hits = tp.empty_records(n, HIT, device)
            for _ in range(3):
                tp.trace_closest(scene, d_rays, n, d_hits)
        torch.cuda.synchronize(device)
        return

    global CLOCK_GHZ, CLOCK_SOURCE
    try:
        khz = int(torch.cuda.get_device_properties(device).clock_rate)       # hipDeviceAttributeClockRate, kHz
        if khz > 0:
            CLOCK_GHZ, CLOCK_SOURCE = khz / 1e6, "hipDeviceAttributeClockRate (the device's peak shader clock; it may hold less under load)"
    except Exception:
        pass

    any_hit = args.kind == "shadow"
    if any_hit and (world > 1 or args.force_dist or args.scaling == "strong"):
        raise SystemExit("--kind shadow is a single-GPU workload (BASELINE configs[3])")
    dist_on = world > 1 or args.force_dist
    result_fd = None
    if dist_on:
        # Everything a rank writes to stdout -- RCCL's version banner comes from C code -- goes to stderr from here on; the result
        # line alone is written to the real stdout at the end, so that stdout carries ONE line whoever launched the ranks.
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
    if dist_on:
        if args.force_dist and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    # ---- scene: CPU build once, upload once (Rebuild) -----------------------------------
    t3 = time.time()
    tris, bvh, host_scene, engine, scene, host_threads = build_scene(args, va, W, dev_index, world)
    d_rays, n, n_total, workload, rays_host = make_rays(args, rank, world, va, W, tp, engine, scene, device)
    d_hits = tp.empty_records(max(n, 1), HIT, device) if not any_hit else torch.empty(max(n, 1), dtype=torch.uint8, device=device)
    image_width = apply_image_hint(args, engine)       # after the rays exist: bounce / shadow rays are made from an un-hinted camera pass
    t4 = time.time()
    log(f"[bench] scene + ray set-up {t4 - t3:.2f}s; workload {workload}, {n} rays on this rank, {n_total} in the job")

    # ---- algorithmic bytes: exact counters from the stats kernel (untimed) --------------
    _, d_stats = tp.trace_any_stats(scene, d_rays, n) if any_hit else tp.trace_stats(scene, d_rays, n)
    torch.cuda.synchronize(device)
    stats = tp.to_host(d_stats, RAY_STATS)
    tot_steps = int(stats["steps"].sum(dtype=np.uint64))
    tot_tests = int(stats["tests"].sum(dtype=np.uint64))
    del d_stats
    out_bytes = 1 if any_hit else 16                     # 16-B hit record, 1 byte for any-hit
    alg_bytes = n * (32 + out_bytes) + 64 * (tot_steps + tot_tests)
    log(f"[bench] steps/ray {tot_steps / max(n, 1):.2f} tests/ray {tot_tests / max(n, 1):.2f} -> {alg_bytes / max(n, 1):.0f} B/ray algorithmic")

    # ---- timed region -------------------------------------------------------------------
    engine.set_timing(True)
    if dist_on and args.reserve_cus > 0 and args.backend == "gloo" and world > torch.cuda.device_count():
        log("[bench] gloo test mode with ranks sharing a GPU: no CUs reserved (several resident grids on one device)")
    elif dist_on and args.reserve_cus > 0:
        # the gather of batch b runs while batch b+1 is traced: its kernels need somewhere to run
        try:
            engine.set_option("reserved_cus", args.reserve_cus)
        except Exception as exc:   # never lose the run over an optimisation: trace without the reservation
            log(f"[bench] reserved_cus not available ({exc}); the gather will not overlap the resident trace grid")

    stream = tp.current_stream_handle(device)
    gather_kind = None
    pipe = native = None
    if dist_on:
        # weak: every rank sends n records; strong: shards are equal (tiles divide evenly) or padded to the largest
        n_send = n if args.scaling == "weak" else ((args.tiles + world - 1) // world) * 1024 * 1024
        if args.overlap == "off":
            engine.set_option("gather_overlap", 0)
        if args.gather == "native" and args.backend == "nccl":
            try:
                native = NativeGather(engine, n_send, world, rank, device, dist)
                gather_kind = "native ncclGather (vt_gather_hits_dev, own communication stream)"
            except Exception as exc:
                # no silent downgrade: a scaling number must say what moved the records (ask for --gather torch explicitly)
                print(f"[bench] FATAL rank {rank}: native gather unavailable ({exc}); re-run with --gather torch to measure "
                      f"torch.distributed.gather instead", file=sys.stderr, flush=True)
                sys.exit(3)
        if native is None:
            pipe = HitGatherPipeline(n_send, device, nchunks=args.chunks, via_host=args.backend == "gloo")
            gather_kind = "torch.distributed.gather" + (" via host (gloo test mode)" if args.backend == "gloo" else " (RCCL)")

    def trace_into(hits_buf, lo=0, hi=None):
        hi = n if hi is None else min(hi, n)
        if hi > lo:
            scene.trace_closest_dev(d_rays.data_ptr() + lo * RAY.itemsize, hi - lo, hits_buf.data_ptr() + lo * HIT.itemsize, stream)

    def step():
        if any_hit:
            tp.trace_any(scene, d_rays, n, d_hits)
        elif not dist_on:
            tp.trace_closest(scene, d_rays, n, d_hits)
        elif native is not None:
            native.submit(trace_into, stream, args.chunks)
        else:
            # chunked and double-buffered: the gather of one chunk/batch overlaps the tracing of the next
            pipe.submit(trace_into)

    def drain():
        if native is not None:
            native.drain()
        elif pipe is not None:
            pipe.drain()

    # N > 1: a collective that never completes (RCCL's kernels and the links have only ever run against a test double here) must
    # end the run with a reason, not with the launcher's timeout: a watchdog armed around the phases that contain a gather
    class Watchdog:
        def __init__(self):
            self.t = None
        def arm(self, seconds, what):
            self.disarm()
            if not dist_on or seconds <= 0:
                return
            import threading
            def fire():
                print(f"[bench] FATAL rank {rank}: {what} did not finish within {seconds:.0f} s (a gather that never completes?); "
                      f"no result line", file=sys.stderr, flush=True)
                os._exit(6)
            self.t = threading.Timer(seconds, fire)
            self.t.daemon = True
            self.t.start()
        def disarm(self):
            if self.t is not None:
                self.t.cancel()
                self.t = None
    watchdog = Watchdog()

    # warm-up; with N > 1 also the proof that the gather delivers every rank's records to rank 0 unchanged
    watchdog.arm(args.gather_timeout, "the warm-up (trace + gather)")
    for _ in range(max(args.warmup, 2 if dist_on else 0)):
        step()
    drain()
    torch.cuda.synchronize(device)
    watchdog.disarm()
    gather_verified = None
    if native is not None:
        b = (native.batch - 1) % 2
        try:
            gather_verified = verify_gather(dist, rank, world, n_send, native.hits[b], native.recv[b], device, args.backend)
        except Exception as exc:
            log(f"[bench] gather verification failed to run: {exc}")
            gather_verified = False
        if not gather_verified:
            print(f"[bench] FATAL rank {rank}: the native gather did NOT deliver every rank's records to rank 0 intact",
                  file=sys.stderr, flush=True)
            sys.exit(4)
    elif pipe is not None and args.backend == "nccl":
        b = (pipe.batch - 1) % 2
        recv = torch.cat(pipe.recv[b]) if rank == 0 else None
        gather_verified = verify_gather(dist, rank, world, n_send, pipe.hits[b], recv, device, args.backend)
        del recv

    # ---- N > 1: how many CUs to leave to the gather's kernels (auto): measured with the gather in flight, the fastest stays ------
    reserve_probe = None
    if args.reserve_probe and native is not None and n > 0 and (world > 1 or args.reserve_probe_always):
        # Every rank takes the same decisions at the same points: the candidate list comes from an all-reduced limit, and behind
        # every step that can fail locally (an option the engine refuses) the ranks exchange a status word, so that one rank's
        # exception ends the probe on ALL ranks together instead of leaving the others in the next collective.
        cdev = device if args.backend == "nccl" else "cpu"

        def all_ok(ok: bool) -> bool:
            t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item() > 0.5)

        watchdog.arm(args.gather_timeout, "the reserved-CU probe")
        lim = torch.tensor([float(engine.get_option("cu_count") // 2)], dtype=torch.float64, device=cdev)
        dist.all_reduce(lim, op=dist.ReduceOp.MIN)
        times, why = {}, None
        for cand in [c for c in (64, 32, 16, 0) if c <= int(lim.item())]:
            try:
                engine.set_option("reserved_cus", cand)
                ok = True
            except Exception as exc:
                ok, why = False, f"reserved_cus={cand}: {exc}"
            if not all_ok(ok):
                why = why or f"another rank could not set reserved_cus={cand}"
                break
            for _ in range(3):
                step()
            drain()
            torch.cuda.synchronize(device)
            dist.barrier()
            tq = time.perf_counter()
            for _ in range(20):
                step()
            drain()
            torch.cuda.synchronize(device)
            tt = torch.tensor([(time.perf_counter() - tq) / 20 * 1e3], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)               # every rank sees the same table and takes the same decision
            times[cand] = round(float(tt.item()), 4)
        watchdog.disarm()
        best = args.reserve_cus
        if why is None and times:
            best = min(times, key=lambda c: (times[c], -c))
            if 32 in times and times[32] <= times[best] * 1.01:         # within 1 %: stay with the calibrated default
                best = 32
            reserve_probe = {"ms_per_step": {str(k): v for k, v in times.items()}, "chosen": best,
                             "how": "20 steps each (trace + gather, double-buffered) behind 3 warm-up steps, max over ranks; scheduling only: results do not depend on it"}
            log(f"[bench] reserved-CU probe: {times} -> {best}")
        else:
            log(f"[bench] reserved-CU probe abandoned on every rank ({why or 'no candidate'}); staying with {args.reserve_cus}")
        try:                                                            # (the same value on every rank, whatever happened above)
            engine.set_option("reserved_cus", best)
        except Exception as exc:   # never lose the run over an optimisation
            log(f"[bench] reserved_cus={best} refused: {exc}")

    # single-launch durations (HIP events on the launch stream around one launch each)
    single_ms = []
    if n > 0:
        for _ in range(3):
            if any_hit:
                tp.trace_any(scene, d_rays, n, d_hits)
            else:
                tp.trace_closest(scene, d_rays, n, d_hits)
            single_ms.append(engine.last_kernel_ms())
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    watchdog.arm(args.gather_timeout + 0.25 * args.steps, "the timed region")      # generous: a step is milliseconds
    start = time.perf_counter()
    ev0.record()                                       # on the launch stream (torch's current stream)
    for _ in range(args.steps):
        step()
    ev1.record()
    drain()                                            # every hit record has reached rank 0
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - start
    watchdog.disarm()
    region_ms = ev0.elapsed_time(ev1)                  # launch-stream time of the K steps
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3

    # ---- N > 1: where the step time goes (untimed diagnostic: a few synchronised steps behind the timed region) -------------
    dist_breakdown = None
    if dist_on and n > 0:
        D = 6
        buf = native.hits[0] if native is not None else (pipe.hits[0] if pipe is not None else d_hits)
        tr, ga = [], []
        for _ in range(D):                                   # the trace alone (reserved CUs included), no gather in flight
            trace_into(buf)
            torch.cuda.synchronize(device)
            tr.append(engine.last_kernel_ms())
        if native is not None:
            for _ in range(D):                               # trace + gather, one batch at a time: the gather's own duration
                native.submit(trace_into, stream)
                native.drain()
                torch.cuda.synchronize(device)
                try:
                    ga.append(engine.last_gather_ms())
                except Exception:
                    pass
        # ONE batch from idle to "every record on rank 0", cut into K pieces: trace + gather for K = 1, about max(trace, gather) +
        # one piece when the pieces overlap (each extra piece costs one more launch's drain)
        single = {}

        def single_batch(K):
            ts = []
            for _ in range(4):
                torch.cuda.synchronize(device)
                dist.barrier()
                t0 = time.perf_counter()
                native.submit(trace_into, stream, K)
                native.drain()
                torch.cuda.synchronize(device)
                ts.append((time.perf_counter() - t0) * 1e3)
            tk = torch.tensor([float(np.mean(ts[1:]))], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(tk, op=dist.ReduceOp.MAX)
            single[str(K)] = round(float(tk.item()), 4)
        if native is not None:
            single_batch(1)          # the path the timed steps used; K > 1 runs at the very end, behind a watchdog (see below)
        t_mean, g_mean = float(np.mean(tr)), (float(np.mean(ga)) if ga else 0.0)
        dev_t = device if args.backend == "nccl" else "cpu"
        hi = torch.tensor([t_mean, g_mean], dtype=torch.float64, device=dev_t)
        lo = hi.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        t_max, g_max = float(hi[0]), float(hi[1])
        hidden = max(0.0, t_max + g_max - ms_per_step)
        dist_breakdown = {
            "trace_ms_per_rank": {"min": round(float(lo[0]), 4), "max": round(t_max, 4)},
            "gather_ms_per_rank": ({"min": round(float(lo[1]), 4), "max": round(g_max, 4)} if ga else None),
            "gather_ms_how": "HIP events on the communication stream around one ncclGather with no trace beside it (vt_engine_last_gather_ms)" if ga
                             else "not measured (torch.distributed.gather path)",
            "step_ms": round(ms_per_step, 4),
            "chunks": args.chunks,
            "single_batch_ms": single or None,
            "single_batch_how": "one isolated batch (device idle before, every record on rank 0 after; wall clock, max over ranks) traced and "
                                "gathered in K pieces, K = the keys",
            "overlap": args.overlap,
            "overlap_frac": round(min(1.0, hidden / min(t_max, g_max)), 3) if ga and min(t_max, g_max) > 0 else None,
            "overlap_note": "(trace + gather - step) / min(trace, gather): 1 = the shorter of the two is fully hidden, 0 = they run back to back",
            "gather_kind": gather_kind,
            "bytes_into_root_per_step": int((world - 1) * n_send * 16),
            "reserved_cus_probe": reserve_probe,
        }
    engine.set_timing(False)

    value = n_total * args.steps / elapsed / 1e6
    k_ms = region_ms / args.steps                      # mean launch duration over the timed region (launch to launch on the stream)
    alg_achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    # ---- fabric-side traffic + SQ counters of that launch (rank 0, N = 1) -------------------------------------------
    sha = kernel_sources_sha()
    pmc, pmc_source = {}, None
    if rank == 0 and world == 1 and not args.force_dist:
        if not args.no_pmc and not under_profiler():
            pmc = collect_pmc_live(args, [p for p in args.pmc_passes.split(",") if p in PMC_PASSES])
            if "FETCH_SIZE" in pmc:
                pmc_source = "live: rocprofv3 --pmc passes of this workload in a child process of this run (last dispatch of the kernel)"
                try:
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(ROOT, "gpurun_out", f"pmc_{workload}_{args.builder}.json"), "w") as f:
                        json.dump({"workload": workload, "builder": args.builder, "kernel_sources_sha": sha, "counters": pmc,
                                   "note": "last dispatch of vt::trace_kernel<false,false,...> in separate rocprofv3 --pmc passes "
                                           "(FETCH_SIZE, WRITE_SIZE in KB; factor 1.000 for 64-B record fetches: profiles/r1/calib_fetch.txt)"}, f, indent=1)
                except OSError:
                    pass
        if "FETCH_SIZE" not in pmc:
            com = committed_pmc(workload, args.builder, sha)
            if "FETCH_SIZE" in com:
                pmc_source = com.pop("_source") + " (kernel sources unchanged since that pass)"
                pmc = com
    traffic = None
    if "FETCH_SIZE" in pmc:
        # FETCH_SIZE / WRITE_SIZE are in KB; x1024 IS the byte count for this kernel's 64-B record fetches (calibrated,
        # profiles/r1/calib_fetch.txt) -- the x2 correction of MI355X_MICROARCH.md applies to wide streaming reads only
        traffic = int(pmc["FETCH_SIZE"] * 1024 + pmc.get("WRITE_SIZE", 0.0) * 1024)
    achieved = traffic / (k_ms * 1e-3) / 1e9 if traffic else None

    persistent = bool(engine.get_option("last_persistent"))
    dma = bool(engine.get_option("last_fetch_dma"))
    result = {
        "metric": ("Mrays/s %s, 1M-triangle scene" if args.scene == "S1M" else "Mrays/s %%s, scene %s" % args.scene) % ("any-hit" if any_hit else "closest-hit"),
        "value": round(value, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded; rays generated on the %s)" % args.gen,
        "config": {
            "workload": workload,
            "scene_triangles": int(len(tris)),
            "bvh_builder": BUILDER_NAMES[args.builder],
            "rays_per_gpu": n,
            "rays_total": n_total,
            "query": "any-hit (shadow occlusion, tMax early-out)" if any_hit else "closest-hit",
            "ray_kind": "pinhole primary" if (args.kind == "primary" or args.scaling == "strong") else
                        ("shadow rays from primary hits towards 16 point lights" if any_hit else "cosine-hemisphere bounce (incoherent)"),
            "alpha_tested_triangle_share": args.alpha_frac if args.alpha_frac > 0 else None,
            "parallelism": f"rays sharded x{world}, BVH replicated" + (
                f", hits gathered to rank 0 inside the step: {gather_kind}, double-buffered and overlapped with tracing; "
                f"{engine.get_option('reserved_cus')} CUs keep room for its kernels" if dist_on else ""),
            "gather_verified": gather_verified,
            "dist_breakdown": dist_breakdown,
            "kernel_mode": ("persistent" + ("+lds-dma-fetch" if dma else "")) if persistent else "static",
            "launch_options": {k: engine.get_option(k) for k in ("lds_entries", "blocks_per_cu", "block_rays", "refill_threshold", "tri_threshold", "alpha_threshold", "reserved_cus", "reserved_limit", "ray_image_width")},
            "launch": engine.launch_info(),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1) if achieved else None,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
            "traffic": traffic,
            "traffic_source": pmc_source,
            "traffic_note": "FETCH_SIZE + WRITE_SIZE (fabric side of L2, x1024 B; Infinity-Cache hits are counted, so true HBM traffic is lower still)",
            "alg_achieved": round(alg_achieved, 1),
            "alg_over_peak": round(alg_achieved / HBM_PEAK_GBS, 4),
            "alg_note": "algorithmic bytes (SURVEY 8(d)) / launch duration; > peak because records are served by L1/L2/Infinity Cache",
            "traffic_over_alg": round(traffic / alg_bytes, 4) if traffic else None,
            "compulsory_bytes": int(n * (32 + out_bytes) + scene.device_bytes),
            "kernel": ("vt::trace_kernel_alpha<%s,false,%s,%s>" if args.alpha_frac > 0 else "vt::trace_kernel<%s,false,%s,%s,false>") % (
                "true" if any_hit else "false", "true" if persistent else "false", "true" if dma else "false"),   # <ANY_HIT, STATS, PERSISTENT, FETCH_DMA[, ALPHA]>
            "kernel_ms": round(k_ms, 4),
            "kernel_ms_how": "HIP events on the launch stream around the K timed steps / K (cursor memset + kernel); single launches: %s ms" % (
                ", ".join(f"{x:.3f}" for x in single_ms)),
            "alg_bytes_per_ray": round(alg_bytes / max(n, 1), 1),
            "steps_per_ray": round(tot_steps / max(n, 1), 2),
            "tests_per_ray": round(tot_tests / max(n, 1), 2),
            "bound_actual": bound_actual(pmc, k_ms, engine.get_option("cu_count")),
            "gather_path": gather_path(pmc, k_ms, engine.get_option("cu_count")),
            "kernel_sources_sha": sha,
        },
    }
    # the limits that actually bind, flat beside the HBM figures the contract asks for (details in bound_actual / gather_path):
    # GB/s per CU through the vector L1 against the per-CU gather ceiling MI355X_MICROARCH.md measures, VALU issue, lane use
    ba, gp = result["roofline"]["bound_actual"] or {}, result["roofline"]["gather_path"] or {}
    result["roofline"].update({
        "l1_gather_gbs_per_cu": gp.get("l1_gb_s_per_cu"),
        "l1_gather_reference_loop": [66, 73],
        "l1_gather_frac": round(gp["l1_gb_s_per_cu"] / 73.0, 3) if gp.get("l1_gb_s_per_cu") else None,
        # the vector L1's own limit: one 64-B access per cycle and CU
        "l1_frac_of_64B_per_clk": round(gp["tcp_accesses_per_cycle_per_cu"], 3) if gp.get("tcp_accesses_per_cycle_per_cu") else (
            round(gp["l1_gb_s_per_cu"] / (64.0 * CLOCK_GHZ), 3) if gp.get("l1_gb_s_per_cu") else None),
        # L1 accesses that carry no algorithmic byte: idle lanes of the quad-cooperative fetch ask for record 0 to keep the four DMA
        # loads branch-free (always L1 hits).  algorithmic accesses = one per node step and triangle test + the ray / hit lines
        "dummy_fetch_share": round(1.0 - (tot_steps + tot_tests + n * (32 + out_bytes) / 64.0) / gp["tcp_accesses"], 4) if gp.get("tcp_accesses") else None,
        "l1_gather_note": "64-B record accesses of every CU's vector L1 per second x 64 B / CUs.  l1_gather_frac compares it with 73 GB/s per CU, the "
                          "guide's REFERENCE gather loop for L2-resident random gathers (which the guide labels a lower bound, measured for a different "
                          "loop): not a ceiling -- camera rays run this kernel's L1 at 115 GB/s per CU.  l1_frac_of_64B_per_clk is the same rate against "
                          "the L1's own limit of one 64-B access per cycle and CU (~154 GB/s at 2.4 GHz).  What binds the headline is a three-way "
                          "balance of VALU issue (valu_busy_frac), the L1 access rate and the dependent-fetch latency of a wave's chain; none is at its limit alone",
        "l1_hit_rate": gp.get("l1_hit_rate"),
        "valu_busy_frac": ba.get("valu_busy_frac"),
        "lane_utilisation": ba.get("lane_utilisation"),
    })

    # ---- secondary figures (never `value`): bench_legs.py -- the other builder's tree, two streams, merged launches, the
    # transfer-inclusive rate, the scene beyond every cache, what a Rebuild costs; selected by --legs / --alt-builder / --rebuild-leg
    import bench_legs
    ctx = SimpleNamespace(**{**globals(), **locals()})
    bench_legs.run(ctx)
    rays_host = ctx.rays_host

    # ---- CPU baseline + parity on a bounded sample (rank 0, N = 1 only) -------------------
    if rank == 0 and world == 1 and not args.no_cpu and n > 0:
        from oracle import binding as O
        if rays_host is None:
            rays_host = tp.to_host(d_rays, RAY)
        nodes = bvh.nodes().view(O.NODE)
        pidx = bvh.prim_indices()
        otris = O.tris_from_tri64(tris)
        n_host = len(rays_host)                               # shadow batches keep the first 16 Mi rays on the host
        pilot = min(n_host, 1 << 19)
        rig = getattr(scene, "alpha_rig", None)
        if rig:                                               # the alpha test lives in the baseline build of the oracle only
            O.set_alpha(otris, rig[1]["uv"].reshape(len(otris), 6), rig[1]["material"], rig[2].view(O.ALPHA_MATERIAL), rig[3])
        # the checker's own check: the -O3 x86-64-v3 build used for timing must equal the baseline build bit for bit
        if rig:
            class _PlainCtx:                                  # vto_traverse_batch names alpha triangles by address: no NUMA replicas
                fast, replicas = False, 1
                def traverse(self, rays, any_hit=False, want_stats=False, nthreads=0):
                    return O.traverse_batch(nodes, pidx, otris, rays, any_hit=any_hit, want_stats=want_stats, nthreads=nthreads)
                def close(self):
                    pass
            ctx = _PlainCtx()
        else:
            ctx = O.BatchContext(nodes, pidx, otris, nthreads=host_threads, fast=True)
        chk = min(n_host, 1 << 16)
        a = ctx.traverse(rays_host[:chk], any_hit=any_hit, want_stats=True)
        b = O.traverse_batch(nodes, pidx, otris, rays_host[:chk], any_hit=any_hit, want_stats=True)
        fast_equal = bool((a[0].view(np.uint8) == b[0].view(np.uint8)).all() and (a[1] == b[1]).all())
        if not fast_equal:
            log("[bench] the -O3 oracle build differs from the baseline build: timing the baseline build instead")
            ctx.close()
            if not rig:
                ctx = O.BatchContext(nodes, pidx, otris, nthreads=host_threads, fast=False)
        # the CPU gets its best thread count: all logical CPUs or one per physical core (SMT can hurt or help this walk)
        rate, cpu_threads = 0.0, host_threads
        for cand in sorted({host_threads, max(1, host_threads // 2)}, reverse=True):
            ctx.traverse(rays_host[:pilot], any_hit=any_hit, nthreads=cand)              # warm-up
            tp0 = time.perf_counter()
            ctx.traverse(rays_host[:pilot], any_hit=any_hit, nthreads=cand)
            r = pilot / (time.perf_counter() - tp0)
            if r > rate:
                rate, cpu_threads = r, cand
        sample = int(min(n_host, max(pilot, rate * args.cpu_seconds)))
        sample = max(4096, (sample // 4096) * 4096) if n_host >= 4096 else n_host
        tc0 = time.perf_counter()
        ref, ref_stats, s_steps, s_tests, threads = ctx.traverse(rays_host[:sample], any_hit=any_hit, want_stats=True, nthreads=cpu_threads)
        cpu_s = time.perf_counter() - tc0
        # one host thread: the closest analogue of what a GLua script gets today (one ray per call, serial).
        # Timed like the multi-thread leg: a slice of the same sample, walked once to warm the caches, then timed.
        # ... on 32 runs of consecutive rays spread evenly over the sample: the same mix of rays AND the same locality between
        # neighbours as the multi-thread leg sees (a strided pick would lose the locality, a prefix the mix)
        one_n = min(sample, 1 << 17)
        run = max(1, one_n // 32)
        starts = np.linspace(0, max(0, sample - run), num=min(32, max(1, sample // run)), dtype=np.int64)
        one_rays = np.ascontiguousarray(np.concatenate([rays_host[int(s0): int(s0) + run] for s0 in starts]))
        one_n = len(one_rays)
        ctx.traverse(one_rays, any_hit=any_hit, nthreads=1)
        t10 = time.perf_counter()
        ctx.traverse(one_rays, any_hit=any_hit, nthreads=1)
        one_thread = one_n / (time.perf_counter() - t10) / 1e6
        if any_hit:
            occ = d_hits[:sample].cpu().numpy()
            same_prim = bool((occ == (ref["prim"] != 0xFFFFFFFF)).all())
            same_tuv = True                                   # any-hit reports one byte
        else:
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
            "logical_cpus": host_threads,
            "build": ("gcc -O3 -march=x86-64-v3 -ffp-contract=off (bit-identical to the -O2 baseline-x86-64 build: checked on %d rays)" % chk)
                     if ctx.fast else "gcc -O2 -ffp-contract=off (baseline x86-64)",
            "threads_pinned": f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
            "numa_replicas": ctx.replicas,
            "one_thread_value": round(one_thread, 4),
            "scaling_vs_one_thread": round(sample / cpu_s / 1e6 / one_thread, 2) if one_thread > 0 else None,
            "one_thread_how": f"{one_n} rays (32 runs of consecutive rays spread evenly over the same sample), one pinned thread, second of two consecutive walks",
        }
        result["parity_sample"] = {"rays": sample, ("occluded_equal" if any_hit else "prim_bit_exact"): same_prim, "tuv_bit_exact": same_tuv,
                                   "counters_equal": same_stats}
        ctx.close()
        if rig:
            O.set_alpha()
        if not (same_prim and same_tuv):
            log("[bench] PARITY FAILURE on the sample")
    # ---- N > 1: ONE batch in K pieces (grouped ncclSend / ncclRecv per piece) -- last, and behind a watchdog: this path has run on
    # hardware with one rank only, and a collective that never completes must not cost the line its measured figures
    pieces_hung = False
    if dist_on and native is not None and n > 0 and dist_breakdown is not None:
        import threading

        def pieces():
            torch.cuda.set_device(device)
            for K in (2, 4, 8):
                single_batch(K)
        th = threading.Thread(target=pieces, daemon=True)
        th.start()
        th.join(timeout=args.pieces_timeout)
        pieces_hung = th.is_alive()
        dist_breakdown["single_batch_ms"] = dict(single)
        if pieces_hung:
            dist_breakdown["single_batch_note"] = f"the measurement in pieces did not finish within {args.pieces_timeout} s and was abandoned (K = 1 is the path of the timed steps)"
            log(f"[bench] rank {rank}: single batch in pieces timed out; leaving without collective teardown")
            if rank == 0:
                line = (json.dumps(result) + "\n").encode()
                os.write(result_fd if result_fd is not None else 1, line)
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
    def emit_result():
        line = (json.dumps(result) + "\n").encode()
        if result_fd is not None:
            os.write(result_fd, line)                      # the real stdout (everything else went to stderr)
        else:
            sys.stdout.write(line.decode()); sys.stdout.flush()
    if dist_on:
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)                 # C stdio of RCCL (its banner), now headed for stderr
        except Exception:
            pass
        dist.barrier()
    if rank == 0:
        emit_result()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
