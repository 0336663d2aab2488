#!/usr/bin/env python3
"""Writes section 5 of BASELINE.md from profiles/r6/*_bench_line.json, so that the document follows the committed evidence figure
for figure (one table for the current round; earlier rounds: profiles/r1..r5/ and their notes).  python scripts/r6_tables.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r6")


def L(name):
    with open(os.path.join(P, f"{name}_bench_line.json")) as f:
        return json.load(f)


def num(x, nd=0):
    return f"{x:,.{nd}f}".replace(",", " ")


def row(label, args, name):
    d = L(name)
    r = d["roofline"]
    gp = r.get("gather_path") or {}
    cpu = d.get("cpu_baseline")
    par = d.get("parity_sample") or {}
    ok = all(v is True for k, v in par.items() if k != "rays") if par else None
    chk = (f"{par['rays']:,} rays bit-equal".replace(",", " ") if ok else ("**PARITY FAILURE**" if par else "—"))
    two = (d.get("two_streams") or {}).get("value")
    return (f"| {label} | `{args}` | {d['ms_per_step']:.4g} | {num(d['value'])}" + (f" ({num(two)})" if two else "") +
            f" | {r['steps_per_ray']:.1f} / {r['tests_per_ray']:.1f} | {r['alg_bytes_per_ray']:.0f} | {(r.get('traffic') or 0) / 1e9:.2f} | "
            f"{r.get('frac') if r.get('frac') is not None else '—'} | {r.get('valu_busy_frac', '—')} {(r.get('bound_actual') or {}).get('valu_busy_bounds_from_counters', '')} / {r.get('lane_utilisation', '—')} | "
            f"{r.get('l1_frac_of_64B_per_clk', '—')} | {chk} |")


def main():
    h = L("headline")
    cpu, rb, hi, bc, ml = h["cpu_baseline"], h["rebuild"], h["host_inclusive"], h["beyond_cache"], h["merged_launch"]
    rb10 = L("s10m").get("rebuild")
    f1 = L("forcedist_1rank")
    out = []
    out.append("## 5. Results\n")
    out.append("### Round 6 (one MI355X; `profiles/r6/`: every row is a `bench.py` line with its own live PMC passes, ONE run of "
               "`scripts/profile_r6.sh` on one box with the round's final sources; written by `scripts/r6_tables.py`)\n")
    out.append("Earlier rounds: `profiles/r1` … `r5` (bench lines, counters, notes).  The traversal kernels are round 3's, instruction for "
               "instruction (round 6 measured one more bookkeeping change and took it out again: `profiles/r6/notes.md` section 5); round 6 "
               "worked on what makes the claims checkable (recall sensitivity of the oracle, kernel mutants, fault injection, counters) and on "
               "the bounce loop and the group-wide Rebuild.  `frac` = (FETCH_SIZE + WRITE_SIZE) × 1024 B ÷ launch time ÷ 8 TB/s; algorithmic "
               "bytes = 32 + 16 + 64·steps + 64·tests per ray (1 result byte for any-hit).  VALU busy = the model (instruction-class counts × "
               "measured issue costs ÷ the launch's cycles at its effective clock) with the bracket the counters alone give.\n")
    out.append("| config | `bench.py` arguments | ms per step | Mrays/s (two streams) | steps / tests per ray | alg. B per ray | fabric GB per launch | "
               "**`frac`** | VALU busy (model) [counter bracket] / lane use | L1 accesses per clk and CU | parity on the timed rays |")
    out.append("|---|---|---|---|---|---|---|---|---|---|---|")
    out.append(row("**3 = headline**: S1M, 16 Mi bounce rays, closest hit", "(default)", "headline"))
    out.append(row("3 on the reference-algorithm tree", "--builder ploc", "ploc"))
    out.append(row("2: S100k, 1 Mi camera rays", "--scene S100k --kind primary --side 1024", "config2"))
    out.append(row("4: S1M, 64 Mi shadow rays, any-hit", "--kind shadow", "config4"))
    out.append(row("5 at N = 1: S10M, 128 camera tiles", "--scaling strong --scene S10M --tiles 128", "strong_s10m_128tiles"))
    out.append(row("S1M, 16 Mi camera rays", "--kind primary", "primary_s1m"))
    out.append(row("S1M, 30 % alpha-tested triangles", "--alpha-frac 0.3", "alpha30"))
    out.append(row("S10M, 16 Mi bounce rays (beyond every cache)", "--scene S10M", "s10m"))
    out.append("")
    out.append(f"config 1 (`accel:Traverse`, one ray per call): host walk, 0.3–0.4 µs per call through the binding (`profiles/r6/binding_bench.txt`).  "
               f"`cpu_baseline` of the headline run: **{cpu['value']:.2f} Mrays/s on {cpu['cores']} threads** ({cpu['one_thread_value']:.2f} on one; {cpu['cpu']}; "
               f"kind `{cpu['kind']}`: {cpu['sample']}).  `host_inclusive` (PCIe inside the call): {num(hi['value'])} Mrays/s pageable, "
               f"{num(hi['page_locked_arrays']['value'])} page-locked.  16 sets of rays: {num(ml['separate_launches_value'])} Mrays/s as 16 launches, "
               f"{num(ml['one_merged_launch_value'])} as one merged launch.\n")
    out.append("**Rebuild** (`rebuild` leg; ref `VisTrace.cpp:798-818` → `AccelStruct.cpp:762-775`), host steps on the lease's "
               f"{rb['host_threads']} cores:\n")
    out.append("| scene | `vt_tris_setup` | `vt_bvh_build` (CPU) | re-pack + upload, host lineariser (rounds 1-4) | **re-pack on the device** (`vt_scene_upload_tree`) | "
               "bytes host→device, link time at 56 GB/s | records byte-equal | host copy back (lazy) | refit / skin refit |")
    out.append("|---|---|---|---|---|---|---|---|---|")
    for r in (rb, rb10):
        if not r:
            continue
        hr, dr = r["host_repack"], r["device_repack"]
        out.append(f"| {r['scene']} ({num(r['triangles'])} tris) | {r['tris_setup_ms']} ms | {r['bvh_build_ms']} ms | {hr['vt_scene_linearise_ms']} + {hr['vt_scene_upload_ms']} = "
                   f"{hr['sum_ms']} ms | **{dr['vt_scene_upload_tree_ms']} ms** ({dr['over_link_time']} × link) | {dr['bytes_h2d'] / 1e6:.0f} MB, {dr['link_time_ms_at_56GBs']} ms | "
                   f"{r['records_byte_equal']} | {r['vt_host_scene_download_ms']} ms | {r['vt_scene_refit_ms']} / {r['vt_scene_skin_refit_ms']} ms |")
    out.append("")
    out.append("**N > 1** (no multi-GPU node exists here; frozen after round 6 until a SCALE record exists; the single-process group's lines "
               "on a simulated group: `profiles/r5/`).  One rank through the N > 1 control flow of the per-rank form, real RCCL (`--force-dist`), "
               "headline shard, 32 CUs reserved:\n")
    out.append("| what ran | ms per step | Mrays/s | trace ms per member (min–max) | gather ms | one isolated batch in 1 / 2 / 4 / 8 pieces (ms) |")
    out.append("|---|---|---|---|---|---|")

    def drow(label, d):
        bd = d["config"]["dist_breakdown"]
        g = bd.get("gather_ms_per_rank") or {}
        sb = bd.get("single_batch_ms") or {}
        return (f"| {label} | {d['ms_per_step']:.4g} | {num(d['value'])} | {bd['trace_ms_per_rank']['min']:.2f}–{bd['trace_ms_per_rank']['max']:.2f} | "
                f"{g.get('max', '—')} | {' / '.join(str(sb.get(k, '—')) for k in ('1', '2', '4', '8'))} |")
    out.append(drow("`--force-dist`, one rank", f1))
    out.append("")
    for name, title in (("group_update_rate.txt", "Group-wide Rebuild / refit, members sharing one GPU (`scripts/group_update_rate.py`)"),
                        ("bounce_loop_rate.txt", "Bounce loop, 16 Mi paths x depth 4 (`scripts/bounce_loop_rate.py`)")):
        path = os.path.join(P, name)
        if os.path.exists(path):
            out.append(f"**{title}**\n\n```\n" + open(path).read().strip() + "\n```\n")
    text = "\n".join(out) + "\n"
    path = os.path.join(ROOT, "BASELINE.md")
    src = open(path).read()
    cut = src.index("## 5. Results")
    open(path, "w").write(src[:cut] + text)
    print(text)


if __name__ == "__main__":
    main()
