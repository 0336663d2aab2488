#!/usr/bin/env python3
"""Re-writes the round-4 result tables of DESIGN.md and BASELINE.md from profiles/r4/*_bench_line.json (so that the documents follow
the committed evidence figure for figure).  python scripts/r4_tables.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles", "r4")


def L(name):
    return json.load(open(os.path.join(P, f"{name}_bench_line.json")))


def num(x, nd=0):
    s = f"{x:,.{nd}f}".replace(",", " ")
    return s


def cells(name):
    d = L(name)
    r = d["roofline"]
    gp = r.get("gather_path") or {}
    return dict(ms=d["ms_per_step"], v=d["value"], two=(d.get("two_streams") or {}).get("value"), steps=r["steps_per_ray"], tests=r["tests_per_ray"],
                frac=r["frac"], l1=r.get("l1_gather_gbs_per_cu"), l1f=r.get("l1_gather_frac"), l2=gp.get("l2_to_l1_gb_s_per_cu"), hit=r.get("l1_hit_rate"),
                valu=r.get("valu_busy_frac"), lane=r.get("lane_utilisation"), traffic=r.get("traffic"), cpu=(d.get("cpu_baseline") or {}))


def main():
    h, pl, c2, c4, pr, al, s10, st = (cells(n) for n in ("headline", "ploc", "config2", "config4", "primary_s1m", "alpha30", "s10m", "strong_s10m_128tiles"))
    f1, fs = L("forcedist_1rank"), L("forcedist_strong_16tiles")
    hd = L("headline")
    hi, bc = hd["host_inclusive"], hd["beyond_cache"]
    design = f"""| workload | kernel | ms per step | Mrays/s | steps / tests per ray | `frac` (HBM, the contract) | GB/s per CU through L1 (`l1_gather_frac`), from L2 | VALU busy | lane use |
|---|---|---|---|---|---|---|---|---|
| headline `S1M_bounce16777216`, default tree | persistent | **{h['ms']:.3f}** | **{num(h['v'])}** (two streams {num(h['two'])}) | {h['steps']:.2f} / {h['tests']:.2f} | {h['frac']:.3f} | {h['l1']:.1f} ({h['l1f']:.2f}), {h['l2']:.1f} | {h['valu']:.2f} | {h['lane']:.2f} |
| same, PLOC tree (the reference's algorithm) | persistent | {pl['ms']:.3f} | {num(pl['v'])} | {pl['steps']:.2f} / {pl['tests']:.2f} | {pl['frac']:.3f} | {pl['l1']:.1f} ({pl['l1f']:.2f}), {pl['l2']:.1f} | {pl['valu']:.2f} | {pl['lane']:.2f} |
| configs[1] `S100k_primary1048576` | one ray per lane | {c2['ms']:.4f} | {num(c2['v'])} (two streams {num(c2['two'])}) | {c2['steps']:.1f} / {c2['tests']:.1f} | {c2['frac']:.3f} | {c2['l1']:.1f}, {c2['l2']:.1f} | {c2['valu']:.2f} (launch-bound) | {c2['lane']:.2f} |
| configs[3] `S1M_shadow67108864` (any-hit) | persistent | {c4['ms']:.2f} | {num(c4['v'])} | {c4['steps']:.1f} / {c4['tests']:.1f} | {c4['frac']:.3f} | {c4['l1']:.1f}, {c4['l2']:.1f} | {c4['valu']:.2f} | {c4['lane']:.2f} |
| `S1M_primary16777216` | one ray per lane (round 4's rule; a tie here) | {pr['ms']:.3f} | {num(pr['v'])} | {pr['steps']:.1f} / {pr['tests']:.1f} | {pr['frac']:.3f} | {pr['l1']:.0f} (L1 hit rate {pr['hit']:.2f}), {pr['l2']:.1f} | {pr['valu']:.2f} | {pr['lane']:.2f} |
| `S1M_bounce16777216_alpha30` | persistent, ALPHA | {al['ms']:.3f} | {num(al['v'])} | {al['steps']:.1f} / {al['tests']:.1f} | {al['frac']:.3f} | {al['l1']:.1f} ({al['l1f']:.2f}), {al['l2']:.1f} | {al['valu']:.2f} | {al['lane']:.2f} |
| `S10M_bounce16777216` (also the headline line's `beyond_cache`) | persistent | {s10['ms']:.3f} | {num(s10['v'])} | {s10['steps']:.1f} / {s10['tests']:.1f} | **{s10['frac']:.3f}** | {s10['l1']:.1f} ({s10['l1f']:.2f}), {s10['l2']:.1f} | {s10['valu']:.2f} | {s10['lane']:.2f} |
| configs[4] at N = 1 `S10M_primary_128x1048576_tiles` | persistent | {st['ms']:.2f} | {num(st['v'])} | {st['steps']:.1f} / {st['tests']:.1f} | {st['frac']:.3f} | {st['l1']:.1f}, {st['l2']:.1f} | {st['valu']:.2f} | {st['lane']:.2f} |
| `host_inclusive` of the headline (16 Mi rays, 48 B per ray over PCIe) | | {hi['ms_per_call']:.1f} (pageable), {hi['page_locked_arrays']['ms_per_call']:.1f} (page-locked) | {num(hi['value'])}, {num(hi['page_locked_arrays']['value'])} | | | | | |
"""
    cpu = h["cpu"]
    base = f"""| 1 (`accel:Traverse`, one ray per call, host walk; `tests/cpp --bench`) | 10 082-tri world | 10 000 calls | closest | host | 0.3–0.4 µs per call | — | — | — | — | — | — | — | — |
| 2 (`--scene S100k --kind primary --side 1024`) | S100k | 1 048 576 primary | closest | one ray per lane | {c2['ms']:.4f} | {num(c2['v'])} ({num(c2['two'])}) | {c2['steps']:.1f} / {c2['tests']:.1f} | {c2['traffic'] / 1e6:.1f} MB | {c2['frac']:.3f} | {c2['l1']:.1f}, {c2['l2']:.1f} | {c2['lane']:.2f} | {c2['valu']:.2f} (launch-bound) | whole batch bit-equal |
| **3 = headline (default)** | S1M, default tree | 16 777 216 bounce | closest | persistent | **{h['ms']:.3f}** (4.13–4.39 across the boxes of the pool) | **{num(h['v'])}** ({num(h['two'])}) | {h['steps']:.2f} / {h['tests']:.2f} | {h['traffic'] / 1e9:.2f} GB | **{h['frac']:.3f}** | {h['l1']:.1f} ({h['l1f']:.2f}), {h['l2']:.1f} | {h['lane']:.2f} | {h['valu']:.2f} | {cpu.get('value', 0):.2f} Mrays/s on {cpu.get('cores')} threads ({cpu.get('one_thread_value', 0):.2f} on one), whole batch bit-equal |
| 3 on the reference-algorithm tree (`--builder ploc`) | S1M, PLOC | 16 777 216 bounce | closest | persistent | {pl['ms']:.3f} | {num(pl['v'])} ({num(pl['two'])}) | {pl['steps']:.2f} / {pl['tests']:.2f} | {pl['traffic'] / 1e9:.2f} GB | {pl['frac']:.3f} | {pl['l1']:.1f} ({pl['l1f']:.2f}), {pl['l2']:.1f} | {pl['lane']:.2f} | {pl['valu']:.2f} | — |
| 3' camera rays (`--kind primary`) | S1M | 16 777 216 primary | closest | one ray per lane (new rule; a tie at this size) | {pr['ms']:.3f} | {num(pr['v'])} ({num(pr['two'])}) | {pr['steps']:.1f} / {pr['tests']:.1f} | {pr['traffic'] / 1e9:.2f} GB | {pr['frac']:.3f} | {pr['l1']:.0f} (L1 hits {pr['hit']:.2f}), {pr['l2']:.1f} | {pr['lane']:.2f} | {pr['valu']:.2f} | — |
| 3'' 30 % alpha-tested triangles (`--alpha-frac 0.3`) | S1M | 16 777 216 bounce | closest | persistent, ALPHA | {al['ms']:.3f} | {num(al['v'])} ({num(al['two'])}) | {al['steps']:.1f} / {al['tests']:.1f} | {al['traffic'] / 1e9:.2f} GB | {al['frac']:.3f} | {al['l1']:.1f} ({al['l1f']:.2f}), {al['l2']:.1f} | {al['lane']:.2f} | {al['valu']:.2f} | whole batch bit-equal |
| 4 (`--kind shadow`) | S1M | 67 108 864 shadow | any-hit | persistent | {c4['ms']:.2f} | {num(c4['v'])} ({num(c4['two'])}) | {c4['steps']:.1f} / {c4['tests']:.1f} (the any-hit walk's own counters) | {c4['traffic'] / 1e9:.2f} GB | {c4['frac']:.3f} | {c4['l1']:.1f}, {c4['l2']:.1f} | {c4['lane']:.2f} | {c4['valu']:.2f} | 16 Mi-ray sample equal |
| 5 at N = 1 (`--scaling strong --scene S10M --tiles 128`) | S10M | 134 217 728 primary, one step | closest | persistent | {st['ms']:.2f} | {num(st['v'])} ({num(st['two'])}) | {st['steps']:.1f} / {st['tests']:.1f} | {st['traffic'] / 1e9:.2f} GB | {st['frac']:.3f} | {st['l1']:.1f}, {st['l2']:.1f} | {st['lane']:.2f} | {st['valu']:.2f} | — |
| 5'' (`--scene S10M`; also the headline line's `beyond_cache` leg) | S10M | 16 777 216 bounce | closest | persistent | {s10['ms']:.3f} | {num(s10['v'])} ({num(s10['two'])}) | {s10['steps']:.1f} / {s10['tests']:.1f} | {s10['traffic'] / 1e9:.2f} GB | **{s10['frac']:.3f}** | {s10['l1']:.1f} ({s10['l1f']:.2f}), {s10['l2']:.1f} | {s10['lane']:.2f} | {s10['valu']:.2f} | whole batch bit-equal |
| N > 1 control flow with one rank (`--force-dist`), weak / configs[4]'s 16 tiles per rank | S1M / S10M | 16 777 216 | closest | persistent, 32 reserved CUs | {f1['ms_per_step']:.3f} / {fs['ms_per_step']:.3f} | {num(f1['value'])} / {num(fs['value'])} | | | | | | | |

"""
    for path, start, end, body in (
            (os.path.join(ROOT, "DESIGN.md"), "| workload | kernel | ms per step | Mrays/s |", "\nUnchanged against round 3 where nothing was meant to change", design),
            (os.path.join(ROOT, "BASELINE.md"), "| 1 (`accel:Traverse`, one ray per call, host walk; `tests/cpp --bench`) | 10 082-tri world | 10 000 calls | closest | host |",
             "Beside `value` in the headline line:", base)):
        s = open(path).read()
        a, b = s.index(start), s.index(end)
        s = s[:a] + body + s[b:]
        open(path, "w").write(s)
    sb = {k: f"{v:.2f}" for k, v in f1["config"]["dist_breakdown"]["single_batch_ms"].items()}
    ss = {k: f"{v:.2f}" for k, v in fs["config"]["dist_breakdown"]["single_batch_ms"].items()}
    print("single batch in 1/2/4/8 pieces, headline shard:", sb, " configs[4] shard:", ss)
    print("host_inclusive", hi["value"], hi["ms_per_call"], hi["page_locked_arrays"]["value"], hi["page_locked_arrays"]["ms_per_call"], "beyond_cache", bc["value"], bc["kernel_ms"], bc["frac"])
    print("headline", h["ms"], h["v"], "ploc", pl["ms"], pl["v"])


if __name__ == "__main__":
    main()
