#!/usr/bin/env python3
"""Per-workload kernel statistics from a rocprofv3 kernel TRACE (not its --stats roll-up, which averages every launch of a kernel
in the process -- round 5's kernel_stats.csv row mixed the camera pass that seeds the bounce rays with the headline launches).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --alt-builder none --legs off
    python3 scripts/kernel_stats_headline.py DIR profiles/r6/kernel_stats_headline.csv

bench.py's launches of the dominant kernel, in dispatch order: ONE camera pass (primary rays; its hits seed the bounce batch), then
the headline batch only -- W warm-up launches, three single launches, the K timed steps (the counters / parity launches use other
kernels).  One row per phase; the headline row is the K launches of the timed region and nothing else, and the per-dispatch
durations are written out so that the rows can be re-derived.  Usage: ... DIR OUT.csv [--warmup W] [--steps K] [kernel]"""
import csv
import glob
import os
import statistics
import sys

argv = sys.argv[1:]
W, K = 5, 20
for flag in ("--warmup", "--steps"):
    if flag in argv:
        i = argv.index(flag)
        if flag == "--warmup":
            W = int(argv[i + 1])
        else:
            K = int(argv[i + 1])
        del argv[i:i + 2]
src, out = argv[0], argv[1]
kernel = argv[2] if len(argv) > 2 else "trace_kernel<false,false,true,true,false>"
files = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
assert files, f"no *_kernel_trace.csv under {src}"
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mine = [(e - s) for s, e, k in rows if kernel in k.replace(" ", "")]
assert len(mine) == 1 + W + 3 + K, f"{len(mine)} dispatches of {kernel}, expected 1 camera pass + {W} warm-up + 3 single + {K} timed"
phases = [("the camera pass that seeds the bounce rays (another batch)", mine[:1]),
          (f"warm-up: the first {W} launches of the headline batch (cold caches, launch-slot allocations)", mine[1:1 + W]),
          ("three single launches (HIP events around each)", mine[1 + W:1 + W + 3]),
          (f"THE TIMED REGION: {K} launches of the headline batch and nothing else", mine[1 + W + 3:])]
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "MinOverMean", "MaxOverMean", "What", "DurationsNs"])
    for what, v in reversed(phases):
        m = statistics.mean(v)
        w.writerow([f"vt::{kernel}", len(v), sum(v), round(m, 1), min(v), max(v), round(statistics.pstdev(v), 1), round(min(v) / m, 4), round(max(v) / m, 4),
                    what, " ".join(str(x) for x in v)])
    others = {}
    for s_, e_, k in rows:
        if kernel in k.replace(" ", ""):
            continue
        others.setdefault(k, []).append(e_ - s_)
    for k, v in sorted(others.items(), key=lambda kv: -sum(kv[1])):
        m = statistics.mean(v)
        w.writerow([k, len(v), sum(v), round(m, 1), min(v), max(v), round(statistics.pstdev(v), 1), round(min(v) / m, 4), round(max(v) / m, 4), "", ""])
print("\n".join(ln[:260] for ln in open(out).read().splitlines()[1:5]))
