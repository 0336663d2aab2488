#!/usr/bin/env python3
"""Per-workload kernel statistics from a rocprofv3 kernel TRACE (not its --stats roll-up, which averages every launch of a kernel
in the process -- round 5's kernel_stats.csv row mixed the camera pass that seeds the bounce rays with the headline launches).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-pmc --alt-builder none --legs off
    python3 scripts/kernel_stats_headline.py DIR profiles/r6/kernel_stats_headline.csv

bench.py's launches of the dominant kernel, in dispatch order: ONE camera pass (primary rays; its hits seed the bounce batch), then
the headline batch only -- warm-up, three single launches, the K timed steps, the counters / parity launches use other kernels.
The first dispatch is dropped (and reported on its own line); the row that remains is the headline batch and nothing else."""
import csv
import glob
import os
import statistics
import sys

src, out = sys.argv[1], sys.argv[2]
kernel = sys.argv[3] if len(sys.argv) > 3 else "trace_kernel<false,false,true,true,false>"
files = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
assert files, f"no *_kernel_trace.csv under {src}"
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
mine = [(e - s) for s, e, k in rows if kernel in k.replace(" ", "")]
assert len(mine) >= 3, f"{len(mine)} dispatches of {kernel}"
seed, mine = mine[0], mine[1:]
mean = statistics.mean(mine)
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "MinOverMean", "MaxOverMean", "What"])
    w.writerow([f"vt::{kernel}", len(mine), sum(mine), round(mean, 1), min(mine), max(mine), round(statistics.pstdev(mine), 1),
                round(min(mine) / mean, 4), round(max(mine) / mean, 4), "the headline batch only (warm-up + single + timed launches)"])
    w.writerow([f"vt::{kernel}", 1, seed, seed, seed, seed, 0, 1, 1, "the camera pass that seeds the bounce rays (dropped from the row above)"])
    others = {}
    for s, e, k in rows:
        if kernel in k.replace(" ", ""):
            continue
        others.setdefault(k, []).append(e - s)
    for k, v in sorted(others.items(), key=lambda kv: -sum(kv[1])):
        m = statistics.mean(v)
        w.writerow([k, len(v), sum(v), round(m, 1), min(v), max(v), round(statistics.pstdev(v), 1), round(min(v) / m, 4), round(max(v) / m, 4), ""])
print(open(out).read().splitlines()[1])
