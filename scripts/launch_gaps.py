#!/usr/bin/env python3
"""Gaps between consecutive trace launches of a bench.py run, from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace --output-format csv -d out -o kt -- python3 bench.py --steps 30 --no-cpu --no-pmc --alt-builder none
    python scripts/launch_gaps.py out/*/kt_kernel_trace.csv"""
import csv
import sys

rows = []
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
prev = None
gaps, durs = [], []
for k, (s, e, name) in enumerate(rows):
    if "trace_kernel" in name:
        if prev is not None and k - prev[2] <= 2:      # trace, (cursor fill,) trace
            gaps.append((s - prev[1]) / 1e3)
        durs.append((e - s) / 1e3)
        prev = (s, e, k)
import statistics as st
print(f"{len(durs)} trace launches, median duration {st.median(durs):.1f} us; {len(gaps)} back-to-back pairs, "
      f"gap end->start median {st.median(gaps):.1f} us, min {min(gaps):.1f}, max {max(gaps):.1f}")
for k, (s, e, name) in enumerate(rows[-9:]):
    print(f"  {(s - rows[-9][0]) / 1e3:10.1f} .. {(e - rows[-9][0]) / 1e3:10.1f} us  {name[:60]}")
