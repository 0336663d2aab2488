#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: per kernel name, last dispatch + mean."""
import collections, csv, glob, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "trace_kernel<false, false"
for d in sorted(glob.glob(root + "/pmc_*/")):
    for f in glob.glob(d + "**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(f"{d.split('/')[-2]:10s} {k:40s} n={len(v)} last={v[-1]:.5g} mean={sum(v)/len(v):.5g}")
