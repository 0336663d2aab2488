#!/usr/bin/env python3
"""Copy the judged summaries of a scripts/profile.sh run into profiles/<round>/.

    python scripts/collect_profile.py gpurun_out/prof_r1b profiles/r1

Writes kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_summary.txt (one line per
counter: last dispatch = one timed step of the bench workload), bench_line.json (the JSON line
bench.py printed under the profiler) and traffic.json (fabric-side bytes per launch:
FETCH_SIZE x 1024 + WRITE_SIZE x 1024.  The gfx950 x2 correction of MI355X_MICROARCH.md "HBM" applies to wide
streaming reads (128-B requests tallied at 64 B); this kernel fetches random 64-B records with 64-B requests,
and scripts/calib_fetch.hip measures FETCH_SIZE x 1024 / true bytes = 1.000 for exactly that pattern
(profiles/r1/calib_fetch.txt), so no doubling is applied).
"""
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc_summary.txt"), os.path.join(dst, "pmc_summary.txt"))
line = {}
try:
    line = json.loads(open(os.path.join(src, "bench_line.json")).read())
    json.dump(line, open(os.path.join(dst, "bench_line.json"), "w"), indent=1)
except (OSError, ValueError):
    pass
vals = {}
for l in open(os.path.join(src, "pmc_summary.txt")):
    p = l.split()
    if len(p) >= 4:
        vals[p[1]] = float(p[3].split("=")[1])
if "FETCH_SIZE" in vals:
    fetch_kb, write_kb = vals["FETCH_SIZE"], vals.get("WRITE_SIZE", 0.0)
    out = {
        "workload": line.get("config", {}).get("workload"),
        "kernel": line.get("roofline", {}).get("kernel"),
        "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
        "hbm_bytes_per_launch": int(fetch_kb * 1024 + write_kb * 1024),
        "note": "per launch of the dominant kernel (last dispatch of the PMC pass); FETCH_SIZE x 1024 + WRITE_SIZE x 1024, "
                "calibrated for this access pattern (64-B random records: factor 1.000, profiles/r1/calib_fetch.txt); "
                "these are L2-miss (fabric-side) bytes: the 104 MB scene is Infinity-Cache resident, so true HBM "
                "traffic is lower still",
    }
    json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(out)
