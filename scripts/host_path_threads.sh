#!/bin/bash
# vt_trace_closest on host arrays against the number of staging-copy threads (VT_COPY_THREADS, dev knob of parallel_copy)
cd "$(dirname "$0")/.."
for t in 4 8 16 32 64; do echo "copy threads $t:"; VT_COPY_THREADS=$t timeout 300 python scripts/host_path_rate.py 2>&1 | grep "rays in"; done
