#!/usr/bin/env python3
"""The start-up and the drain of one persistent trace launch, from a VT_EXP_TIMELINE build:
    make -C vistrace_amd/csrc variant NAME=tl DEFS=-DVT_EXP_TIMELINE=1
    VT_TIMELINE_FILE=/tmp/tl.bin VISTRACE_HIP_LIB=$PWD/vistrace_amd/lib/variants/libvistrace_hip_tl.so python scripts/kernel_time.py --reps 3
    python scripts/timeline.py /tmp/tl.bin
Per wave: s_memrealtime (100 MHz) at its start, when it first found the ray cursor exhausted, at its exit; iterations in all and
after exhaustion; active lanes summed over those."""
import sys

import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t = t[t[:, 2] != 0]
tick = 0.01   # us
t0 = t[:, 0].min()
start, exh, end = (t[:, 0] - t0) * tick, (t[:, 1] - t0) * tick, (t[:, 2] - t0) * tick
it, tail_it, tail_lanes = t[:, 3], t[:, 4], t[:, 5]
print(f"{len(t)} waves; kernel span {end.max():.0f} us")
print(f"wave start: median {np.median(start):.1f} us, last {start.max():.1f} us")
print(f"cursor found exhausted: first {exh[t[:, 1] != 0].min():.0f} us, median {np.median(exh):.0f}, last {exh.max():.0f}")
print(f"wave exit: first {end.min():.0f} us, p10 {np.percentile(end, 10):.0f}, median {np.median(end):.0f}, p90 {np.percentile(end, 90):.0f}, "
      f"p99 {np.percentile(end, 99):.0f}, last {end.max():.0f}")
first_exh = exh[t[:, 1] != 0].min()
print(f"drain = last exit - first exhaustion = {end.max() - first_exh:.0f} us "
      f"({100 * (end.max() - first_exh) / end.max():.1f} % of the kernel)")
steady = (it - tail_it).sum() / ((exh - start).sum())      # iterations per us per wave before exhaustion
print(f"iterations per wave: {it.mean():.0f} (tail {tail_it.mean():.0f}); steady state {1 / steady:.2f} us per iteration, "
      f"tail {((end - exh).sum() / max(tail_it.sum(), 1)):.2f} us per iteration; lanes active in tail iterations: {tail_lanes.sum() / max(tail_it.sum(), 1):.1f}")
# live waves over the drain
edges = np.linspace(first_exh, end.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    live = ((end > a)).sum()
    print(f"  {a - first_exh:7.0f} us after exhaustion: {live:5d} waves live")
hw = t[:, 7]
wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print("HW_ID.WAVE_ID histogram:", np.bincount(wave_id, minlength=16).tolist())
print("SIMD histogram:", np.bincount(simd, minlength=4).tolist(), " SE:", np.bincount(se).tolist(), " SH:", np.bincount(sh).tolist(), " CU:", np.bincount(cu).tolist())
key = (t[:, 6] << 12) | (se << 8) | (sh << 7) | cu
blocks = np.arange(len(t)) // 4
print("distinct CUs:", len(np.unique(key)))
for k in np.unique(key)[:3]:
    sel = key == k
    print(f"  CU key {k:#x}: blocks {sorted(set(blocks[sel].tolist()))}  wave ids {sorted(wave_id[sel].tolist())} simd {np.bincount(simd[sel], minlength=4).tolist()}")
# waves of one block: are they on four SIMDs?
b4 = simd[: len(t) // 4 * 4].reshape(-1, 4)
print("blocks whose four waves sit on four different SIMDs:", int((np.sort(b4, 1) == np.arange(4)).all(1).sum()), "of", len(b4))
