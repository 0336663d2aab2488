#!/bin/bash
# one ray per lane against persistent waves by scene size, ray kind and batch size (the engine's auto rule, plan_launch in engine.hip)
cd "$(dirname "$0")/.."
for scene in S10k S100k S1M; do
  for kind in primary bounce; do
    for side in 512 1024 2048 4096; do
      for p in 0 1; do
        w=0; [ $kind = primary ] && w=$side
        echo -n "$scene $kind side $side persistent $p : "
        timeout 300 python scripts/kernel_time.py --work $scene:$kind --side $side --reps 9 --opt persistent=$p --opt ray_image_width=$w 2>&1 | grep -o "median [0-9.]* min [0-9.]*"
      done
    done
  done
done
