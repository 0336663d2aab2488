#!/bin/bash
# tri_threshold x refill_threshold on the headline workload (engine options: no rebuild), median kernel ms of 9 launches each
cd "$(dirname "$0")/.."
for t in 2 4 6 8 16 32; do for r in 4 8 16 32; do
  echo -n "tri_threshold $t refill_threshold $r : "
  timeout 300 python scripts/kernel_time.py --work ${1:-S1M:bounce} --reps 9 --opt tri_threshold=$t --opt refill_threshold=$r 2>&1 | grep -o "median [0-9.]* min [0-9.]* ms  chk [0-9a-f]*"
done; done
