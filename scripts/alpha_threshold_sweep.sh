#!/bin/bash
# alpha30 (16 Mi bounce rays into S1M, 30 % alpha-tested triangles) against engine option alpha_threshold; parity_sample must stay true
cd "$(dirname "$0")/.."
for K in 1 2 3 4 6 8 16 1 4; do
  python3 bench.py --alpha-frac 0.3 --steps 100 --warmup 5 --no-pmc --alt-builder none --legs off --cpu-seconds 3 --engine-opt alpha_threshold=$K 2>/dev/null | grep '^{' | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('alpha_threshold', $K, 'ms', d['ms_per_step'], 'Mrays/s', d['value'], 'parity', d.get('parity_sample'))"
done
