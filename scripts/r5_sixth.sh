#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r5f; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "alpha or image_width" > $O/pytest_alpha.log 2>&1; tail -3 $O/pytest_alpha.log
# camera rays into an alpha-tested scene: previous library (hint ignored for alpha scenes: persistent, untiled) against this tree (one ray per lane, tiled)
for r in 1 2 3; do
 for v in prev new; do
  if [ $v = prev ]; then export VISTRACE_HIP_LIB=$R/vistrace_amd/lib/variants/libvistrace_hip_prev.so; else unset VISTRACE_HIP_LIB; fi
  for W in "--kind primary" "--kind primary --scene S100k --side 1024" "--kind bounce"; do
   python3 bench.py $W --alpha-frac 0.3 --steps 100 --warmup 5 --no-pmc --no-cpu --alt-builder none --legs off 2>/dev/null | grep '^{' | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['workload'], 'ms', d['ms_per_step'], 'Mrays/s', d['value'], d['config']['kernel_mode'], d['config']['launch_options'].get('alpha_threshold'))"
  done
 done
done 2>&1 | tee $O/alpha_ab.txt
