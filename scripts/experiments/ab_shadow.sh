for r in 1 2; do for v in base any1 any2; do
  if [ $v = base ]; then L=vistrace_amd/lib/libvistrace_hip.so; else L=vistrace_amd/lib/variants/libvistrace_hip_$v.so; fi
  echo -n "$v: "; VISTRACE_HIP_LIB=$PWD/$L timeout 400 python bench.py --kind shadow --side 2048 --steps 30 --no-cpu --no-pmc --alt-builder none 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('parity_sample'))"
done; done
