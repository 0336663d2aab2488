for r in 1 2 3; do
for v in base pipe; do
if [ $v = base ]; then L=$PWD/vistrace_amd/lib/libvistrace_hip.so; else L=$PWD/vistrace_amd/lib/variants/libvistrace_hip_$v.so; fi
VISTRACE_HIP_LIB=$L python scripts/kernel_time.py --work "S1M:bounce,S1M:primary" --tag $v 2>&1 | grep -E "median|rror" | cut -c1-100
done; done
