L=$PWD/vistrace_amd/lib/libvistrace_hip.so
N=$PWD/vistrace_amd/lib/variants/libvistrace_hip_nodefer.so
for r in 1 2; do
VISTRACE_HIP_LIB=$N python scripts/kernel_time.py --work "S1M:bounce" --tag nodefer_t4 2>&1 | grep -E "median|rror"
VISTRACE_HIP_LIB=$L python scripts/kernel_time.py --work "S1M:bounce" --opt tri_threshold=10 --tag defer_t10 2>&1 | grep -E "median|rror"
for e in 6 7 8; do
VISTRACE_HIP_LIB=$L python scripts/kernel_time.py --work "S1M:bounce" --opt tri_threshold=10 --opt lds_entries=$e --tag defer_t10_e$e 2>&1 | grep -E "median|rror"
VISTRACE_HIP_LIB=$N python scripts/kernel_time.py --work "S1M:bounce" --opt lds_entries=$e --tag nodefer_e$e 2>&1 | grep -E "median|rror"
done; done
