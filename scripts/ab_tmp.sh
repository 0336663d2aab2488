N=$PWD/vistrace_amd/lib/variants/libvistrace_hip_nodefer.so
for b in ploc sah; do for e in 10 12 14; do
VISTRACE_HIP_LIB=$N python scripts/kernel_time.py --work "S1M:bounce" --builder $b --opt lds_entries=$e --tag ${b}_e$e 2>&1 | grep -E "median|rror"
done; done
