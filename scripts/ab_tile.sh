run() { tag=$1; lib=$2; shift 2; echo -n "$tag: "; VISTRACE_HIP_LIB=$PWD/$lib timeout 300 python scripts/kernel_time.py "$@" 2>&1 | grep median | sed 's/default //' | tr "\n" " "; echo; }
OLD=vistrace_amd/lib/variants/libvistrace_hip_old.so; NEW=vistrace_amd/lib/libvistrace_hip.so
for r in 1 2; do
run old4096 $OLD --work S1M:bounce,S1M:primary,S10M:primary
run new4096 $NEW --work S1M:bounce,S1M:primary,S10M:primary
run hint4096 $NEW --work S1M:bounce,S1M:primary,S10M:primary --opt ray_image_width=4096
run old1024 $OLD --work S100k:primary,S100k:bounce --side 1024 --reps 41
run new1024 $NEW --work S100k:primary,S100k:bounce --side 1024 --reps 41
run hint1024 $NEW --work S100k:primary,S100k:bounce --side 1024 --reps 41 --opt ray_image_width=1024
done
