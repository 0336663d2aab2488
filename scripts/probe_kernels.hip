// probe_kernels.hip -- dev helper for scripts/overlap_probe.py: a stand-in for a communication kernel
// (few blocks, a chosen LDS footprint, spins for a chosen time) and a CU-masked stream.
// Build: hipcc -O2 --offload-arch=gfx950 -shared -fPIC scripts/probe_kernels.hip -o scripts/_build/libprobe.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

template <bool FAT>
__global__ __launch_bounds__(256) void spin_kernel(long long cycles, uint32_t* sink)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    if constexpr (FAT) asm volatile("; register footprint of rcclGenericKernel (about 280 unified VGPRs)" ::: "v255", "a23");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { lds[threadIdx.x] += 1; }   // wall_clock64: 100 MHz constant counter
    if (lds[threadIdx.x] == 0xFFFFFFFFu) *sink = 1;
}

extern "C" int probe_spin(void* stream, int blocks, int lds_bytes, long long ticks_100mhz, void* sink, int fat)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (fat)
        hipLaunchKernelGGL(spin_kernel<true>, dim3(blocks), dim3(256), size_t(lds_bytes), static_cast<hipStream_t>(stream),
                           ticks_100mhz, static_cast<uint32_t*>(sink));
    else
        hipLaunchKernelGGL(spin_kernel<false>, dim3(blocks), dim3(256), size_t(lds_bytes), static_cast<hipStream_t>(stream),
                           ticks_100mhz, static_cast<uint32_t*>(sink));
    return int(hipGetLastError());
}

// stream whose kernels may use every CU except `reserve` of them (cleared bits spread evenly over the mask)
extern "C" int probe_masked_stream(void** out, int ncu_total, int reserve)
{
    std::vector<uint32_t> mask((ncu_total + 31) / 32, 0xFFFFFFFFu);
    if (ncu_total % 32) mask.back() = (1u << (ncu_total % 32)) - 1u;
    for (int k = 0; k < reserve; ++k) {
        const int bit = int((long long)k * ncu_total / reserve);
        mask[bit / 32] &= ~(1u << (bit % 32));
    }
    hipStream_t s = nullptr;
    hipError_t err = hipExtStreamCreateWithCUMask(&s, uint32_t(mask.size()), mask.data());
    *out = s;
    return int(err);
}
