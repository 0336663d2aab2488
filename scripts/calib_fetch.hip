// calib_fetch.hip -- calibrates rocprofv3 FETCH_SIZE for THIS kernel's access pattern: every quad fetches one
// random 64-B record with a 16-B-per-lane global_load_lds_dwordx4 (as trace_kernel's DMA fetch does) or each lane
// fetches its own record with four 16-B loads (the direct form).  The buffer is 2 GiB (far beyond the 256 MiB
// Infinity Cache) and every record is touched exactly once, so the true fetched bytes are known: records x 64 B.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/calib_fetch.hip -o scripts/_build/calib_fetch
// Run under: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- scripts/_build/calib_fetch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void* global_cptr;
typedef __attribute__((address_space(3))) void* lds_ptr;

// multiplicative hash permutation of [0, n) for n a power of two (odd multiplier => bijection)
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask) { return (i * 2654435761u) & mask; }

__global__ __launch_bounds__(256) void quad_dma(const char* recs, uint32_t mask, uint32_t per_quad, float* sink)
{
    __shared__ __attribute__((aligned(16))) char stage[4 * 1024];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gquad = (blockIdx.x * 256 + threadIdx.x) >> 2;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr)(stage + wave * 1024))));
    float acc = 0.f;
    for (uint32_t k = 0; k < per_quad; ++k) {
        const uint32_t r = perm(gquad * per_quad + k, mask);
        __builtin_amdgcn_global_load_lds((global_cptr)(recs + (size_t(r) << 6) + (lane & 3u) * 16u), (lds_ptr)(uintptr_t)lds_base, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += *reinterpret_cast<const float*>(stage + wave * 1024 + lane * 16);
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void lane_direct(const char* recs, uint32_t mask, uint32_t per_lane, float* sink)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t r = perm(g * per_lane + k, mask);
        const float4* p = reinterpret_cast<const float4*>(recs + (size_t(r) << 6));
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void stream_copy(const float4* src, float4* dst, size_t n)
{
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}

int main()
{
    const uint32_t nrec = 1u << 25;                 // 32 Mi records x 64 B = 2 GiB
    char* recs; float* sink; float4* dst;
    hipMalloc(&recs, size_t(nrec) * 64);
    hipMalloc(&sink, 256);
    hipMalloc(&dst, size_t(1) << 30);
    hipMemset(recs, 1, size_t(nrec) * 64);
    hipDeviceSynchronize();
    // quad form: 2^25 records, one per quad-iteration: grid x 64 quads x per_quad = 2^25
    hipLaunchKernelGGL(quad_dma, dim3(8192), dim3(256), 0, 0, recs, nrec - 1, 64u, sink);       // 8192*64*64 = 2^25
    hipDeviceSynchronize();
    // lane form: one record per lane-iteration: 8192*256*16 = 2^25
    hipLaunchKernelGGL(lane_direct, dim3(8192), dim3(256), 0, 0, recs, nrec - 1, 16u, sink);
    hipDeviceSynchronize();
    // reference: streaming copy of 1 GiB (known 1 GiB read + 1 GiB written)
    hipLaunchKernelGGL(stream_copy, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(recs), dst, (size_t(1) << 30) / 16);
    hipDeviceSynchronize();
    printf("true bytes: quad_dma %llu, lane_direct %llu, stream_copy read %llu\n", (unsigned long long)nrec * 64ull,
           (unsigned long long)nrec * 64ull, 1ull << 30);
    return 0;
}
