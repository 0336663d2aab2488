// rcp_exhaustive.hip -- is a short Newton sequence on v_rcp_f32 equal to the IEEE quotient 1.0f / x for EVERY float?
// (dev tool: the triangle test needs inv_det = 1.0f / nDotDir correctly rounded; hipcc expands that into ~11 VALU)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-gpu-flush-denormals-to-zero scripts/rcp_exhaustive.hip -o /tmp/rcpx && /tmp/rcpx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ float rcp_hw(float x) { float r; asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }

template <int STEPS>
__device__ __forceinline__ float rcp_newton(float x)
{
    float r = rcp_hw(x);
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const float e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return r;
}

// mismatch counts per biased exponent of x (0..255), for 1 and 2 Newton steps
__global__ void check(unsigned long long* bad1, unsigned long long* bad2, unsigned long long* first_bad)
{
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = uint32_t(i);
        const float x = __uint_as_float(bits);
        const float ref = 1.0f / x;
        const uint32_t rb = __float_as_uint(ref);
        const bool ref_nan = ref != ref;
        const float a = rcp_newton<1>(x), b = rcp_newton<2>(x);
        const bool oka = ref_nan ? (a != a) : (__float_as_uint(a) == rb);
        const bool okb = ref_nan ? (b != b) : (__float_as_uint(b) == rb);
        const uint32_t ex = (bits >> 23) & 255u;
        if (!oka) atomicAdd(&bad1[ex], 1ull);
        if (!okb) { atomicAdd(&bad2[ex], 1ull); atomicMin(first_bad, (unsigned long long)bits); }
    }
}

int main()
{
    unsigned long long *d1, *d2, *df;
    hipMalloc(&d1, 256 * 8); hipMalloc(&d2, 256 * 8); hipMalloc(&df, 8);
    hipMemset(d1, 0, 256 * 8); hipMemset(d2, 0, 256 * 8); hipMemset(df, 0xFF, 8);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d1, d2, df);
    unsigned long long h1[256], h2[256], hf;
    hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost); hipMemcpy(h2, d2, sizeof(h2), hipMemcpyDeviceToHost); hipMemcpy(&hf, df, 8, hipMemcpyDeviceToHost);
    unsigned long long t1 = 0, t2 = 0;
    for (int e = 0; e < 256; ++e) { t1 += h1[e]; t2 += h2[e]; }
    printf("mismatches vs IEEE 1.0f/x over all 2^32 inputs: 1 Newton step %llu, 2 steps %llu (first bad bits %08llx)\n", t1, t2, hf);
    printf("by biased exponent of x (exp: 1-step / 2-step):");
    for (int e = 0; e < 256; ++e) if (h1[e] || h2[e]) printf(" %d:%llu/%llu", e, h1[e], h2[e]);
    printf("\n");
    return 0;
}
