// ubench_exec.hip -- dev microbenchmark: VALU issue cost on gfx950 as a function of the EXEC mask
// (does a sparse mask cost more or less than a full wave?).
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value scripts/ubench_exec.hip -o scripts/_build/ubench_exec
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define OP8(X)                                                                                                   \
    X("%0", "%0", "%8") X("%1", "%1", "%8") X("%2", "%2", "%8") X("%3", "%3", "%8")                              \
    X("%4", "%4", "%8") X("%5", "%5", "%8") X("%6", "%6", "%8") X("%7", "%7", "%8")

#define KERNEL_MASKED(NAME, X)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed, uint64_t mask)                \
    {                                                                                                            \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,           \
              a6 = a0 + 6, a7 = a0 + 7;                                                                          \
        float c = 1.0001f;                                                                                       \
        if ((mask >> (threadIdx.x & 63u)) & 1u) {                                                                \
            for (int i = 0; i < iters; ++i) {                                                                    \
                asm volatile(OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X)                             \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                             : "v"(c)                                                                            \
                             : "vcc", "s20", "s21");                                                             \
            }                                                                                                    \
        }                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                             \
    }

#define X_MUL(d, a, b) "v_mul_f32 " d ", " a ", " b "\n"
#define X_MAX(d, a, b) "v_max_f32 " d ", " a ", " b "\n"
#define X_FMA(d, a, b) "v_fma_f32 " d ", " a ", " b ", " b "\n"
#define X_CND_SGPR(d, a, b) "v_cndmask_b32 " d ", " a ", " b ", s[20:21]\n"
#define X_RCP(d, a, b) "v_rcp_f32 " d ", " a "\n"
#define X_SUB(d, a, b) "v_sub_f32 " d ", " a ", " b "\n"

KERNEL_MASKED(k_mul, X_MUL)
KERNEL_MASKED(k_max, X_MAX)
KERNEL_MASKED(k_fma, X_FMA)
KERNEL_MASKED(k_cnd, X_CND_SGPR)
KERNEL_MASKED(k_rcp, X_RCP)
KERNEL_MASKED(k_sub, X_SUB)

typedef void (*kern_t)(float*, int, float, uint64_t);

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, bpc = 4, iters = 20000;
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    struct { const char* name; uint64_t mask; } masks[] = {
        {"all 64", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 0-15", 0xFFFFull}, {"lanes 0-7", 0xFFull}, {"lane 0", 1ull},
        {"lane 37", 1ull << 37}, {"even lanes", 0x5555555555555555ull}, {"1 of 4", 0x1111111111111111ull},
        {"1 of 8", 0x0101010101010101ull}, {"1 of 16", 0x0001000100010001ull}, {"lanes 0,16,32,48 + 1", 0x0003000300030003ull},
        {"lanes 0-8 (9)", 0x1FFull}, {"lanes 0-9 (10)", 0x3FFull}, {"lanes 0-11 (12)", 0xFFFull}, {"lanes 0-7 + 63 (9)", 0x80000000000000FFull},
        {"scattered 9", 0x0101010101010103ull}, {"scattered 10", 0x0101010101010507ull & 0x0101010101010503ull | 0x0000000000100000ull},
        {"scattered 12", 0x0101110101110111ull}, {"2 per row of 16 (8)", 0x0011001100110011ull}, {"3 per row (12)", 0x0111011101110111ull},
        {"row 0 only: 9 lanes", 0x01FFull}, {"rows 0,1: 5+4", 0x000F001Full}, {"rows 0-3: 3+2+2+2", 0x0003000300030007ull},
        {"random 7", 0x0040100800220400ull}, {"random 32", 0xA3C5961E4B87D20Full}, {"all but lane 5", ~(1ull << 5)},
    };
    struct { const char* name; kern_t k; } kerns[] = {{"v_mul_f32", k_mul}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max},
                                                      {"v_fma_f32", k_fma}, {"v_cndmask sgpr", k_cnd}, {"v_rcp_f32", k_rcp}};
    printf("cycles (nominal 2.4 GHz) per wave-instruction per SIMD, 4 waves per SIMD\n%-24s", "EXEC mask");
    for (auto& k : kerns) printf("%16s", k.name);
    printf("\n");
    for (auto& m : masks) {
        printf("%-24s", m.name);
        for (auto& k : kerns) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k.k, dim3(cus * bpc), dim3(256), 0, 0, out, 100, 1.0f, m.mask);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.k, dim3(cus * bpc), dim3(256), 0, 0, out, iters, 1.0f, m.mask);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%16.2f", ms * 1e6 / (double(bpc) * iters * 64) * 2.4);
        }
        printf("\n");
    }
    return 0;
}
