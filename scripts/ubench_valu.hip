// ubench_valu.hip -- dev microbenchmark: VALU issue cost on gfx950 for the ops the traversal uses.
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value scripts/ubench_valu.hip -o scripts/_build/ubench_valu
// Prints ns (and cycles at the nominal 2.4 GHz) per wave-instruction per SIMD at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 64

// eight independent instructions on eight registers; X(d, a, b) expands to one instruction string
#define OP8(X)                                                                                                   \
    X("%0", "%0", "%8") X("%1", "%1", "%8") X("%2", "%2", "%8") X("%3", "%3", "%8")                              \
    X("%4", "%4", "%8") X("%5", "%5", "%8") X("%6", "%6", "%8") X("%7", "%7", "%8")

#define KERNEL(NAME, PRE, X)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)                               \
    {                                                                                                            \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,           \
              a6 = a0 + 6, a7 = a0 + 7;                                                                          \
        float c = 1.0001f;                                                                                       \
        for (int i = 0; i < iters; ++i) {                                                                        \
            asm volatile(PRE OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X)                             \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                         : "v"(c)                                                                                \
                         : "vcc", "s20", "s21");                                                                 \
        }                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                             \
    }

#define X_MUL(d, a, b) "v_mul_f32 " d ", " a ", " b "\n"
#define X_ADD(d, a, b) "v_add_f32 " d ", " a ", " b "\n"
#define X_MAX(d, a, b) "v_max_f32 " d ", " a ", " b "\n"
#define X_FMA(d, a, b) "v_fma_f32 " d ", " a ", " b ", " b "\n"
#define X_MAX3(d, a, b) "v_max3_f32 " d ", " a ", " b ", " b "\n"
#define X_CND_VCC(d, a, b) "v_cndmask_b32 " d ", " a ", " b ", vcc\n"
#define X_CND_SGPR(d, a, b) "v_cndmask_b32 " d ", " a ", " b ", s[20:21]\n"
#define X_CMP_VCC(d, a, b) "v_cmp_gt_f32 vcc, " a ", " b "\n"
#define X_CMP_SGPR(d, a, b) "v_cmp_gt_f32 s[20:21], " a ", " b "\n"
#define X_DPP(d, a, b) "v_mov_b32_dpp " d ", " a " quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define X_RCP(d, a, b) "v_rcp_f32 " d ", " a "\n"
#define X_MOV(d, a, b) "v_mov_b32 " d ", " b "\n"
#define X_AND(d, a, b) "v_and_b32 " d ", " a ", " b "\n"
#define X_LSHLADD(d, a, b) "v_lshl_add_u32 " d ", " a ", 2, " b "\n"
#define X_MULLEG(d, a, b) "v_mul_legacy_f32 " d ", " a ", " b "\n"

KERNEL(k_mul, "", X_MUL)
KERNEL(k_add, "", X_ADD)
KERNEL(k_max, "", X_MAX)
KERNEL(k_fma, "", X_FMA)
KERNEL(k_max3, "", X_MAX3)
KERNEL(k_cnd_vcc, "v_cmp_gt_f32 vcc, %8, %0\n", X_CND_VCC)
KERNEL(k_cnd_sgpr, "v_cmp_gt_f32 s[20:21], %8, %0\n", X_CND_SGPR)
KERNEL(k_cmp_vcc, "", X_CMP_VCC)
KERNEL(k_cmp_sgpr, "", X_CMP_SGPR)
KERNEL(k_dpp, "", X_DPP)
KERNEL(k_rcp, "", X_RCP)
KERNEL(k_mov, "", X_MOV)
KERNEL(k_and, "", X_AND)
KERNEL(k_lshladd, "", X_LSHLADD)

// same loop with only part of the wave enabled in EXEC: does the SIMD skip 16-lane groups that are all off?
#define KERNEL_MASKED(NAME, COND, X)                                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)                               \
    {                                                                                                            \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,           \
              a6 = a0 + 6, a7 = a0 + 7;                                                                          \
        float c = 1.0001f;                                                                                       \
        const unsigned l = threadIdx.x & 63u;                                                                    \
        if (COND) {                                                                                              \
            for (int i = 0; i < iters; ++i) {                                                                    \
                asm volatile(OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X) OP8(X)                             \
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                             : "v"(c)                                                                            \
                             : "vcc", "s20", "s21");                                                             \
            }                                                                                                    \
        }                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                             \
    }
KERNEL_MASKED(k_max_lo16, l < 16u, X_MAX)
KERNEL_MASKED(k_max_lo32, l < 32u, X_MAX)
KERNEL_MASKED(k_max_8scattered, (l & 7u) == 0u, X_MAX)
KERNEL_MASKED(k_mul_lo16, l < 16u, X_MUL)
KERNEL_MASKED(k_mul_lo32, l < 32u, X_MUL)
KERNEL_MASKED(k_rcp_lo16, l < 16u, X_RCP)

// scalar ALU: eight independent s_add_u32 / s_and_b64 chains
#define KERNEL_SALU(NAME, BODY, ...)                                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, int iters, float seed)                               \
    {                                                                                                            \
        for (int i = 0; i < iters; ++i) {                                                                        \
            asm volatile(BODY BODY BODY BODY BODY BODY BODY BODY ::: __VA_ARGS__);                                \
        }                                                                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = seed;                                                              \
    }
KERNEL_SALU(k_sadd, "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\ns_add_u32 s22, s22, 1\ns_add_u32 s23, s23, 1\n"
                    "s_add_u32 s24, s24, 1\ns_add_u32 s25, s25, 1\ns_add_u32 s26, s26, 1\ns_add_u32 s27, s27, 1\n",
            "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc")
KERNEL_SALU(k_sand64, "s_and_b64 s[20:21], s[20:21], exec\ns_and_b64 s[22:23], s[22:23], exec\ns_and_b64 s[24:25], s[24:25], exec\n"
                      "s_and_b64 s[26:27], s[26:27], exec\ns_and_b64 s[28:29], s[28:29], exec\ns_and_b64 s[30:31], s[30:31], exec\n"
                      "s_and_b64 s[32:33], s[32:33], exec\ns_and_b64 s[34:35], s[34:35], exec\n",
            "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "scc")
// a VALU stream and a SALU stream interleaved 1:1 in one wave: do they overlap across waves?
KERNEL_SALU(k_mix, "v_mul_f32 v10, v10, v11\ns_add_u32 s20, s20, 1\nv_mul_f32 v12, v12, v11\ns_add_u32 s21, s21, 1\n"
                   "v_mul_f32 v13, v13, v11\ns_add_u32 s22, s22, 1\nv_mul_f32 v14, v14, v11\ns_add_u32 s23, s23, 1\n",
            "s20", "s21", "s22", "s23", "scc", "v10", "v11", "v12", "v13", "v14")

typedef void (*kern_t)(float*, int, float);

void run(const char* name, kern_t kf, int cus)
{
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    const int iters = 20000;
    printf("%-22s", name);
    for (int bpc : {1, 2, 4, 8}) { // blocks of 4 waves per CU => waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kf, dim3(cus * bpc), dim3(256), 0, 0, out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kf, dim3(cus * bpc), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double instrs_per_simd = double(bpc) * iters * REP;
        double ns = ms * 1e6 / instrs_per_simd;
        printf("  w%d: %6.3f ns (%5.2f cyc)", bpc, ns, ns * 2.4);
    }
    printf("\n");
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    printf("CUs %d; per wave-instruction per SIMD, cycles at nominal 2.4 GHz\n", cus);
    run("v_mul_f32", k_mul, cus);
    run("v_add_f32", k_add, cus);
    run("v_max_f32", k_max, cus);
    run("v_fma_f32", k_fma, cus);
    run("v_max3_f32", k_max3, cus);
    run("v_cndmask vcc", k_cnd_vcc, cus);
    run("v_cndmask s[20:21]", k_cnd_sgpr, cus);
    run("v_cmp -> vcc", k_cmp_vcc, cus);
    run("v_cmp -> sgpr", k_cmp_sgpr, cus);
    run("v_mov_dpp quad_perm", k_dpp, cus);
    run("v_rcp_f32", k_rcp, cus);
    run("v_mov_b32", k_mov, cus);
    run("v_and_b32", k_and, cus);
    run("v_lshl_add_u32", k_lshladd, cus);
    run("s_add_u32", k_sadd, cus);
    run("s_and_b64", k_sand64, cus);
    run("v_mul + s_add pairs (per pair/2)", k_mix, cus);
    run("v_max_f32 lanes 0-15", k_max_lo16, cus);
    run("v_max_f32 lanes 0-31", k_max_lo32, cus);
    run("v_max_f32 8 scattered", k_max_8scattered, cus);
    run("v_mul_f32 lanes 0-15", k_mul_lo16, cus);
    run("v_mul_f32 lanes 0-31", k_mul_lo32, cus);
    run("v_rcp_f32 lanes 0-15", k_rcp_lo16, cus);
    return 0;
}
