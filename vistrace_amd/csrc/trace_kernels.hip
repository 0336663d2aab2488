// trace_kernels.hip -- CDNA4 (gfx950) traversal kernels for AccelStruct::Traverse.
//
// One lane = one ray.  The per-ray algorithm is EXACTLY the single-ray walk the reference
// runs at source/objects/AccelStruct.cpp:818 (madmann91/bvh v1 SingleRayTraverser +
// FastNodeIntersector + Closest/AnyPrimitiveIntersector) with the
// in-tree triangle test of source/objects/Primitives.h:168-215, in the same visitation
// order (the tie-broken primitive index depends on it) and with the same fp32 rounding
// (no FMA contraction, IEEE divide).  What is MI355X-specific is everything around it:
//
//  * node pairs and triangles are 64-B records in ONE device array (pairs first, then the
//    leaf-ordered triangles), so a lane's next fetch is always "record r";
//  * the walk is a two-state machine per lane (NODE step / TRI test): leaves found by a node
//    step become a pending [tri_cur, tri_end) range (two leaf siblings are contiguous in the
//    triangle array) that is drained before the lane's next node step -- the order of slab
//    tests and triangle tests per ray is unchanged, but a wave never sits in one lane's leaf
//    loop, and the TRI branch only runs once enough lanes are waiting for it;
//  * DMA fetch (FETCH_DMA): the four lanes of a quad fetch each other's records together --
//    for k = 0..3 all four lanes load the four 16-B pieces of quad-lane k's record with one
//    `global_load_lds_dwordx4`, which lands as one contiguous 64-B line in a wave-private LDS
//    staging row; every lane then reads its own record back with four ds_read_b128.  The
//    vector L1 sees ONE 64-B request per record instead of four 16-B requests to four
//    different lines per lane (the direct form was L1-request bound: profiles/r1);
//  * the traversal stack lives in LDS as stack[entry][lane] (bank = lane, conflict-free),
//    `lds_entries` deep, with a per-lane global overflow area for deeper trees;
//  * persistent waves: a wave pulls blocks of consecutive rays from a global cursor and
//    re-fills idle lanes (ballot + mbcnt prefix) once enough of them have retired, so
//    divergent ray lengths do not leave lanes empty.  The cursor is one word (~90 M claims/s):
//    block w goes to wave w without an atomic, and for small scenes one atomic claims several
//    blocks while plenty are left (guided self-scheduling); optional per-XCD cursors;
//  * software CU reservation: blocks landing on a reserved CU beyond its quota exit at once,
//    leaving registers and LDS for the kernels of a concurrent collective (RCCL gather);
//  * ALPHA variants add the alpha test of Primitives.h:196-208 to the triangle step;
//  * coherence probe: a wave that starts from empty with all rays in one direction octant
//    (camera-like packets) fetches records directly and is only re-filled as a whole;
//  * PERSISTENT=false: one ray per lane, no cursor and no re-fill -- what the engine's auto mode
//    launches for small batches (no end-of-queue tail across the grid).
// Measured issue costs behind the instruction choices (selects with SGPR-pair masks, v_max3/
// v_min3, branch-free DMA): scripts/ubench_valu.hip, profiles/r1/notes.md.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstddef>
#include <cstdint>

#include "trace_kernels.h"

// ---- mutation testing of the parity suite (scripts/mutants.sh -> profiles/r6/mutants.txt) ---------------------------------
// `make variant NAME=mut_<k> HIPDEFS=-DVT_MUTANT=<k>` builds a deliberately WRONG traversal kernel: ONE deviation from the walk
// of SURVEY.md 3.2 / the triangle test of source/objects/Primitives.h:168-215, each of the kind a slip of the pen (or a wrong
// recall of bvh v1) would produce.  Every such library must turn at least one `-m gpu` test red; a mutant that survives is a
// missing test.  The product build never defines VT_MUTANT: VT_MUT(k, wrong, right) is then the token `right` after
// preprocessing, nothing of a mutant is compiled in, and the library does not export vt_mutant (tests/test_abi_symbols.py,
// check_isa.py).
//   1 near/far swap on fl >= fr            2 hit accepted on t < tmax            3 fp contraction on (fused multiply-add)
//   4 pending leaf range drained from the back (right leaf before left, descending slots)
//   5 plain 1/x instead of safe_inverse    6 slab entry without the tmin term    7 node accepted on first < second
//   8 hit needs u > 0                      9 back face culled on n.d >= 0       10 stack entries beyond the LDS part hold the near child
//  11 hit accepted on t > tmin            12 w = 1 - (u + v)                   13 FRONT faces culled (n.d < 0)
//  14 hit needs v > 0                     15 hit needs w > 0                   16 stack entry `lds_entries` still written to LDS
// Either side of the walk (the rows of SURVEY.md 8(f); killed by test_hit_attrs_vs_oracle, the refit / skin / bounce-loop tests):
//  21 GetPos with w = 1 - (u + v)          22 frontFacing on dot(wo, n) > 0      23 texUV weights u and v swapped
//  24 refit: e1 = p1 - p0                  25 refit: a leaf's box misses vertex 2  26 skinning: weight of bone 0 for every bone
//  27 CalcRayOrigin: |pos| <= 1/32         28 bounce direction: sin and cos of phi swapped
// The bounce loop's queue step (killed by the bounce-loop tests):
//  51 count: one entry behind a short queue is counted     52 scan: carry between 16-B groups drops a count
//  53 scan of the partials: thread d skips its add           54 emit: a hit's slot counts the hit itself
//  55 miss fill skipped when ONE path has died              56 emit: rows of paths that missed are not written
// The alpha test of the ALPHA variants (Primitives.h:196-208; killed by the alpha parity tests):
//  61 pass on alpha > ref instead of >=   62 negative texel index mirrored, not repeated   63 bilinear without the half-texel shift
//  64 right neighbour clamped at the border, not wrapped   65 untextured material passes whatever its ref   66 nearest: a / 256
//  71 merged launch: the first block of a set is looked up in the set before it     73 a NaN range costs no step (STATS counters)
// (31-32: scene_build.hip, 43-45: shading.hip, 57-58: batch.hip's range check)
// (9 is an EQUIVALENT mutant, kept as the record of why: with n.d == +-0 the test goes on to inv_det = +-inf, and then u, v are
//  NaN or infinite -- if both are +inf, w = 1 - u - v is -inf -- so the triangle is rejected either way: no input tells 9 apart.)
#ifdef VT_MUTANT
#define VT_MUT(k, wrong, right) ((VT_MUTANT == (k)) ? (wrong) : (right))
extern "C" __attribute__((visibility("default"))) int vt_mutant(void) { return VT_MUTANT; }
#else
#define VT_MUT(k, wrong, right) (right)
#endif     // (vt_internal.h has the same definition for the host walk; this file does not include it)

// Source anchors for scripts/isa_audit.py: where a body of the traversal loop starts.  The script compiles this file with line
// tables and attributes every generated instruction to the body its source line lies in.  Expands to nothing.
#define VT_ISA_MARK(name)

#if defined(VT_MUTANT) && VT_MUTANT == 3
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(off)
#endif

namespace vt {

namespace {

constexpr uint32_t kDone    = 0xFFFFFFFFu; // node cursor: nothing left to walk
constexpr uint32_t kNoFetch = 0xFFFFFFFFu; // record cursor: lane fetches nothing this round

__device__ __forceinline__ uint32_t lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t prefix_count(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// value of quad-lane K (0..3) broadcast to the four lanes of every quad (DPP quad_perm)
template <int K>
__device__ __forceinline__ uint32_t quad_broadcast(uint32_t v)
{
    return uint32_t(__builtin_amdgcn_mov_dpp(int(v), K * 0x55, 0xF, 0xF, true));
}

// bvh v1 safe_inverse
__device__ __forceinline__ float safe_inverse(float x)
{
    return VT_MUT(5, 1.0f / x, fabsf(x) <= FLT_EPSILON ? copysignf(1.0f / FLT_EPSILON, x) : 1.0f / x);
}

// bvh v1 robust_max(a,b) = a > b ? a : b and robust_min(a,b) = a < b ? a : b, nested as
//   first  = rmax(e0, rmax(e1, rmax(e2, tmin)))   second = rmin(x0, rmin(x1, rmin(x2, tmax))).
// With tmin/tmax not NaN (guarded at ray start) every nested call has a non-NaN second
// operand, for which rmax(a,b) == maxnum(a,b) and rmin(a,b) == minnum(a,b) up to the sign of
// a zero result (a NaN first operand yields b in both).  Zero signs never change the
// comparisons first <= second / first_l > first_r, so v_max3/v_min3 give the same decisions.
// v_max3_f32 / v_min3_f32 written out so hipcc does not put a canonicalising v_max in front of
// every use of tmin/tmax: all NaNs that can reach these (inf - inf, 0 * inf) are quiet, and
// for quiet NaNs the instruction returns the non-NaN operand(s), like maxnum/minnum.
__device__ __forceinline__ float slab_first(float e0, float e1, float e2, float tmin)
{
    float m, r;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(e2), "v"(VT_MUT(6, e2, tmin)));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(e0), "v"(e1), "v"(m));
    return r;
}
__device__ __forceinline__ float slab_second(float x0, float x1, float x2, float tmax)
{
    float m, r;
    asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(x2), "v"(tmax));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x0), "v"(x1), "v"(m));
    return r;
}

// lane-wise select with the mask in an SGPR pair: mask bit set -> if_set, else if_clear
__device__ __forceinline__ float sel(uint64_t mask, float if_set, float if_clear)
{
    float r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

struct Lane {
    // ray (bvh::Ray, Primitives.h:11-33); tmax shrinks as hits are found
    float ox, oy, oz, dx, dy, dz, tmin, tmax;
    // FastNodeIntersector state
    float ix, iy, iz, sx, sy, sz;
    // best hit
    uint32_t prim;
    float u, v;
    // cursors
    uint32_t node;             // pair index to visit next, or kDone
    uint32_t tri_cur, tri_end; // pending leaf triangles
    uint32_t sp;               // stack entries in use
    uint32_t steps, tests;
    // ALPHA variants: a parked candidate hit (Primitives.h:196-208 decides whether it counts)
    uint32_t astate;           // 0 none, 1 = wants its triangle's AlphaRec, 2 = texels requested
    uint32_t cprim;  float cu, cv, ct;
    float ax, ay, aref;        // state 2: bilinear weights (ax < 0: nearest, one texel) and the material's reference
};

// ALPHA: the texel loads of state 1 land one round later in v76..v79 -- registers the compiler does not allocate (the kernel
// asks for at most kCompilerVgprs through amdgpu_num_vgpr; 76 + these 4 = 80 = six waves per SIMD).
// Nothing the register allocator does (copies, re-use, spills) can then touch a load in flight; state 2 moves the values
// out behind its own wait.  tests/test_kernel_asm.py reads the generated code of every ALPHA variant and fails if the
// compiler ever mentions one of the four (the attribute is a request: a variant that needs more registers gets them).
#define VT_TEXEL_REGS "v76", "v77", "v78", "v79"
constexpr int kCompilerVgprs = 76;

typedef __attribute__((address_space(1))) const void* global_cptr;
// batch descriptors are read through the constant address space: wave-uniform scalar loads, nothing of a batch occupies
// registers while the wave walks (the kernels sit at the SGPR limit)
typedef __attribute__((address_space(4))) const TraceSeg* seg_cptr;
typedef __attribute__((address_space(3))) void*       lds_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint32_t kStageRow   = 1024 + 16;      // one staging row: 16 quads x 64 B, +16 B so that
constexpr uint32_t kStageBytes = 4 * kStageRow;  // the four rows start on different banks

} // namespace

// ---- alpha test inside intersect(): Primitives.h:196-208, TransformTexcoord Utils.h:65-72 -------------------
// The texel lookup (IVTFTexture::Sample lives in the absent VTFParser submodule) is the one defined in
// include/vistrace_hip.h at vt_alpha_material: mip 0 alpha plane, repeat addressing, nearest or bilinear.
// In the ALPHA kernel variants the test is two more states of a lane instead of a call inside the triangle test: a
// candidate hit on a flagged triangle is parked, the lane's next record is the triangle's 64-B AlphaRec (uvs + the
// material's transform, through the same fetch as every other record), the texels are requested when it arrives and
// judged one round later -- the wave never waits for one lane's dependent loads (profiles/r3/notes.md).
__device__ __forceinline__ uint32_t wrap_index(float f, uint32_t n)
{
    const int i = int(f);                        // f is integral and |f| < 1e9: fits; n <= 65535
    const int m = i % int(n);
    return uint32_t(m < 0 ? VT_MUT(62, -m, m + int(n)) : m);
}

// alpha of the material's plane from the texel(s) read (a / 255; bilinear with texel centres at (i + 0.5) / W): ax < 0 = nearest
__device__ __forceinline__ float alpha_from_texels(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, float ax, float ay)
{
    if (ax < 0.0f) return float(t0) / VT_MUT(66, 256.0f, 255.0f);
    const float a00 = float(t0), a10 = float(t1), a01 = float(t2), a11 = float(t3);
    const float top = a00 * (1.0f - ax) + a10 * ax;
    const float bot = a01 * (1.0f - ax) + a11 * ax;
    return (top * (1.0f - ay) + bot * ay) / 255.0f;
}

// one thread per triangle slot: the AlphaRec of the triangle in that slot from the caller's side tables
__global__ __launch_bounds__(kBlockThreads) void alpha_records_kernel(AlphaRecArgs a)
{
    const uint32_t slot = blockIdx.x * kBlockThreads + threadIdx.x;
    if (slot >= a.n) return;
    const uint32_t prim = a.tris[slot].prim;
    const vt_tri_attribs A = a.attribs[prim];
    AlphaRec r;
    for (int k = 0; k < 3; ++k) { r.uv[k][0] = A.uv[k][0]; r.uv[k][1] = A.uv[k][1]; }
    if (A.material >= a.n_mats) {                // no such material: the reference's test is skipped, the hit stands
        for (int q = 0; q < 2; ++q) r.m[q][0] = r.m[q][1] = r.m[q][2] = 0.0f;
        r.tex_scale = 0.0f; r.alpha_ref = 0.0f; r.dims = kAlphaAlwaysPass; r.offset_filter = 0;
    } else {
        const vt_alpha_material M = a.mats[A.material];
        for (int q = 0; q < 2; ++q) {
            r.m[q][0] = M.tex_mat[q][0]; r.m[q][1] = M.tex_mat[q][1];
            r.m[q][2] = M.tex_mat[q][2] + M.tex_mat[q][3];           // the sum TransformTexcoord forms first (Utils.h:65-72)
        }
        r.tex_scale = M.tex_scale; r.alpha_ref = M.alpha_ref;
        r.dims = M.width | (M.height << 16);
        r.offset_filter = uint32_t(M.offset) | (M.filter << 31);
    }
    a.out[slot] = r;
}

// The ray-block cursors of a launch slot are zero when a launch starts: the last wave of the persistent grid to leave puts them
// back (all others have made their last claim by then), so that no fill kernel has to run between two launches -- on one stream
// that was fill, wait, launch: 20 us between two traces instead of 8 (scripts/launch_gaps.py).
__device__ __forceinline__ void leave_grid(const TraceArgs& a, uint32_t waves_leaving)
{
    const uint32_t waves = gridDim.x * (kBlockThreads / 64);
    if (atomicAdd(a.block_cursor + kExitWord, waves_leaving) + waves_leaving == waves) {
        for (uint32_t x = 0; x < 8u; ++x) __hip_atomic_store(a.block_cursor + 16u * x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.block_cursor + kExitWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// DEVN (vt_bounce_loop_dev, depth >= 1): the launch was sized on the host for an upper bound of its ONE batch; how many rays the
// batch really holds is a device word (the live-path count the previous depth's queue step left there), read once per wave.
template <bool ANY_HIT, bool STATS, bool PERSISTENT, bool FETCH_DMA, bool ALPHA, bool DEVN = false>
__device__ __forceinline__ void trace_body(const TraceArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_dyn[];
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* const st_lds = lds_dyn + size_t(wave) * a.lds_entries * 64 + lane; // entry k at [k*64]
    const uint32_t gthread = blockIdx.x * kBlockThreads + threadIdx.x;
    const uint32_t gstride = gridDim.x * kBlockThreads;
    uint32_t* const st_ovf = a.overflow + gthread;                               // entry k at [k*gstride]
    // wave-private staging rows behind the stacks of the block's four waves (DMA fetch only)
    char* const stage = reinterpret_cast<char*>(lds_dyn) + size_t(kBlockThreads / 64) * a.lds_entries * 256 +
                        size_t(wave) * kStageBytes;

    const char* const records = reinterpret_cast<const char*>(a.records);
    // LDS byte address of the wave's staging rows, made provably wave-uniform (it becomes M0 of the
    // DMA loads), and of this lane's 64-B slot: row (lane & 3), quad (lane >> 2)
    const uint32_t stage_lds = __builtin_amdgcn_readfirstlane(uint32_t(uintptr_t((lds_ptr)stage)));
    const uint32_t my_rec = stage_lds + (lane & 3u) * kStageRow + (lane >> 2) * 64u;

    if constexpr (PERSISTENT) {
        // Software CU reservation (engine option "reserved_cus"): a reserved CU keeps only `reserved_limit` blocks
        // of this grid, later arrivals leave at once, so that registers and LDS stay free there for the kernels
        // of a concurrent collective -- a resident persistent grid otherwise fills every CU until its last ray
        // and such kernels only start when the trace ends (scripts/overlap_probe.py, profiles/r1/notes.md).
        if (a.reserved_cus != nullptr) {
            const uint32_t id = __smid() & 1023u;        // xcc[9:6] se[5:4] cu[3:0]: the same for the whole block
            if ((a.reserved_cus[id >> 5] >> (id & 31u)) & 1u) {
                // one LDS word that nothing else uses: the pad behind the last staging row (DMA variants) or
                // the word behind the stacks (see trace_lds_bytes)
                uint32_t* const flag = lds_dyn + size_t(kBlockThreads / 64) * a.lds_entries * 64 +
                                       (FETCH_DMA ? size_t(kBlockThreads / 64) * kStageBytes / 4 - 1 : 0);
                if (threadIdx.x == 0) *flag = atomicAdd(&a.cu_slots[id], 1u) >= a.reserved_limit ? 1u : 0u;
                __syncthreads();
                if (*flag != 0) {
                    if (threadIdx.x == 0) leave_grid(a, kBlockThreads / 64);
                    return;
                }
            }
        }
    }

    Lane L;
    if constexpr (ALPHA) { L.astate = 0; L.ax = L.ay = L.aref = 0.f; L.cprim = 0; L.cu = L.cv = L.ct = 0.f; }
    uint64_t ray_idx = 0;
    bool has_ray = false;
    // rays and ray blocks of the launch: kernel arguments, or (DEVN) derived from the device word
    uint32_t dev_n = 0, dev_nblocks = 0;
    if constexpr (DEVN) {
        dev_n = __builtin_amdgcn_readfirstlane(*a.live_n);
        const uint32_t unit = PERSISTENT ? a.block_rays : kBlockThreads;
        dev_nblocks = dev_n / unit + (dev_n % unit != 0 ? 1u : 0u);
    }
    auto launch_nblocks = [&]() -> uint32_t { if constexpr (DEVN) return dev_nblocks; else return a.nblocks; };

    // wave-uniform block cursor (PERSISTENT)
    uint64_t blk_cur = 0, blk_end = 0;
    bool blk_tiled = false;             // wave-uniform: the current block is taken tile-wise
    uint32_t blk_tile_w = 0;            // ... in rows of this many rays
    uint32_t blk_first = 0;             // its first (tile-order) index
    uint64_t blk_base = 0;              // ray index of its top left pixel
    bool exhausted = false;
    bool coherent = false;   // wave-uniform: this wave's rays share a direction octant
    // wave-uniform: the batch the current ray block belongs to (see TraceSeg).  Only the pointer lives in registers; rays,
    // result offset, size and tiling are scalar loads when a block is acquired / a lane is re-filled.
    seg_cptr cur = (seg_cptr)((__attribute__((address_space(4))) const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TraceArgs, seg0));
    // wave-uniform: the first block of rays is assigned statically (not with reserved CUs: a block that
    // leaves must not take rays with it, so the cursor hands out everything)
    bool first_block = a.reserved_cus == nullptr && a.xcd_cursors == 0;
    // per-XCD cursors: XCD x hands out the x-th eighth of the ray blocks, so that neighbouring rays meet in one
    // L2; a wave whose XCD has run dry steals from the next ones
    const uint32_t my_xcd = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u;   // XCC_ID[3:0]
    uint32_t xcd_off = 0;    // wave-uniform: XCDs (starting at its own) this wave has found exhausted
    // guided self-scheduling on the global cursor: one atomic claims several consecutive ray blocks while plenty
    // are left, single blocks near the end.  Every wave of the grid adds to ONE word; that word hands out at most
    // ~90 M claims per second (11 ns each, measured), i.e. 5.8 Grays/s with 64-ray claims.
    uint32_t claim_cur = 0, claim_end = 0;   // wave-uniform: ray blocks claimed but not yet started

    // the batch that holds ray block b (wave-uniform) becomes the current one; returns b's index within it.  Batches are
    // listed in block order; an empty batch shares its first block with its successor, which the search then prefers.
    auto enter_batch = [&](uint32_t b) -> uint32_t {
        uint32_t first = cur->first_block;
        if (b < first || b >= cur->end_block) {        // never on a plain launch: its one batch holds every block
            const seg_cptr table = (seg_cptr)(uintptr_t)a.segs;
            uint32_t lo = 0, hi = a.nseg;              // table[lo].first_block <= b < table[hi].first_block
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (VT_MUT(71, table[mid].first_block < b, table[mid].first_block <= b)) lo = mid; else hi = mid;
            }
            cur = table + lo;
            first = cur->first_block;
        }
        return b - first;
    };

    auto start_ray = [&](uint64_t idx) {
        // rays are read once and hits written once: non-temporal, so that 48 B per ray of streaming data do not push
        // records out of L2 / Infinity Cache (S10M, whose records do not fit there: 5.97 -> 5.86 ms; S1M unchanged)
        const f32x4* r4 = reinterpret_cast<const f32x4*>(cur->rays + idx);
        const f32x4 r0 = __builtin_nontemporal_load(r4), r1 = __builtin_nontemporal_load(r4 + 1);
        L.ox = r0.x; L.oy = r0.y; L.oz = r0.z; L.dx = r0.w;
        L.dy = r1.x; L.dz = r1.y; L.tmin = r1.z; L.tmax = r1.w;
        L.ix = safe_inverse(L.dx); L.iy = safe_inverse(L.dy); L.iz = safe_inverse(L.dz);
        L.sx = -L.ox * L.ix; L.sy = -L.oy * L.iy; L.sz = -L.oz * L.iz;
        L.prim = VT_MISS; L.u = 0.f; L.v = 0.f;
        L.sp = 0; L.steps = 0; L.tests = 0;
        if constexpr (ALPHA) L.astate = 0;
        if (a.root_leaf_count != 0) {        // the root is a leaf: no slab test at all
            L.node = kDone; L.tri_cur = 0; L.tri_end = a.root_leaf_count;
        } else {
            L.node = a.npairs != 0 ? 0u : kDone; L.tri_cur = 0; L.tri_end = 0;
        }
        // A NaN or infinite origin / direction component can never produce a hit: the triangle test needs
        // u, v, w >= 0 and t in range, and with a non-finite component n.d or cross(d, p0 - o) is NaN or
        // infinite in a way that makes u or v NaN, or u, v infinite with w = -inf (Primitives.h:173-189).
        // The reference nevertheless WALKS for such a ray -- NaN slab terms drop out of robust_max/min, so
        // with all three axes poisoned every box passes and one ray visits the whole tree (555 k steps on
        // S1M; 8 Mi such rays kept this kernel busy for 54 s).  The result is known, so the walk is skipped;
        // only the STATS variant, which must also report the reference's step and test counts, still walks.
        if constexpr (!STATS) {
            const bool finite = fabsf(L.ox) <= FLT_MAX && fabsf(L.oy) <= FLT_MAX && fabsf(L.oz) <= FLT_MAX &&
                                fabsf(L.dx) <= FLT_MAX && fabsf(L.dy) <= FLT_MAX && fabsf(L.dz) <= FLT_MAX;
            if (!finite) { L.node = kDone; L.tri_cur = 0; L.tri_end = 0; }
        }
        // A NaN tmin or tmax makes every slab test and every triangle range test of the
        // reference false: the ray misses after one step (or after testing a leaf root).
        if (L.tmin != L.tmin || L.tmax != L.tmax) {
            L.steps = (a.root_leaf_count == 0 && a.npairs != 0) ? VT_MUT(73, 0u, 1u) : 0u;
            L.tests = a.root_leaf_count;
            L.node = kDone; L.tri_cur = 0; L.tri_end = 0;
        }
        ray_idx = uint64_t(cur->out_off) + idx;    // where the result goes: all that a lane keeps of its batch
        has_ray = true;
    };

    auto finish_ray = [&]() {
        if constexpr (ANY_HIT) {
            a.occluded[ray_idx] = L.prim != VT_MISS ? 1 : 0;
        } else {
            f32x4 h;
            h.x = __uint_as_float(L.prim);
            h.y = L.prim != VT_MISS ? L.tmax : 0.f;
            h.z = L.u; h.w = L.v;
            __builtin_nontemporal_store(h, reinterpret_cast<f32x4*>(a.hits) + ray_idx);
        }
        if constexpr (STATS) {
            a.ray_stats[ray_idx] = vt_ray_stats{L.steps, L.tests};
        }
        has_ray = false;
    };

    if constexpr (!PERSISTENT) {
        uint64_t idx = uint64_t(enter_batch(blockIdx.x)) * kBlockThreads + threadIdx.x;
        const uint32_t tile_w = cur->tile_w;
        // (the ALPHA variants too: the mapping is worked out once, in front of the loop, and costs the loop no register)
        if (tile_w != 0 && idx < cur->tiled_rays) { // this wave's 64 rays are one 4 x 16 pixel tile (see TraceSeg::tile_w)
            const uint32_t j = uint32_t(idx), t = j >> 6, k = j & 63u, tpr = tile_w >> 2;
            const uint32_t ty = t / tpr, tx = t - ty * tpr;
            idx = uint64_t(ty * 16u + (k >> 2)) * tile_w + tx * 4u + (k & 3u);
        }
        if (DEVN ? idx < dev_n : idx < cur->n) start_ray(idx);
        if (has_ray && L.node == kDone && L.tri_cur >= L.tri_end) finish_ray();
    }

    for (;;) {
        // Wave priority: high from here until this iteration's record fetch has been issued, low while the wave waits for
        // the records and computes on them.  Waves that are about to put loads in flight are then picked ahead of waves
        // that are computing, so the fetches of a SIMD's waves overlap better: 16 Mi bounce rays 4.52 -> 4.33 ms, camera
        // rays 2.14 -> 2.08 ms, S10M 5.98 -> 5.87 ms (profiles/r2/notes.md; the opposite assignment is 1 % slower
        // than none).  Scheduling only: results are unaffected.
        __builtin_amdgcn_s_setprio(3);
        VT_ISA_MARK("refill");
        if constexpr (PERSISTENT) {
            const uint64_t idle = __ballot(!has_ray);
            if (idle != 0 && !exhausted) {
                const uint32_t nidle = __popcll(idle);
                // a coherent wave (see below) is only re-filled as a whole, so it stays coherent
                if ((nidle >= a.refill_threshold && !coherent) || idle == ~0ull) {
                    if (blk_cur == blk_end) { // acquire the next block of consecutive rays
                        uint32_t b = 0;
                        if (first_block) {            // block w goes to wave w without touching the cursor,
                            b = blockIdx.x * (kBlockThreads / 64) + wave;   // which hands out the blocks from cursor_base = #waves on
                            first_block = false;
                        } else if (a.xcd_cursors != 0) {
                            b = 0xFFFFFFFFu;
                            while (xcd_off < 8u) {
                                const uint32_t x = (my_xcd + xcd_off) & 7u;
                                const uint32_t lo = uint32_t((uint64_t(x) * launch_nblocks()) >> 3);
                                const uint32_t hi = uint32_t((uint64_t(x + 1) * launch_nblocks()) >> 3);
                                uint32_t t = 0xFFFFFFFFu;
                                if (lane == 0 && lo < hi) t = atomicAdd(a.block_cursor + 16u * x, 1u);
                                t = __builtin_amdgcn_readfirstlane(t);
                                if (lo < hi && t < hi - lo) { b = lo + t; break; }
                                ++xcd_off;
                            }
                            if (b == 0xFFFFFFFFu) b = launch_nblocks();          // everything handed out
                        } else {
                            if (claim_cur == claim_end) {
                                const uint32_t left = launch_nblocks() > claim_end ? launch_nblocks() - claim_end : 0u;   // stale, fine
                                const uint32_t waves = gridDim.x * (kBlockThreads / 64);
                                uint32_t k = left / (2u * waves);
                                k = k < 1u ? 1u : (k > a.max_claim ? a.max_claim : k);
                                uint32_t c = 0;
                                if (lane == 0) c = atomicAdd(a.block_cursor, k);
                                claim_cur = a.cursor_base + __builtin_amdgcn_readfirstlane(c);
                                claim_end = claim_cur + k;
                            }
                            b = claim_cur++;
                        }
                        b = __builtin_amdgcn_readfirstlane(b);
                        if (b >= launch_nblocks()) {
                            exhausted = true; blk_cur = blk_end = 0;
                        } else {
                            b = enter_batch(b);                  // now the block's index within its batch
                            const uint64_t n = DEVN ? uint64_t(dev_n) : cur->n;
                            blk_cur = uint64_t(b) * a.block_rays;
                            blk_end = blk_cur + a.block_rays < n ? blk_cur + a.block_rays : n;
                            // image-order batch: the block is one or two 4 x 16 pixel tiles side by side (see TraceSeg::tile_w)
                            // (not in the persistent ALPHA variants: their register budget has no room for the extra wave state; the
                            // one-ray-per-lane ALPHA variants do tile, and the engine picks them for hinted camera batches)
                            blk_tile_w = cur->tile_w;
                            blk_tiled = !ALPHA && blk_tile_w != 0 && blk_cur + a.block_rays <= cur->tiled_rays;
                            if (blk_tiled) {
                                const uint32_t t = uint32_t(blk_cur >> 6), tpr = blk_tile_w >> 2;
                                const uint32_t ty = t / tpr, tx = t - ty * tpr;
                                blk_first = uint32_t(blk_cur);
                                blk_base = uint64_t(ty) * 16u * blk_tile_w + tx * 4u;
                            }
                        }
                    }
                    if (!exhausted) {
                        const uint64_t avail = blk_end - blk_cur;
                        const uint32_t mine = prefix_count(idle);
                        if (!has_ray && mine < avail) {
                            uint64_t idx = blk_cur + mine;
                            if (blk_tiled) {
                                const uint32_t k = uint32_t(idx) - blk_first;
                                idx = blk_base + ((k >> 6) << 2) + (k & 3u) + uint64_t((k >> 2) & 15u) * blk_tile_w;
                            }
                            start_ray(idx);
                            if (L.node == kDone && L.tri_cur >= L.tri_end) finish_ray(); // empty scene / NaN range
                        }
                        blk_cur += nidle < avail ? nidle : avail;
                        if constexpr (FETCH_DMA) {
                            // Coherence probe when a wave starts from empty: if all of its rays share one
                            // direction octant (camera-like packets) the lanes walk the tree together --
                            // fetch records directly (neighbouring lanes hit the same L1 lines) and never
                            // mix new rays into the wave until it has drained.
                            if (idle == ~0ull) {
                                const uint64_t act = __ballot(has_ray);
                                const uint64_t ax = __ballot(has_ray && (__float_as_uint(L.dx) >> 31)),
                                               ay = __ballot(has_ray && (__float_as_uint(L.dy) >> 31)),
                                               az = __ballot(has_ray && (__float_as_uint(L.dz) >> 31));
                                // ... and point the same way: within ~18 degrees of the wave's first ray.  (Sharing
                                // an octant alone is not coherence: a batch of bounce rays sorted by octant would
                                // otherwise run in this mode, 8.2 instead of 5.3 ms.)
                                const int first = act != 0 ? __builtin_ctzll(act) : 0;
                                const float fx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(L.dx), first)),
                                            fy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(L.dy), first)),
                                            fz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(L.dz), first));
                                const float dd = L.dx * fx + L.dy * fy + L.dz * fz;
                                const float l2 = (L.dx * L.dx + L.dy * L.dy + L.dz * L.dz) * (fx * fx + fy * fy + fz * fz);
                                // ... and start close together (camera rays share their origin; one direction from
                                // origins scattered over the scene is not a packet: 7.5 instead of 4.4 ms)
                                const float gx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(L.ox), first)),
                                            gy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(L.oy), first)),
                                            gz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(L.oz), first));
                                const float ex = L.ox - gx, ey = L.oy - gy, ez = L.oz - gz;
                                const bool near = ex * ex + ey * ey + ez * ez <= a.coherent_radius2;
                                const uint64_t narrow = __ballot(!has_ray || (near && dd > 0.0f && dd * dd >= 0.9f * l2));
                                coherent = a.coherent_detect != 0 && (ax == 0 || ax == act) && (ay == 0 || ay == act) &&
                                           (az == 0 || az == act) && narrow == ~0ull;
                            } else {
                                coherent = false;
                            }
                        }
                    }
                }
            }
            if (exhausted && __ballot(has_ray) == 0) break;
        } else {
            if (__ballot(has_ray) == 0) break;
        }

        // ---- which record does this lane need? ----------------------------------------------
        VT_ISA_MARK("select");
        bool alpha1 = false, alpha2 = false;                // ALPHA: the lane's parked candidate waits for its AlphaRec / its texels
        if constexpr (ALPHA) { alpha1 = has_ray && L.astate == 1; alpha2 = has_ray && L.astate == 2; }
        const bool want_tri  = has_ray && !alpha1 && !alpha2 && L.tri_cur < L.tri_end;
        const bool want_node = has_ray && !alpha1 && !alpha2 && !want_tri;       // node != kDone is implied (else finished)
        // the TRI branch runs only when enough lanes wait for it, or nobody can step a node
        const uint64_t tri_mask = __ballot(want_tri);
        const bool run_tri = tri_mask != 0 && (uint32_t(__popcll(tri_mask)) >= a.tri_threshold || __ballot(want_node) == 0);
        const bool do_tri = want_tri && run_tri;
        // idle lanes (no ray, or waiting for the TRI branch) fetch record 0 in the DMA form: an always-valid
        // address keeps the four DMA loads branch-free; the direct form skips them instead.  (11 % of the L1 accesses of the
        // headline launch are such dummies.  Neither way of removing them pays: EXEC-masked rows +4 % (round 3); handing an idle
        // lane the record of its neighbour quad's lane so that the row names 15 lines instead of 16 does not lower the access
        // count at all -- every quad of a DMA instruction is its own L1 access -- and costs +2 % (round 5, profiles/r5/notes.md).)
        // (ALPHA: the AlphaRec of the triangle just tested sits at alpha_base + its slot; tri_cur is already past it)
        const uint32_t rec = do_tri ? a.tri_base + VT_MUT(4, L.tri_end - 1u, L.tri_cur)
                                    : (want_node ? L.node : (alpha1 ? a.alpha_base + VT_MUT(4, L.tri_end, L.tri_cur - 1u) : (FETCH_DMA ? 0u : kNoFetch)));
        // ALPHA: the block that turns an AlphaRec into texel addresses is ~60 instructions for the WHOLE wave whenever one lane
        // needs it; like the TRI branch it waits until `alpha_threshold` lanes have a candidate parked, or nobody else can make
        // progress.  A waiting lane asks for its AlphaRec again next iteration (an L1 hit).  Scheduling only: the order of a ray's
        // own steps and tests is untouched.
        bool do_alpha1 = alpha1;
        if constexpr (ALPHA) {
            const uint64_t a1_mask = __ballot(alpha1);
            const bool run_a1 = a1_mask != 0 && (uint32_t(__popcll(a1_mask)) >= a.alpha_threshold || __ballot(want_node || do_tri || alpha2) == 0);
            do_alpha1 = alpha1 && run_a1;
        }

        VT_ISA_MARK("fetch");
        float4 q0, q1, q2, q3;   // the record
        bool fetched = false;
        if constexpr (FETCH_DMA) {
            if (coherent) {      // wave-uniform
                if (do_tri || want_node || alpha1) {
                    const float4* g = reinterpret_cast<const float4*>(records + (size_t(rec) << 6));
                    q0 = g[0]; q1 = g[1]; q2 = g[2]; q3 = g[3];
                }
                __builtin_amdgcn_s_setprio(0);
                fetched = true;
            }
        }
        if (FETCH_DMA && !fetched) {
            // quad-cooperative fetch through LDS: row k receives the records of every quad's lane k.
            // Addresses are base + 32-bit byte offset (the engine uses this kernel below 4 GiB).
            const uint32_t piece = (lane & 3u) * 16u;
            const uint32_t r0 = quad_broadcast<0>(rec), r1 = quad_broadcast<1>(rec),
                           r2 = quad_broadcast<2>(rec), r3 = quad_broadcast<3>(rec);
            __builtin_amdgcn_global_load_lds((global_cptr)(records + ((r0 << 6) | piece)),
                                             (lds_ptr)(uintptr_t)(stage_lds + 0u * kStageRow), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((global_cptr)(records + ((r1 << 6) | piece)),
                                             (lds_ptr)(uintptr_t)(stage_lds + 1u * kStageRow), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((global_cptr)(records + ((r2 << 6) | piece)),
                                             (lds_ptr)(uintptr_t)(stage_lds + 2u * kStageRow), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((global_cptr)(records + ((r3 << 6) | piece)),
                                             (lds_ptr)(uintptr_t)(stage_lds + 3u * kStageRow), 16, 0, 0);
            // Wait for the DMA rows, then read this lane's 64-B record back with four ds_read_b128
            // (conflict-free with the padded rows).  One asm statement holds the reads and their
            // waits, so hipcc can neither split the reads nor consume a destination early
            // (cdna_hip_programming.md 5.7 item 1).  Octant-ordered ds_read2_b32 reads would save
            // the twelve selects of a NODE step but are 4-way bank conflicted (64-B slot stride,
            // 32 banks for 4-byte reads) and made the LDS the bottleneck: profiles/r1/notes.md.
            f32x4 v0, v1, v2, v3;
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(0)\n\t"
                         "ds_read_b128 %0, %4\n\t"
                         "ds_read_b128 %1, %4 offset:16\n\t"
                         "ds_read_b128 %2, %4 offset:32\n\t"
                         "ds_read_b128 %3, %4 offset:48\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                         : "v"(my_rec)
                         : "memory");
            q0 = make_float4(v0.x, v0.y, v0.z, v0.w); q1 = make_float4(v1.x, v1.y, v1.z, v1.w);
            q2 = make_float4(v2.x, v2.y, v2.z, v2.w); q3 = make_float4(v3.x, v3.y, v3.z, v3.w);
        } else if (!fetched) {
            if (rec != kNoFetch) {
                const float4* g = reinterpret_cast<const float4*>(records + size_t(rec) * 64);
                q0 = g[0]; q1 = g[1]; q2 = g[2]; q3 = g[3];
            }
            __builtin_amdgcn_s_setprio(0);
        }

        VT_ISA_MARK("tri");
        if (do_tri) {
            // ---- TRI: TriangleBackfaceCull::intersect, Primitives.h:168-215 --------------
            VT_MUT(4, --L.tri_end, ++L.tri_cur);
            if constexpr (STATS) ++L.tests;
            const float p0x = q0.x, p0y = q0.y, p0z = q0.z;
            const float e1x = q0.w, e1y = q1.x, e1z = q1.y;
            const float e2x = q1.z, e2y = q1.w, e2z = q2.x;
            const float nx = q2.y, ny = q2.z, nz = q2.w;
            const uint32_t tprim = __float_as_uint(q3.x), tflags = __float_as_uint(q3.y);

            const float nDotDir = (nx * L.dx + ny * L.dy) + nz * L.dz;                 // :173
            const bool culled = (tflags & VT_TRI_CULL_BACKFACE) && VT_MUT(13, nDotDir < 0.0f, VT_MUT(9, nDotDir >= 0.0f, nDotDir > 0.0f));   // :174
            const float cx = p0x - L.ox, cy = p0y - L.oy, cz = p0z - L.oz;             // :176
            const float rx = L.dy * cz - L.dz * cy;                                    // :177
            const float ry = L.dz * cx - L.dx * cz;
            const float rz = L.dx * cy - L.dy * cx;
            const float inv_det = 1.0f / nDotDir;                                      // :178
            const float u = ((rx * e2x + ry * e2y) + rz * e2z) * inv_det;              // :180
            const float v = ((rx * e1x + ry * e1y) + rz * e1z) * inv_det;              // :181
            const float w = VT_MUT(12, 1.0f - (u + v), 1.0f - u - v);                  // :182
            const float t = ((nx * cx + ny * cy) + nz * cz) * inv_det;                 // :188
            bool hit = !culled && VT_MUT(8, u > 0.0f, u >= 0.0f) && VT_MUT(14, v > 0.0f, v >= 0.0f) && VT_MUT(15, w > 0.0f, w >= 0.0f) &&        // :187
                       VT_MUT(11, t > L.tmin, t >= L.tmin) && VT_MUT(2, t < L.tmax, t <= L.tmax);    // :189
            if constexpr (ALPHA) {                                                     // :196-208
                if (hit && (tflags & VT_TRI_ALPHATEST)) {        // the candidate is parked until its alpha is known
                    L.astate = 1; L.cprim = tprim; L.cu = u; L.cv = v; L.ct = t;
                    hit = false;
                }
            }
            if (hit) {
                L.prim = tprim; L.u = u; L.v = v; L.tmax = t;
                if constexpr (ANY_HIT) { L.tri_cur = L.tri_end; L.node = kDone; }
            }
        } else if (ALPHA && (do_alpha1 || alpha2)) {
            if constexpr (ALPHA) {
                bool decided = false, pass = false;
                if (!STATS && alpha2) {
                    // ---- ALPHA 2: the texels requested one round ago: alpha, then :205.  (In program order BEFORE the block
                    // that issues new texel loads, so the wait below never waits for loads of this round; with the DMA fetch
                    // it is already satisfied -- the record wait of this round covered the older texel loads.)
                    uint32_t t0, t1, t2, t3;
                    asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, v76\n\tv_mov_b32 %1, v77\n\tv_mov_b32 %2, v78\n\tv_mov_b32 %3, v79"
                                 : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : : "memory");
                    const float alpha = alpha_from_texels(t0, t1, t2, t3, L.ax, L.ay);
                    decided = true; pass = !VT_MUT(61, alpha <= L.aref, alpha < L.aref);
                }
                if (do_alpha1) {
                    // ---- ALPHA 1: the triangle's AlphaRec has arrived: texUV (:198), TransformTexcoord, texel addresses
                    const float w = 1.0f - L.cu - L.cv;
                    const float tx = (w * q0.x + L.cu * q0.z) + L.cv * q1.x;
                    const float ty = (w * q0.y + L.cu * q0.w) + L.cv * q1.y;
                    const float ss = ((tx * q1.z + ty * q1.w) + q2.x) * q3.x;
                    const float tt = ((tx * q2.y + ty * q2.z) + q2.w) * q3.x;
                    const uint32_t dims = __float_as_uint(q3.z), of = __float_as_uint(q3.w);
                    L.aref = q3.y;
                    if (dims == kAlphaAlwaysPass) { decided = true; pass = true; }
                    else if (dims == 0u) { decided = true; pass = VT_MUT(65, true, !(1.0f < L.aref)); }         // no texture: alpha 1
                    else {
                        const uint32_t W = dims & 0xFFFFu, H = dims >> 16;
                        const uint32_t plane = of & 0x7FFFFFFFu;          // texel offsets fit 32 bits (vt_scene_set_alpha: < 2 GiB)
                        float x = ss * float(W), y = tt * float(H);
                        if (!(fabsf(x) < 1.0e9f)) x = 0.0f;
                        if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
                        // STATS (diagnostic) variants read the texels in place -- the wave waits, only results and counters
                        // matter there, and they may need more registers than the cap leaves.  The others issue the loads
                        // behind the compiler's back (a load it knows of is waited for at the end of the block) into the
                        // reserved registers; ALPHA 2 reads them one round later.
                        uint32_t p0, p1 = plane, p2 = plane, p3 = plane;
                        if ((of >> 31) == 0u) {
                            const uint32_t xi = wrap_index(floorf(x), W), yi = wrap_index(floorf(y), H);
                            p0 = plane + yi * W + xi;
                            L.ax = -1.0f; L.ay = 0.0f;
                        } else {
                            const float fx = x - VT_MUT(63, 0.0f, 0.5f), fy = y - 0.5f;
                            const float x0 = floorf(fx), y0 = floorf(fy);
                            L.ax = fx - x0; L.ay = fy - y0;
                            const uint32_t i0 = wrap_index(x0, W), i1 = VT_MUT(64, min(i0 + 1u, W - 1u), wrap_index(x0 + 1.0f, W));
                            const uint32_t j0 = wrap_index(y0, H), j1 = wrap_index(y0 + 1.0f, H);
                            p0 = plane + j0 * W + i0; p1 = plane + j0 * W + i1;
                            p2 = plane + j1 * W + i0; p3 = plane + j1 * W + i1;
                        }
                        if constexpr (STATS) {
                            decided = true;
                            const uint8_t* img = a.alpha_texels;
                            pass = !(alpha_from_texels(img[p0], img[p1], img[p2], img[p3], L.ax, L.ay) < L.aref);
                        } else {
                            if (L.ax < 0.0f)
                                asm volatile("global_load_ubyte v76, %0, %1" : : "v"(p0), "s"(a.alpha_texels) : VT_TEXEL_REGS);
                            else
                                asm volatile("global_load_ubyte v76, %0, %4\n\tglobal_load_ubyte v77, %1, %4\n\t"
                                             "global_load_ubyte v78, %2, %4\n\tglobal_load_ubyte v79, %3, %4"
                                             : : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(a.alpha_texels) : VT_TEXEL_REGS);
                            L.astate = 2;
                        }
                    }
                }
                if (decided) {
                    L.astate = 0;
                    if (pass) {
                        L.prim = L.cprim; L.u = L.cu; L.v = L.cv; L.tmax = L.ct;
                        if constexpr (ANY_HIT) { L.tri_cur = L.tri_end; L.node = kDone; }
                    }
                }
            }
        } else if (want_node) {
            // ---- NODE: one iteration of SingleRayTraverser::traverse -------------------
            VT_ISA_MARK("node");
            if constexpr (STATS) ++L.steps;
            // bounds[2a + octant[a]] (entry side) and bounds[2a + 1 - octant[a]] (exit side) of
            // the left and the right child; counts and first indices
            float lnx, lfx, lny, lfy, lnz, lfz, rnx, rfx, rny, rfy, rnz, rfz;
            uint32_t lcount, lfirst, rcount, rfirst;
            {
                // octant = signbit(dir).  The selects are written as v_cndmask with an SGPR-pair
                // mask: the VCC form hipcc often picks issues ~4x slower on gfx950 (measured,
                // scripts/ubench_valu.hip: 16 vs 4.2 cycles per wave-instruction).
                const uint64_t mx = __ballot(__float_as_uint(L.dx) >> 31), my = __ballot(__float_as_uint(L.dy) >> 31),
                               mz = __ballot(__float_as_uint(L.dz) >> 31);
                // left: bounds {q0.x q0.y | q0.z q0.w | q1.x q1.y}; right: {q2.x q2.y | q2.z q2.w | q3.x q3.y}
                lnx = sel(mx, q0.y, q0.x); lfx = sel(mx, q0.x, q0.y);
                lny = sel(my, q0.w, q0.z); lfy = sel(my, q0.z, q0.w);
                lnz = sel(mz, q1.y, q1.x); lfz = sel(mz, q1.x, q1.y);
                rnx = sel(mx, q2.y, q2.x); rfx = sel(mx, q2.x, q2.y);
                rny = sel(my, q2.w, q2.z); rfy = sel(my, q2.z, q2.w);
                rnz = sel(mz, q3.y, q3.x); rfz = sel(mz, q3.x, q3.y);
                lcount = __float_as_uint(q1.z); lfirst = __float_as_uint(q1.w);
                rcount = __float_as_uint(q3.z); rfirst = __float_as_uint(q3.w);
            }
            const float fl = slab_first(lnx * L.ix + L.sx, lny * L.iy + L.sy, lnz * L.iz + L.sz, L.tmin);
            const float sl = slab_second(lfx * L.ix + L.sx, lfy * L.iy + L.sy, lfz * L.iz + L.sz, L.tmax);
            const float fr = slab_first(rnx * L.ix + L.sx, rny * L.iy + L.sy, rnz * L.iz + L.sz, L.tmin);
            const float sr = slab_second(rfx * L.ix + L.sx, rfy * L.iy + L.sy, rfz * L.iz + L.sz, L.tmax);

            const bool hit_l = VT_MUT(7, fl < sl, fl <= sl), hit_r = VT_MUT(7, fr < sr, fr <= sr);
            const bool leaf_l = lcount != 0, leaf_r = rcount != 0;

            // leaves that were hit become the pending triangle range, left before right;
            // two leaf siblings are contiguous in the leaf-ordered triangle array
            uint32_t tc = 0, te = 0;
            if (hit_l && leaf_l) { tc = lfirst; te = lfirst + lcount; }
            if (hit_r && leaf_r) { if (te == 0) tc = rfirst; te = rfirst + rcount; }
            L.tri_cur = tc; L.tri_end = te;

            const bool go_l = hit_l && !leaf_l, go_r = hit_r && !leaf_r;
            uint32_t next;
            if (go_l && go_r) {
                // near child first (ties keep left first); push the far child's pair
                const bool swap = VT_MUT(1, fl >= fr, fl > fr);
                next = swap ? rfirst : lfirst;
                uint32_t far = swap ? lfirst : rfirst;
                if (VT_MUT(16, L.sp <= a.lds_entries, L.sp < a.lds_entries)) st_lds[L.sp * 64] = far;
                else st_ovf[size_t(L.sp - a.lds_entries) * gstride] = VT_MUT(10, next, far);
                ++L.sp;
            } else if (go_l) {
                next = lfirst;
            } else if (go_r) {
                next = rfirst;
            } else if (L.sp != 0) {
                --L.sp;
                // two separate loads on purpose: a pointer select would turn this into a flat_load
                if (L.sp < a.lds_entries) next = st_lds[L.sp * 64];
                else next = st_ovf[size_t(L.sp - a.lds_entries) * gstride];
            } else {
                next = kDone;
            }
            L.node = next;
        }
        VT_ISA_MARK("finish");
        if (has_ray && L.node == kDone && L.tri_cur >= L.tri_end && (!ALPHA || L.astate == 0)) finish_ray();
    }
    VT_ISA_MARK("exit");
    if constexpr (PERSISTENT) {
        if (lane == 0) leave_grid(a, 1);
    }
}

// The two kernels around the body.  The variants without the alpha test are compiled as before; the ALPHA variants ask the
// compiler to stay within kCompilerVgprs registers (see VT_TEXEL_REGS) -- an attribute cannot depend on a template argument,
// hence two templates (`ALPHA` stays a parameter of the first so that its name reads as in rounds 1 and 2).
template <bool ANY_HIT, bool STATS, bool PERSISTENT, bool FETCH_DMA, bool ALPHA>
__global__ __launch_bounds__(kBlockThreads) void trace_kernel(TraceArgs a)
{
    static_assert(!ALPHA, "the alpha-test variants are trace_kernel_alpha");
    trace_body<ANY_HIT, STATS, PERSISTENT, FETCH_DMA, false>(a);
}

template <bool ANY_HIT, bool STATS, bool PERSISTENT, bool FETCH_DMA>
__global__ __launch_bounds__(kBlockThreads) __attribute__((amdgpu_num_vgpr(kCompilerVgprs))) void trace_kernel_alpha(TraceArgs a)
{
    trace_body<ANY_HIT, STATS, PERSISTENT, FETCH_DMA, true>(a);
}

// closest hit with the ray count in device memory (TraceArgs::live_n): the traces of vt_bounce_loop_dev behind its first depth
template <bool PERSISTENT, bool FETCH_DMA>
__global__ __launch_bounds__(kBlockThreads) void trace_kernel_devn(TraceArgs a)
{
    trace_body<false, false, PERSISTENT, FETCH_DMA, false, true>(a);
}

// ---- TraceResult batch core: TraceResult.cpp:45-86, 255-262 -----------------------------
__device__ __forceinline__ vt_hit_attrs make_hit_attrs(const vt_tri64& T, const vt_ray& r, const vt_hit& h)
{
    vt_hit_attrs o;
    // AccelStruct.cpp:826 glm::normalize(dir); TraceResult.cpp:56 wo = -direction
    const float d2 = (r.dir[0] * r.dir[0] + r.dir[1] * r.dir[1]) + r.dir[2] * r.dir[2];
    const float inv = 1.0f / sqrtf(d2);
    o.wo[0] = -(r.dir[0] * inv); o.wo[1] = -(r.dir[1] * inv); o.wo[2] = -(r.dir[2] * inv);
    const float w = VT_MUT(21, 1.0f - (h.u + h.v), 1.0f - h.u - h.v);  // TraceResult.cpp:70
    o.uvw[0] = h.u; o.uvw[1] = h.v; o.uvw[2] = w;
    const float len = sqrtf((T.n[0] * T.n[0] + T.n[1] * T.n[1]) + T.n[2] * T.n[2]);
    for (int k = 0; k < 3; ++k) {
        o.ngeo[k] = T.n[k] / len;                                   // Primitives.h:100
        const float v0 = T.p0[k], v1 = T.p0[k] - T.e1[k], v2 = T.p0[k] + T.e2[k];
        o.pos[k] = (w * v0 + h.u * v1) + h.v * v2;                  // TraceResult.cpp:258
    }
    o.t = h.t;
    o.prim = h.prim;
    const float facing = (o.wo[0] * o.ngeo[0] + o.wo[1] * o.ngeo[1]) + o.wo[2] * o.ngeo[2];
    o.front = VT_MUT(22, facing > 0.0f, facing >= 0.0f) ? 1u : 0u; // :85
    o.hit = 1;
    return o;
}

__global__ __launch_bounds__(kBlockThreads) void hit_attrs_kernel(HitAttrsArgs a)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlockThreads + threadIdx.x;
    if (i >= a.n) return;
    const vt_hit h = a.hits[i];
    vt_hit_attrs o;
    if (h.prim == VT_MISS) o = vt_hit_attrs{};
    else o = make_hit_attrs(a.tris[a.prim_to_slot[h.prim]], a.rays[i], h);
    a.attrs[i] = o;
}

// ---- rest of the TraceResult constructor: TraceResult.cpp:73-78 ---------------------------------
__global__ __launch_bounds__(kBlockThreads) void hit_shade_kernel(HitShadeArgs a)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlockThreads + threadIdx.x;
    if (i >= a.n) return;
    const vt_hit h = a.hits[i];
    vt_hit_shade o{};
    if (h.prim == VT_MISS) {
        o.ent_id = VT_MISS; o.material = VT_MISS;
    } else {
        const vt_tri_attribs A = a.attribs[h.prim];
        const float w = 1.0f - h.u - h.v;                                               // uvw = (u, v, 1-u-v) :70
        o.blend = (w * A.alpha[0] + h.u * A.alpha[1]) + h.v * A.alpha[2];               // :73
        o.tex_uv[0] = (w * A.uv[0][0] + VT_MUT(23, h.v, h.u) * A.uv[1][0]) + VT_MUT(23, h.u, h.v) * A.uv[2][0];   // :74
        o.tex_uv[1] = (w * A.uv[0][1] + h.u * A.uv[1][1]) + h.v * A.uv[2][1];
        o.ent_id = A.ent_id;                                                            // :76
        o.material = A.material;                                                        // :78
    }
    a.out[i] = o;
}

// ---- device-side ray generation (harness for wavefront callers) ----
constexpr uint32_t kGenPixelsPerThread = 8;   // the camera basis (two sqrt, six divides, one tan in fp64) is set up once per thread

__global__ __launch_bounds__(kBlockThreads) void gen_primary_kernel(GenPrimaryArgs a)
{
    const uint32_t w = a.cam.width, h = a.cam.height;
    const uint64_t n = uint64_t(w) * h;
    // double precision set-up, rounded once to fp32 (as the host generator does)
    double f[3] = {a.cam.forward[0], a.cam.forward[1], a.cam.forward[2]};
    double up[3] = {a.cam.up[0], a.cam.up[1], a.cam.up[2]};
    double fl = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    f[0] /= fl; f[1] /= fl; f[2] /= fl;
    double r[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    double rl = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    r[0] /= rl; r[1] /= rl; r[2] /= rl;
    const double u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};
    const double th = tan(double(a.cam.vfov_deg) * 3.14159265358979323846 / 180.0 / 2.0);
    // consecutive lanes write consecutive rays; a thread's pixels lie one block apart
    uint64_t i = uint64_t(blockIdx.x) * (kBlockThreads * kGenPixelsPerThread) + threadIdx.x;
    for (uint32_t k = 0; k < kGenPixelsPerThread && i < n; ++k, i += kBlockThreads) {
        const uint32_t px = uint32_t(i % w), py = uint32_t(i / w);
        const double sx = ((double(px) + 0.5) / w * 2.0 - 1.0) * th * (double(w) / h);
        const double sy = (1.0 - (double(py) + 0.5) / h * 2.0) * th;
        double d[3] = {f[0] + sx * r[0] + sy * u[0], f[1] + sx * r[1] + sy * u[1], f[2] + sx * r[2] + sy * u[2]};
        const double dl = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        vt_ray ray;
        for (int c = 0; c < 3; ++c) { ray.org[c] = a.cam.pos[c]; ray.dir[c] = float(d[c] / dl); }
        ray.tmin = 0.f; ray.tmax = FLT_MAX;
        a.rays[i] = ray;
    }
}

__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one bounce ray from a hit record; i = the counter of the path (its index in the batch)
__device__ __forceinline__ vt_ray make_bounce_ray(const vt_hit_attrs& A, uint64_t seed, uint64_t i)
{
    vt_ray ray{};
    // geometric normal flipped towards wo (the side the ray arrived from)
    float n[3];
    for (int k = 0; k < 3; ++k) n[k] = A.front ? A.ngeo[k] : -A.ngeo[k];
    // vistrace.CalcRayOrigin, VisTrace.cpp:1495-1517
    const float origin = 1.f / 32.f, fScale = 1.f / 65536.f, iScale = 256.f;
    for (int k = 0; k < 3; ++k) {
        const int32_t iOff = int32_t(n[k] * iScale);
        const int32_t bits = int32_t(__float_as_uint(A.pos[k])) + (A.pos[k] < 0.f ? -iOff : iOff);
        const float iPos = __uint_as_float(uint32_t(bits));
        ray.org[k] = VT_MUT(27, fabsf(A.pos[k]) <= origin, fabsf(A.pos[k]) < origin) ? A.pos[k] + n[k] * fScale : iPos;
    }
    // hemisphere_cos, BSDF.cpp:69-77, samples = top 24 bits of splitmix64 outputs 2i and 2i+1
    const float r1 = float(splitmix64_at(seed, 2 * i) >> 40) * (1.0f / 16777216.0f);
    const float r2 = float(splitmix64_at(seed, 2 * i + 1) >> 40) * (1.0f / 16777216.0f);
    const float z = sqrtf(r1), sinTheta = sqrtf(1.f - r1), phi = 2.f * 3.14159265358979323846f * r2;
    const float lx = sinTheta * VT_MUT(28, sinf(phi), cosf(phi)), ly = sinTheta * VT_MUT(28, cosf(phi), sinf(phi));
    // orthonormal basis around n (Duff et al. 2017), as vistrace_amd/workloads.py::_onb
    const float sign = n[2] >= 0.f ? 1.f : -1.f;
    const float aa = -1.f / (sign + n[2]);
    const float b = n[0] * n[1] * aa;
    const float b1[3] = {1.f + sign * n[0] * n[0] * aa, sign * b, -sign * n[0]};
    const float b2[3] = {b, sign + n[1] * n[1] * aa, -n[1]};
    for (int k = 0; k < 3; ++k) ray.dir[k] = (b1[k] * lx + b2[k] * ly) + n[k] * z;
    ray.tmin = 0.f; ray.tmax = FLT_MAX;
    return ray;
}

__global__ __launch_bounds__(kBlockThreads) void gen_bounce_kernel(GenBounceArgs a)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlockThreads + threadIdx.x;
    if (i >= a.n) return;
    const vt_hit_attrs A = a.attrs[i];
    if (A.hit == 0) {   // null ray: keeps batch size and order, cannot hit anything
        vt_ray ray{};
        ray.dir[0] = 1.f; ray.tmin = 0.f; ray.tmax = 1e-30f;
        a.rays[i] = ray;
        return;
    }
    a.rays[i] = make_bounce_ray(A, a.seed, i);
}

// ---- device-resident bounce loop: queue of live paths, compacted in path order ----
// The queue holds m entries; with QueueArgs::m_dev the host only knows an upper bound (a.m) and the real count is a device word.
// The kernels therefore run over 256-entry CHUNKS in a grid-stride loop bounded by the real count: a launch sized for 16 Mi
// entries whose queue holds half of them does not start 32 Ki blocks that find nothing to do (first form of round 6: one block
// per chunk of the upper bound, 0.12 ms per depth slower than the host-synchronised loop on an open scene).
__device__ __forceinline__ uint64_t queue_size(const QueueArgs& a) { return a.m_dev ? uint64_t(*a.m_dev) : a.m; }

// step 1: hits per 256-entry chunk of the queue
__global__ __launch_bounds__(kBlockThreads) void queue_count_kernel(QueueArgs a)
{
    __shared__ uint32_t wave_count[kBlockThreads / 64];
    const uint64_t m = queue_size(a);
    const uint32_t chunks = uint32_t((m + kBlockThreads - 1) / kBlockThreads);
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint64_t j = uint64_t(chunk) * kBlockThreads + threadIdx.x;
        const bool live = VT_MUT(51, j < m + (m < a.m ? 1u : 0u), j < m) && a.hits_q[j].prim != VT_MISS;
        const uint64_t mask = __ballot(live);
        if ((threadIdx.x & 63u) == 0) wave_count[threadIdx.x >> 6] = uint32_t(__popcll(mask));
        __syncthreads();
        if (threadIdx.x == 0) a.block_offsets[chunk] = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
}

// step 2: exclusive scan of the chunk counts in place (one block; <= 2^24 chunks), total -> *live_out.  Every thread owns a
// contiguous run of counts, a multiple of four long, and moves it as 16-B words (the runs of neighbouring lanes lie 256 B apart:
// with 4-B loads this one block took 0.1 ms for the 65 536 chunks of a 16 Mi-entry queue, 2 % of a depth).
__global__ __launch_bounds__(1024) void queue_scan_kernel(QueueArgs a, uint32_t* live_out)
{
    __shared__ uint32_t part[1024];
    uint32_t* const counts = a.block_offsets;
    const uint32_t nblocks = uint32_t((queue_size(a) + kBlockThreads - 1) / kBlockThreads);
    const uint32_t per = ((nblocks + 1023u) / 1024u + 3u) & ~3u;        // (block_offsets is 256-B aligned: every run starts on 16 B)
    const uint32_t lo = min(threadIdx.x * per, nblocks), hi = min(lo + per, nblocks);
    const uint32_t hi4 = lo + ((hi - lo) & ~3u);
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi4; k += 4) { const uint4 c = *reinterpret_cast<const uint4*>(counts + k); sum += (c.x + c.y) + (c.z + c.w); }
    for (uint32_t k = hi4; k < hi; ++k) sum += counts[k];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {                    // Hillis-Steele inclusive scan of the partials
        const uint32_t add = VT_MUT(53, threadIdx.x > d, threadIdx.x >= d) ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = lo; k < hi4; k += 4) {
        const uint4 c = *reinterpret_cast<const uint4*>(counts + k);
        uint4 o;
        o.x = run; o.y = o.x + c.x; o.z = o.y + c.y; o.w = o.z + c.z;
        run = o.w + VT_MUT(52, c.z, c.w);
        *reinterpret_cast<uint4*>(counts + k) = o;
    }
    for (uint32_t k = hi4; k < hi; ++k) { const uint32_t c = counts[k]; counts[k] = run; run += c; }
    if (threadIdx.x == 1023u) *live_out = part[1023];
}

// step 3: scatter this depth's hits to their paths; every hit emits its bounce ray and path id to the next queue
__global__ __launch_bounds__(kBlockThreads) void queue_emit_kernel(QueueArgs a)
{
    __shared__ uint32_t wave_count[kBlockThreads / 64];
    const uint64_t m = queue_size(a);
    const uint32_t chunks = uint32_t((m + kBlockThreads - 1) / kBlockThreads);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint64_t j = uint64_t(chunk) * kBlockThreads + threadIdx.x;
        const bool valid = j < m;
        vt_hit h{VT_MISS, 0.f, 0.f, 0.f};
        if (valid) h = a.hits_q[j];
        const uint32_t path = valid ? (a.ids_q ? a.ids_q[j] : uint32_t(j)) : 0u;
        if (VT_MUT(56, valid && h.prim != VT_MISS, valid) && a.hits_out) a.hits_out[path] = h;
        if (!a.rays_next) continue;                                // last depth: nothing to emit (wave-uniform)
        const bool live = valid && h.prim != VT_MISS;
        const uint64_t mask = __ballot(live);
        if (lane == 0) wave_count[wave] = uint32_t(__popcll(mask));
        __syncthreads();
        if (live) {
            uint32_t dst = a.block_offsets[chunk] + uint32_t(__popcll(mask & ((uint64_t(VT_MUT(54, 2, 1)) << lane) - 1)));
            for (uint32_t w = 0; w < wave; ++w) dst += wave_count[w];
            const vt_hit_attrs A = make_hit_attrs(a.tris[a.prim_to_slot[h.prim]], a.rays_q[j], h);
            a.rays_next[dst] = make_bounce_ray(A, a.seed, path);
            a.ids_next[dst] = path;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlockThreads) void fill_miss_kernel(vt_hit* hits, uint64_t n, const uint32_t* count)
{
    if (count && VT_MUT(55, uint64_t(*count) + 1 >= n, uint64_t(*count) == n)) return;        // every path is still alive: the queue step writes the whole row
    for (uint64_t i = uint64_t(blockIdx.x) * kBlockThreads + threadIdx.x; i < n; i += uint64_t(gridDim.x) * kBlockThreads)
        hits[i] = vt_hit{VT_MISS, 0.f, 0.f, 0.f};
}

// ---- refit: triangle records and pair bounds recomputed in place ----------
// triangle record from three vertices: Triangle ctor Primitives.h:82 + ComputeNormalAndLoD :93
__device__ __forceinline__ void store_tri_record(const float* v, uint32_t prim, uint32_t flags, vt_tri64* dst, uint32_t* bad)
{
    bool finite = true;
    for (int k = 0; k < 9; ++k) finite = finite && fabsf(v[k]) <= FLT_MAX;
    if (!finite && bad) atomicAdd(bad, 1u);          // the engine refuses to trace a scene refitted with such data
    vt_tri64 t;
    for (int k = 0; k < 3; ++k) {
        t.p0[k] = v[k];
        t.e1[k] = VT_MUT(24, v[3 + k] - v[k], v[k] - v[3 + k]);
        t.e2[k] = v[6 + k] - v[k];
    }
    t.n[0] = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
    t.n[1] = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
    t.n[2] = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    t.prim = prim;
    t.flags = flags;
    t.pad[0] = t.pad[1] = 0;
    *dst = t;
}

__global__ __launch_bounds__(kBlockThreads) void refit_tris_kernel(RefitTrisArgs a)
{
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= a.n) return;
    const float* src = a.verts + size_t(i) * 9;
    float v[9];
    for (int k = 0; k < 9; ++k) v[k] = src[k];
    const uint32_t slot = a.prim_to_slot[i];
    store_tri_record(v, i, a.flags ? uint32_t(a.flags[i]) : a.tris[slot].flags, &a.tris[slot], a.bad);
}

// bones[i] * binds[i], the product TransformToBone forms per vertex and bone (AccelStruct.cpp:44);
// glm mat4 * mat4: Result[c] = A[0]*B[c][0] + A[1]*B[c][1] + A[2]*B[c][2] + A[3]*B[c][3].  One thread per element.
__global__ __launch_bounds__(kBlockThreads) void skin_matrices_kernel(SkinMatricesArgs a)
{
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= a.nmat * 16u) return;
    const uint32_t m = i >> 4, c = (i >> 2) & 3u, r = i & 3u;
    const float* A = a.bones + size_t(m) * 16;
    const float* B = a.binds + size_t(m) * 16;
    float acc = A[r] * B[c * 4];
    acc = acc + A[4 + r] * B[c * 4 + 1];
    acc = acc + A[8 + r] * B[c * 4 + 2];
    acc = acc + A[12 + r] * B[c * 4 + 3];
    a.mats[i] = acc;
}

// SkinTriangle (AccelStruct.cpp:66-102), positions only: one thread per triangle; the three vertices are
// re-derived through p0/e1/e2 (:68-72), moved by TransformToBone (:35-47; glm mat4 * vec4 =
// (m0*x + m1*y) + (m2*z + m3*w), then * weight, accumulated from 0) and the record rebuilt in its leaf slot.
__global__ __launch_bounds__(kBlockThreads) void skin_tris_kernel(SkinTrisArgs a)
{
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= a.n) return;
    const float* b = a.bind_verts + size_t(i) * 9;
    float pos[9], v[9];
    for (int k = 0; k < 3; ++k) {
        const float p0 = b[k], e1 = b[k] - b[3 + k], e2 = b[6 + k] - b[k];
        pos[k] = p0; pos[3 + k] = p0 - e1; pos[6 + k] = p0 + e2;
    }
    const uint32_t base = a.matrix_base[i];
    for (int vi = 0; vi < 3; ++vi) {
        const vt_skin_vertex sv = a.skin[size_t(i) * 3 + vi];
        float fin[3] = {0.f, 0.f, 0.f};
        for (uint32_t q = 0; q < sv.num_bones && q < 3u; ++q) {
            uint32_t mi = base + uint32_t(int(sv.bone[q]));
            mi = mi < a.nmat ? mi : 0u;                            // out-of-range bone id: stay inside the table
            // the four columns as 16-B loads: the kernel is bound by the gather rate on the matrix table (scalar loads: 89 us per 1 M triangles)
            const float4* M = reinterpret_cast<const float4*>(a.mats + size_t(mi) * 16);
            const float4 c0 = M[0], c1 = M[1], c2 = M[2], c3 = M[3];
            const float m0[3] = {c0.x, c0.y, c0.z}, m1[3] = {c1.x, c1.y, c1.z}, m2[3] = {c2.x, c2.y, c2.z}, m3[3] = {c3.x, c3.y, c3.z};
            for (int r = 0; r < 3; ++r) {
                const float a0 = m0[r] * pos[vi * 3] + m1[r] * pos[vi * 3 + 1];
                const float a1 = m2[r] * pos[vi * 3 + 2] + m3[r] * 1.f;
                fin[r] = fin[r] + (a0 + a1) * sv.weight[VT_MUT(26, 0u, q)];
            }
        }
        v[vi * 3] = fin[0]; v[vi * 3 + 1] = fin[1]; v[vi * 3 + 2] = fin[2];
    }
    const uint32_t slot = a.prim_to_slot[i];
    store_tri_record(v, i, a.tris[slot].flags, &a.tris[slot], a.bad);
}

__device__ __forceinline__ void box_of_child(const vt_node_pair* pairs, const vt_tri64* tris, const vt_bvh_node& c, float* b)
{
    float lo[3], hi[3];
    if (c.prim_count != 0) {                           // leaf: Triangle::bounding_box(), Primitives.h:107-113
        for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
        for (uint32_t q = 0; q < c.prim_count; ++q) {
            const vt_tri64& t = tris[c.first + q];
            for (int k = 0; k < 3; ++k) {
                const float p0 = t.p0[k], p1 = t.p0[k] - t.e1[k], p2 = t.p0[k] + t.e2[k];
                float l = p0, h = p0;
                l = p1 < l ? p1 : l; h = p1 > h ? p1 : h;
                l = VT_MUT(25, l, p2 < l ? p2 : l); h = VT_MUT(25, h, p2 > h ? p2 : h);
                lo[k] = l < lo[k] ? l : lo[k]; hi[k] = h > hi[k] ? h : hi[k];
            }
        }
    } else {                                           // inner: union of its two (already refitted) children
        const vt_node_pair& p = pairs[c.first];
        for (int k = 0; k < 3; ++k) {
            const float l0 = p.child[0].bounds[2 * k], l1 = p.child[1].bounds[2 * k];
            const float h0 = p.child[0].bounds[2 * k + 1], h1 = p.child[1].bounds[2 * k + 1];
            lo[k] = l0 < l1 ? l0 : l1; hi[k] = h0 > h1 ? h0 : h1;
        }
    }
    for (int k = 0; k < 3; ++k) { b[2 * k] = lo[k]; b[2 * k + 1] = hi[k]; }
}

// one thread per (pair, child) of one tree level; levels run deepest first
__global__ __launch_bounds__(kBlockThreads) void refit_level_kernel(RefitLevelArgs a)
{
    const uint32_t i = blockIdx.x * kBlockThreads + threadIdx.x;
    if (i >= a.count * 2) return;
    vt_node_pair& p = a.pairs[a.level_pairs[i >> 1]];
    vt_bvh_node& c = p.child[i & 1];
    float b[6];
    box_of_child(a.pairs, a.tris, c, b);
    for (int k = 0; k < 6; ++k) c.bounds[k] = b[k];
}

// ---- launchers ---------------------------------------------------------------------------
template <bool ANY_HIT, bool STATS, bool PERSISTENT, bool FETCH_DMA, bool ALPHA>
static hipError_t launch_one(const TraceArgs& a, dim3 grid, size_t lds_bytes, hipStream_t stream)
{
    if constexpr (ALPHA)
        hipLaunchKernelGGL((trace_kernel_alpha<ANY_HIT, STATS, PERSISTENT, FETCH_DMA>), grid, dim3(kBlockThreads), lds_bytes, stream, a);
    else
        hipLaunchKernelGGL((trace_kernel<ANY_HIT, STATS, PERSISTENT, FETCH_DMA, false>), grid, dim3(kBlockThreads), lds_bytes, stream, a);
    return hipGetLastError();
}

size_t trace_lds_bytes(uint32_t lds_entries, bool fetch_dma)
{
    size_t b = size_t(lds_entries) * 64 * sizeof(uint32_t) * (kBlockThreads / 64);
    // The stay-or-leave flag of a reserved CU lives in the 16-B pad of the last staging row (DMA) or in one extra
    // word.  Not a byte more with DMA: LDS is granted in 1 280-B granules (scripts/lds_granule_probe.py), 10 stack
    // entries + staging = 26 880 B = exactly 21 granules, and 16 B more would drop a CU from 6 to 5 resident blocks
    // (hipOccupancyMaxActiveBlocksPerMultiprocessor does not model the granule and still answers 6).
    if (fetch_dma) b += size_t(kStageBytes) * (kBlockThreads / 64);
    else b += 16;
    return b;
}

namespace {

// one entry per compiled variant: launch it, or ask how many blocks fit on a CU
template <bool ANY_HIT, bool STATS, bool PERSISTENT, bool FETCH_DMA, bool ALPHA>
hipError_t variant_op(const TraceArgs* a, dim3 grid, size_t lds_bytes, hipStream_t stream, int* blocks_per_cu)
{
    if (blocks_per_cu) {
        if constexpr (ALPHA)
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel_alpha<ANY_HIT, STATS, PERSISTENT, FETCH_DMA>,
                                                                int(kBlockThreads), lds_bytes);
        else
            return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, trace_kernel<ANY_HIT, STATS, PERSISTENT, FETCH_DMA, false>,
                                                                int(kBlockThreads), lds_bytes);
    }
    return launch_one<ANY_HIT, STATS, PERSISTENT, FETCH_DMA, ALPHA>(*a, grid, lds_bytes, stream);
}

template <bool ALPHA>
hipError_t dispatch(const TraceArgs* a, bool any_hit, bool stats, bool persistent, bool fetch_dma, dim3 grid,
                    size_t lds_bytes, hipStream_t stream, int* occ)
{
    // any-hit counters (vt_trace_any_stats_dev): one variant, one ray per lane -- counters do not depend on the schedule
    if (any_hit && stats) return variant_op<true, true, false, false, ALPHA>(a, grid, lds_bytes, stream, occ);
    if (persistent && fetch_dma) {
        if (any_hit) return variant_op<true, false, true, true, ALPHA>(a, grid, lds_bytes, stream, occ);
        if (stats)   return variant_op<false, true, true, true, ALPHA>(a, grid, lds_bytes, stream, occ);
        return variant_op<false, false, true, true, ALPHA>(a, grid, lds_bytes, stream, occ);
    }
    if (persistent) {
        if (any_hit) return variant_op<true, false, true, false, ALPHA>(a, grid, lds_bytes, stream, occ);
        if (stats)   return variant_op<false, true, true, false, ALPHA>(a, grid, lds_bytes, stream, occ);
        return variant_op<false, false, true, false, ALPHA>(a, grid, lds_bytes, stream, occ);
    }
    if (any_hit) return variant_op<true, false, false, false, ALPHA>(a, grid, lds_bytes, stream, occ);
    if (stats)   return variant_op<false, true, false, false, ALPHA>(a, grid, lds_bytes, stream, occ);
    return variant_op<false, false, false, false, ALPHA>(a, grid, lds_bytes, stream, occ);
}

} // namespace

hipError_t launch_trace(const TraceArgs& a, bool any_hit, bool stats, bool persistent, bool fetch_dma, bool alpha,
                        uint32_t grid_blocks, size_t lds_bytes, hipStream_t stream)
{
    if (a.live_n) {                        // the ray count lives on the device: closest hit, plain scenes (engine.hip checks)
        if (any_hit || stats || alpha || a.nseg != 0) return hipErrorInvalidValue;
        const dim3 grid(grid_blocks), block(kBlockThreads);
        if (persistent && fetch_dma) hipLaunchKernelGGL((trace_kernel_devn<true, true>), grid, block, lds_bytes, stream, a);
        else if (persistent) hipLaunchKernelGGL((trace_kernel_devn<true, false>), grid, block, lds_bytes, stream, a);
        else hipLaunchKernelGGL((trace_kernel_devn<false, false>), grid, block, lds_bytes, stream, a);
        return hipGetLastError();
    }
    return alpha ? dispatch<true>(&a, any_hit, stats, persistent, fetch_dma, dim3(grid_blocks), lds_bytes, stream, nullptr)
                 : dispatch<false>(&a, any_hit, stats, persistent, fetch_dma, dim3(grid_blocks), lds_bytes, stream, nullptr);
}

// Run-time half of the ALPHA build contract (check_isa.py is the build-time half): the variants that keep texel loads in
// flight in v76..v79 must not have been given more registers than 76 + those four.
hipError_t alpha_kernels_within_budget(bool* ok)
{
    *ok = true;
    auto probe = [&](const void* fn) -> hipError_t {
        hipFuncAttributes at;
        const hipError_t err = hipFuncGetAttributes(&at, fn);
        if (err == hipSuccess && at.numRegs > kCompilerVgprs + 4) *ok = false;
        return err;
    };
    hipError_t err = probe(reinterpret_cast<const void*>(&trace_kernel_alpha<false, false, true, true>));
    if (err == hipSuccess) err = probe(reinterpret_cast<const void*>(&trace_kernel_alpha<true, false, true, true>));
    if (err == hipSuccess) err = probe(reinterpret_cast<const void*>(&trace_kernel_alpha<false, false, true, false>));
    if (err == hipSuccess) err = probe(reinterpret_cast<const void*>(&trace_kernel_alpha<true, false, true, false>));
    if (err == hipSuccess) err = probe(reinterpret_cast<const void*>(&trace_kernel_alpha<false, false, false, false>));
    if (err == hipSuccess) err = probe(reinterpret_cast<const void*>(&trace_kernel_alpha<true, false, false, false>));
    return err;
}

hipError_t trace_blocks_per_cu(bool any_hit, bool stats, bool persistent, bool fetch_dma, bool alpha, size_t lds_bytes, int* out)
{
    return alpha ? dispatch<true>(nullptr, any_hit, stats, persistent, fetch_dma, dim3(1), lds_bytes, nullptr, out)
                 : dispatch<false>(nullptr, any_hit, stats, persistent, fetch_dma, dim3(1), lds_bytes, nullptr, out);
}

__global__ __launch_bounds__(kBlockThreads) void cu_probe_kernel(uint32_t* seen)
{
    extern __shared__ uint32_t probe_lds[];              // a large LDS footprint spreads the blocks over the CUs
    probe_lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();                 // 100 MHz
    while (wall_clock64() - t0 < 2000) probe_lds[threadIdx.x] += 1u;
    if (threadIdx.x == 0 && probe_lds[0] != 0xFFFFFFFFu) {
        const uint32_t id = __smid() & 1023u;
        atomicOr(&seen[id >> 5], 1u << (id & 31u));
    }
}

hipError_t launch_cu_probe(uint32_t* seen, uint32_t blocks, hipStream_t stream)
{
    const int lds = 60 * 1024;
    hipLaunchKernelGGL(cu_probe_kernel, dim3(blocks), dim3(kBlockThreads), lds, stream, seen);
    return hipGetLastError();
}

hipError_t launch_alpha_records(const AlphaRecArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(alpha_records_kernel, dim3((a.n + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_hit_attrs(const HitAttrsArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    const uint64_t blocks = (a.n + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(hit_attrs_kernel, dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_gen_primary(const GenPrimaryArgs& a, hipStream_t stream)
{
    const uint64_t n = uint64_t(a.cam.width) * a.cam.height;
    if (n == 0) return hipSuccess;
    const uint64_t per_block = uint64_t(kBlockThreads) * kGenPixelsPerThread;
    hipLaunchKernelGGL(gen_primary_kernel, dim3(uint32_t((n + per_block - 1) / per_block)), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_gen_bounce(const GenBounceArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(gen_bounce_kernel, dim3(uint32_t((a.n + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0,
                       stream, a);
    return hipGetLastError();
}

// grid of the streaming helper kernels of the bounce loop: one block per 256 entries up to a few blocks per CU, a loop beyond
constexpr uint32_t kQueueGridCap = 8192;

hipError_t launch_queue_step(const QueueArgs& a, uint32_t* live_out, hipStream_t stream)
{
    if (a.m == 0) return hipSuccess;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((a.m + kBlockThreads - 1) / kBlockThreads, kQueueGridCap));
    if (a.rays_next) {
        hipLaunchKernelGGL(queue_count_kernel, dim3(blocks), dim3(kBlockThreads), 0, stream, a);
        hipLaunchKernelGGL(queue_scan_kernel, dim3(1), dim3(1024), 0, stream, a, live_out);
    }
    hipLaunchKernelGGL(queue_emit_kernel, dim3(blocks), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_fill_miss(vt_hit* hits, uint64_t n, const uint32_t* count, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_miss_kernel, dim3(uint32_t(std::min<uint64_t>((n + kBlockThreads - 1) / kBlockThreads, kQueueGridCap))), dim3(kBlockThreads), 0, stream,
                       hits, n, count);
    return hipGetLastError();
}

hipError_t launch_refit_tris(const RefitTrisArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(refit_tris_kernel, dim3((a.n + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_skin_matrices(const SkinMatricesArgs& a, hipStream_t stream)
{
    if (a.nmat == 0) return hipSuccess;
    hipLaunchKernelGGL(skin_matrices_kernel, dim3((a.nmat * 16u + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                       stream, a);
    return hipGetLastError();
}

hipError_t launch_skin_tris(const SkinTrisArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(skin_tris_kernel, dim3((a.n + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_refit_level(const RefitLevelArgs& a, hipStream_t stream)
{
    if (a.count == 0) return hipSuccess;
    hipLaunchKernelGGL(refit_level_kernel, dim3((a.count * 2 + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0,
                       stream, a);
    return hipGetLastError();
}

hipError_t launch_hit_shade(const HitShadeArgs& a, hipStream_t stream)
{
    if (a.n == 0) return hipSuccess;
    const uint64_t blocks = (a.n + kBlockThreads - 1) / kBlockThreads;
    hipLaunchKernelGGL(hit_shade_kernel, dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

} // namespace vt
