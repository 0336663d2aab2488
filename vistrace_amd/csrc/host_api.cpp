// host_api.cpp -- the host-only half of the C ABI in include/vistrace_hip.h
// (error state, triangle set-up, BVH build, linearise).  No HIP calls in this file, so
// these entry points work on a machine without a GPU; everything that traces is in
// engine.hip and fails loudly there when no device is present.
#include "vt_internal.h"

#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>

namespace vt {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

// Test hooks (include/vistrace_hip.h, "Test hooks"): environment switches that change what the library does so that a one-GPU box
// can reach code a product run reaches only on other hardware.  They are dead unless VT_ENABLE_TEST_HOOKS=1 is set as well, and
// an active hook says so on stderr once -- a stray variable in a user's environment never changes product behaviour silently.
const char* test_hook(const char* name)
{
    const char* on = std::getenv("VT_ENABLE_TEST_HOOKS");
    if (!on || on[0] != '1' || on[1] != '\0') return nullptr;
    const char* v = std::getenv(name);
    if (!v || !*v) return nullptr;
    static std::mutex mu;
    static std::vector<std::string> announced;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(announced.begin(), announced.end(), name) == announced.end()) {
        announced.emplace_back(name);
        std::fprintf(stderr, "[vistrace_hip] TEST HOOK active: %s=%s (VT_ENABLE_TEST_HOOKS=1) -- not a product configuration\n", name, v);
    }
    return v;
}

// ---- fault injection on the allocation paths (include/vistrace_hip.h, "Test hooks") ------------------------------------------
namespace {
std::atomic<int>      g_alloc_hooks{-1};       // -1 unknown, 0 off, 1 on (VT_ENABLE_TEST_HOOKS=1, read once)
std::atomic<uint64_t> g_alloc_count{0};        // attempts since the hook was last armed
std::atomic<uint64_t> g_alloc_fail_at{0};      // 0 = none
std::atomic<uint64_t> g_hip_count{0}, g_hip_fail_at{0};      // the same pair for the checked HIP calls (VT_HIP sites)
std::atomic<void (*)()> g_test_drain{nullptr};

bool alloc_hooks_on()
{
    int on = g_alloc_hooks.load(std::memory_order_acquire);
    if (on < 0) {
        const char* en = std::getenv("VT_ENABLE_TEST_HOOKS");
        on = en && en[0] == '1' && en[1] == '\0' ? 1 : 0;
        if (on) if (const char* v = test_hook("VT_TEST_FAIL_ALLOC")) g_alloc_fail_at.store(std::strtoull(v, nullptr, 10), std::memory_order_release);
        if (on) if (const char* v = test_hook("VT_TEST_FAIL_HIP")) g_hip_fail_at.store(std::strtoull(v, nullptr, 10), std::memory_order_release);
        g_alloc_hooks.store(on, std::memory_order_release);
    }
    return on == 1;
}
} // namespace

void set_test_drain(void (*drain)()) { g_test_drain.store(drain, std::memory_order_release); }

bool test_hip_fails()
{
    if (g_alloc_hooks.load(std::memory_order_relaxed) == 0) return false;          // the product: one relaxed load
    if (!alloc_hooks_on()) return false;
    const uint64_t mine = g_hip_count.fetch_add(1, std::memory_order_acq_rel) + 1;
    const uint64_t at = g_hip_fail_at.load(std::memory_order_acquire);
    if (at == 0 || mine != at) return false;
    g_hip_fail_at.store(0, std::memory_order_release);
    if (auto drain = g_test_drain.load(std::memory_order_acquire)) drain();
    std::fprintf(stderr, "[vistrace_hip] TEST HOOK: checked HIP call %llu fails on purpose\n", static_cast<unsigned long long>(mine));
    return true;
}

bool test_alloc_fails()
{
    if (!alloc_hooks_on()) return false;
    const uint64_t mine = g_alloc_count.fetch_add(1, std::memory_order_acq_rel) + 1;
    uint64_t at = g_alloc_fail_at.load(std::memory_order_acquire);
    if (at == 0 || mine != at) return false;
    g_alloc_fail_at.store(0, std::memory_order_release);          // once: what cleans up or retries afterwards allocates normally
    std::fprintf(stderr, "[vistrace_hip] TEST HOOK: allocation %llu fails on purpose\n", static_cast<unsigned long long>(mine));
    return true;
}

void parallel_copy(void* dst, const void* src, size_t bytes)
{
    const long long piece = 1 << 20;
    const long long pieces = (static_cast<long long>(bytes) + piece - 1) / piece;
    static const long long cap = [] { const char* v = std::getenv("VT_COPY_THREADS"); return v ? std::atoll(v) : 8ll; }();
    const int threads = int(std::max<long long>(1, std::min<long long>({cap, pieces, omp_get_max_threads()})));
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long k = 0; k < pieces; ++k) {
        const size_t off = size_t(k) * size_t(piece);
        std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, std::min(size_t(piece), bytes - off));
    }
}

// parallel_copy of n rays that also looks at every ray's range on the way (the copy in `dst` is aligned, the source need not
// be): returns the index of the first ray with tmin < 0 or tmax <= tmin (the checks of AccelStruct::Traverse,
// source/objects/AccelStruct.cpp:805-806; NaN ranges pass, as they do there), or n if there is none.
uint64_t parallel_copy_checked(vt_ray* dst, const void* src, uint64_t n)
{
    const long long piece = 1 << 15;                         // rays per piece = 1 MiB
    const long long pieces = (static_cast<long long>(n) + piece - 1) / piece;
    const int threads = int(std::max<long long>(1, std::min<long long>({8, pieces, omp_get_max_threads()})));
    uint64_t bad = n;
#pragma omp parallel for num_threads(threads) schedule(static) reduction(min : bad)
    for (long long k = 0; k < pieces; ++k) {
        const uint64_t lo = uint64_t(k) * uint64_t(piece), hi = std::min<uint64_t>(n, lo + uint64_t(piece));
        std::memcpy(dst + lo, static_cast<const char*>(src) + lo * sizeof(vt_ray), (hi - lo) * sizeof(vt_ray));
        for (uint64_t i = lo; i < hi; ++i)
            if (dst[i].tmin < 0.f || dst[i].tmax <= dst[i].tmin) { bad = std::min(bad, i); break; }
    }
    return bad;
}

} // namespace vt

using namespace vt;

extern "C" {

const char* vt_last_error(void) { return g_last_error.c_str(); }
int vt_abi_version(void) { return VT_ABI_VERSION; }

// ---- ray sharding of the multi-GPU path: contiguous shards, every one but the tail of equal size, so
// that shard g of the rays and shard g of the gathered hit records both start at g * capacity ---------------------
uint64_t vt_shard_capacity(uint64_t n, int ndev)
{
    if (ndev <= 0) return 0;
    const uint64_t per = (n + VT_MUT(84, 0u, uint64_t(ndev) - 1)) / uint64_t(ndev);        // (VT_MUT: mutation sites, vt_internal.h)
    return VT_MUT(81, per, (per + 63) / 64 * 64);                       // whole 64-ray wave blocks: shard boundaries never split one
}

void vt_shard_bounds(uint64_t n, int ndev, int g, uint64_t* lo, uint64_t* hi)
{
    const uint64_t cap = vt_shard_capacity(n, ndev);
    const bool valid = g >= 0 && g < ndev;
    const uint64_t a = valid ? std::min(n, cap * uint64_t(g)) : n;
    if (lo) *lo = a;
    if (hi) *hi = valid ? VT_MUT(82, a + cap, std::min(n, a + cap)) : a;
}

int vt_tris_setup(const float* verts, const uint8_t* flags, uint32_t n, vt_tri64* out)
{
    if (n != 0 && (!verts || !out)) return fail(VT_ERR_INVALID_ARG, "vt_tris_setup: NULL argument");
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        const float* v = verts + size_t(i) * 9;
        tri_setup(v, v + 3, v + 6, uint32_t(i), flags ? flags[i] : 0u, out[i]);
    }
    return VT_OK;
}

int vt_bvh_build(const vt_tri64* tris, uint32_t n, int nthreads, vt_bvh** out)
{
    // default builder: binned SAH (fewer traversal steps per ray than the PLOC tree for the same Rebuild time);
    // VT_BUILDER=ploc in the environment selects the reference's algorithm without a code change
    int builder = VT_BUILDER_BINNED_SAH;
    if (const char* e = std::getenv("VT_BUILDER")) {
        if (std::strcmp(e, "ploc") == 0) builder = VT_BUILDER_PLOC;
        else if (std::strcmp(e, "sah") == 0) builder = VT_BUILDER_BINNED_SAH;
        else if (std::strcmp(e, "sah_refined") == 0) builder = VT_BUILDER_BINNED_SAH_REFINED;
    }
    return vt_bvh_build_ex(tris, n, nthreads, builder, out);
}

int vt_bvh_build_ex(const vt_tri64* tris, uint32_t n, int nthreads, int builder, vt_bvh** out)
{
    if (!out) return fail(VT_ERR_INVALID_ARG, "vt_bvh_build: out is NULL");
    *out = nullptr;
    try {
        vt_bvh* b = new vt_bvh();
        int rc = bvh_build(tris, n, nthreads, builder, b->bvh);
        if (rc != VT_OK) { delete b; return rc; }
        *out = b;
        return VT_OK;
    } catch (const std::bad_alloc&) {
        return fail(VT_ERR_NOMEM, "vt_bvh_build: out of host memory");
    } catch (const std::exception& e) {
        return fail(VT_ERR_INVALID_ARG, std::string("vt_bvh_build: ") + e.what());
    }
}

int vt_bvh_refit(vt_bvh* bvh, const vt_tri64* tris)
{
    if (!bvh) return fail(VT_ERR_INVALID_ARG, "vt_bvh_refit: bvh is NULL");
    return bvh_refit(bvh->bvh, tris);
}

void vt_bvh_free(vt_bvh* bvh) { delete bvh; }
uint32_t vt_bvh_node_count(const vt_bvh* b) { return b ? uint32_t(b->bvh.nodes.size()) : 0; }
uint32_t vt_bvh_prim_count(const vt_bvh* b) { return b ? uint32_t(b->bvh.prim_indices.size()) : 0; }
const vt_bvh_node* vt_bvh_nodes(const vt_bvh* b) { return b && !b->bvh.nodes.empty() ? b->bvh.nodes.data() : nullptr; }
const uint32_t* vt_bvh_prim_indices(const vt_bvh* b)
{
    return b && !b->bvh.prim_indices.empty() ? b->bvh.prim_indices.data() : nullptr;
}

int vt_scene_linearise(const vt_bvh* bvh, const vt_tri64* tris, vt_host_scene** out)
{
    if (!out || !bvh) return fail(VT_ERR_INVALID_ARG, "vt_scene_linearise: NULL argument");
    *out = nullptr;
    try {
        vt_host_scene* hs = new vt_host_scene();
        int rc = scene_linearise(bvh->bvh, tris, hs->hs);
        if (rc != VT_OK) { delete hs; return rc; }
        *out = hs;
        return VT_OK;
    } catch (const std::bad_alloc&) {
        return fail(VT_ERR_NOMEM, "vt_scene_linearise: out of host memory");
    } catch (const std::exception& e) {
        return fail(VT_ERR_INVALID_ARG, std::string("vt_scene_linearise: ") + e.what());
    }
}

void vt_host_scene_free(vt_host_scene* hs) { delete hs; }
uint32_t vt_host_scene_pair_count(const vt_host_scene* hs) { return hs ? uint32_t(hs->hs.pairs.size()) : 0; }
uint32_t vt_host_scene_tri_count(const vt_host_scene* hs) { return hs ? uint32_t(hs->hs.tris.size()) : 0; }
uint32_t vt_host_scene_max_depth(const vt_host_scene* hs) { return hs ? hs->hs.max_depth : 0; }
uint32_t vt_host_scene_root_leaf_count(const vt_host_scene* hs) { return hs ? hs->hs.root_leaf_count : 0; }
const vt_node_pair* vt_host_scene_pairs(const vt_host_scene* hs)
{
    return hs && !hs->hs.pairs.empty() ? hs->hs.pairs.data() : nullptr;
}
const vt_tri64* vt_host_scene_tris(const vt_host_scene* hs)
{
    return hs && !hs->hs.tris.empty() ? hs->hs.tris.data() : nullptr;
}

} // extern "C"

extern "C" {

int vt_test_fail_alloc(uint64_t k)
{
    if (!vt::alloc_hooks_on()) return vt::fail(VT_ERR_UNSUPPORTED, "vt_test_fail_alloc: test hooks are off (VT_ENABLE_TEST_HOOKS=1)");
    vt::g_alloc_count.store(0, std::memory_order_release);
    vt::g_alloc_fail_at.store(k, std::memory_order_release);
    return VT_OK;
}

uint64_t vt_test_alloc_count(void) { return vt::alloc_hooks_on() ? vt::g_alloc_count.load(std::memory_order_acquire) : 0; }

int vt_test_fail_hip(uint64_t k)
{
    if (!vt::alloc_hooks_on()) return vt::fail(VT_ERR_UNSUPPORTED, "vt_test_fail_hip: test hooks are off (VT_ENABLE_TEST_HOOKS=1)");
    vt::g_hip_count.store(0, std::memory_order_release);
    vt::g_hip_fail_at.store(k, std::memory_order_release);
    return VT_OK;
}

uint64_t vt_test_hip_count(void) { return vt::alloc_hooks_on() ? vt::g_hip_count.load(std::memory_order_acquire) : 0; }

} // extern "C"
