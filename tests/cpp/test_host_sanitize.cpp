// test_host_sanitize.cpp -- host-only pieces of the C ABI (triangle set-up, PLOC build, leaf collapse,
// linearise) under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on this pool).
// Built by `make -C tests/cpp sanitize`; exits non-zero on any sanitizer report or failed check.
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "vistrace_hip.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { ++fails; std::printf("FAIL line %d: %s\n", __LINE__, #c); } } while (0)

static void run(uint32_t n, unsigned seed, bool clustered)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(-100.f, 100.f), S(-1.f, 1.f);
    std::vector<float> verts(size_t(n) * 9);
    for (uint32_t i = 0; i < n; ++i) {
        float c[3] = {U(rng), U(rng), clustered ? 0.f : U(rng)};
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) verts[size_t(i) * 9 + v * 3 + k] = c[k] + (clustered && k == 2 ? 0.f : S(rng));
    }
    std::vector<uint8_t> flags(n);
    for (uint32_t i = 0; i < n; ++i) flags[i] = uint8_t(i & 1);
    std::vector<vt_tri64> recs(n);
    CHECK(vt_tris_setup(verts.data(), flags.data(), n, recs.data()) == VT_OK);
    // every builder: the default (task-parallel binned SAH), the reference's algorithm (PLOC + leaf collapse), SAH + re-insertion
    for (int builder : {int(VT_BUILDER_BINNED_SAH), int(VT_BUILDER_PLOC), int(VT_BUILDER_BINNED_SAH_REFINED)}) {
        vt_bvh* b2 = nullptr;
        CHECK(vt_bvh_build_ex(recs.data(), n, 3, builder, &b2) == VT_OK && b2);
        CHECK(vt_bvh_prim_count(b2) == n);
        vt_host_scene* h2 = nullptr;
        CHECK(vt_scene_linearise(b2, recs.data(), &h2) == VT_OK && h2 && vt_host_scene_tri_count(h2) == n);
        vt_host_scene_free(h2);
        vt_bvh_free(b2);
    }
    vt_bvh* bvh = nullptr;
    CHECK(vt_bvh_build(recs.data(), n, 3, &bvh) == VT_OK && bvh);
    vt_host_scene* hs = nullptr;
    CHECK(vt_scene_linearise(bvh, recs.data(), &hs) == VT_OK && hs);
    CHECK(vt_host_scene_tri_count(hs) == n);
    CHECK(vt_bvh_prim_count(bvh) == n);
    if (n > 1) CHECK(vt_host_scene_pair_count(hs) == (vt_bvh_node_count(bvh) - 1) / 2 || vt_host_scene_root_leaf_count(hs) == n);
    // every triangle appears exactly once in leaf order
    std::vector<int> seen(n, 0);
    const vt_tri64* lt = vt_host_scene_tris(hs);
    for (uint32_t i = 0; i < n; ++i) { CHECK(lt[i].prim < n); if (lt[i].prim < n) seen[lt[i].prim]++; }
    for (uint32_t i = 0; i < n; ++i) CHECK(seen[i] == 1);
    // the host walk (single-ray latency path) over the same scene: brute force agrees on t, and the walk stays in bounds
    if (n > 0) {
        std::vector<vt_ray> rays(256);
        for (auto& r : rays) r = vt_ray{{U(rng), U(rng), 150.f}, {S(rng) * 0.3f, S(rng) * 0.3f, -1.f}, 0.f, 3.0e38f};
        std::vector<vt_hit> hits(rays.size());
        std::vector<uint8_t> occ(rays.size());
        CHECK(vt_host_scene_trace_closest(hs, rays.data(), rays.size(), hits.data()) == VT_OK);
        CHECK(vt_host_scene_trace_any(hs, rays.data(), rays.size(), occ.data()) == VT_OK);
        for (size_t i = 0; i < rays.size(); ++i) {
            CHECK((hits[i].prim != VT_MISS) == (occ[i] != 0));
            CHECK(hits[i].prim == VT_MISS || hits[i].prim < n);
        }
    }
    vt_host_scene_free(hs);
    vt_bvh_free(bvh);
}

// shard bounds of the multi-GPU path: contiguous, disjoint, cover [0, n), shard g starts at g * capacity
static void shards()
{
    for (uint64_t n : {0ull, 1ull, 63ull, 64ull, 65ull, 1000ull, 1048576ull, 16777216ull, 134217728ull, 134217729ull})
        for (int ndev : {1, 2, 3, 4, 7, 8}) {
            const uint64_t cap = vt_shard_capacity(n, ndev);
            CHECK(cap % 64 == 0 && cap * uint64_t(ndev) >= n);
            uint64_t expect = 0;
            for (int g = 0; g < ndev; ++g) {
                uint64_t lo = 1, hi = 0;
                vt_shard_bounds(n, ndev, g, &lo, &hi);
                CHECK(lo == expect && hi >= lo && hi - lo <= cap);
                CHECK(lo == (cap * uint64_t(g) < n ? cap * uint64_t(g) : n));
                expect = hi;
            }
            CHECK(expect == n);
            uint64_t lo = 1, hi = 0;
            vt_shard_bounds(n, ndev, ndev, &lo, &hi);       // out of range: empty
            CHECK(lo == hi);
            vt_shard_bounds(n, ndev, -1, &lo, &hi);
            CHECK(lo == hi);
        }
    CHECK(vt_shard_capacity(100, 0) == 0);
}

// the staging copy of the batch boundary (vt_batch_trace_closest_ex, VT_BATCH_CHECK_RANGES): copies n rays from a source of ANY
// alignment and names the first ray whose range fails the checks of AccelStruct::Traverse
namespace vt { uint64_t parallel_copy_checked(vt_ray* dst, const void* src, uint64_t n); void parallel_copy(void* dst, const void* src, size_t bytes); }
static void staging_copy()
{
    for (uint64_t n : {0ull, 1ull, 5ull, 32768ull, 32769ull, 200000ull}) {
        std::vector<char> raw(n * sizeof(vt_ray) + 3);
        char* src = raw.data() + 3;                              // misaligned on purpose (a Lua string carries no alignment promise)
        std::vector<vt_ray> rays(n), dst(n), plain(n);
        for (uint64_t i = 0; i < n; ++i) rays[i] = vt_ray{{float(i), 1.f, 2.f}, {0.f, 0.f, 1.f}, float(i % 7), float(i % 7) + 1.f};
        if (n) std::memcpy(src, rays.data(), n * sizeof(vt_ray));
        CHECK(vt::parallel_copy_checked(dst.data(), src, n) == n);
        CHECK(n == 0 || std::memcmp(dst.data(), rays.data(), n * sizeof(vt_ray)) == 0);
        vt::parallel_copy(plain.data(), src, n * sizeof(vt_ray));
        CHECK(n == 0 || std::memcmp(plain.data(), rays.data(), n * sizeof(vt_ray)) == 0);
        for (uint64_t bad : {uint64_t(0), n / 2, n ? n - 1 : 0}) {
            if (bad >= n) continue;
            for (int rule = 0; rule < 3; ++rule) {
                vt_ray r = rays[bad];
                if (rule == 0) r.tmin = -0.5f; else if (rule == 1) r.tmax = r.tmin; else r.tmax = std::nanf("");   // NaN ranges pass, as in the reference
                std::memcpy(src + bad * sizeof(vt_ray), &r, sizeof(r));
                const uint64_t got = vt::parallel_copy_checked(dst.data(), src, n);
                CHECK(got == (rule == 2 ? n : bad));
                std::memcpy(src + bad * sizeof(vt_ray), &rays[bad], sizeof(vt_ray));
            }
        }
        if (n > 40000) {                                         // two offenders in different pieces: the FIRST is named
            vt_ray r = rays[100]; r.tmin = -1.f;
            std::memcpy(src + 39000 * sizeof(vt_ray), &r, sizeof(r));
            std::memcpy(src + 100 * sizeof(vt_ray), &r, sizeof(r));
            CHECK(vt::parallel_copy_checked(dst.data(), src, n) == 100);
        }
    }
}

int main()
{
    staging_copy();
    for (uint32_t n : {0u, 1u, 2u, 3u, 17u, 1000u, 20000u, 70000u}) { run(n, 7 + n, false); run(n, 11 + n, true); }
    shards();
    vt_bvh* b = nullptr;
    CHECK(vt_bvh_build(nullptr, 5, 0, &b) != VT_OK && vt_last_error()[0] != 0);   // NULL input is an error, not a crash
    std::printf("host sanitize: %d failed\n", fails);
    return fails ? 1 : 0;
}
