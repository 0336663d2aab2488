// test_host_sanitize.cpp -- host-only pieces of the C ABI (triangle set-up, PLOC build, leaf collapse,
// linearise) under AddressSanitizer + UBSan on the CPU (GPU sanitizers are not available on this pool).
// Built by `make -C tests/cpp sanitize`; exits non-zero on any sanitizer report or failed check.
#include <cstdio>
#include <cstdlib>
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
    vt_host_scene_free(hs);
    vt_bvh_free(bvh);
}

int main()
{
    for (uint32_t n : {0u, 1u, 2u, 3u, 17u, 1000u, 20000u}) { run(n, 7 + n, false); run(n, 11 + n, true); }
    vt_bvh* b = nullptr;
    CHECK(vt_bvh_build(nullptr, 5, 0, &b) != VT_OK && vt_last_error()[0] != 0);   // NULL input is an error, not a crash
    std::printf("host sanitize: %d failed\n", fails);
    return fails ? 1 : 0;
}
