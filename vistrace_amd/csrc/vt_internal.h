// vt_internal.h -- shared declarations of the host-side pieces behind include/vistrace_hip.h
#pragma once

#include "vistrace_hip.h"

#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

static_assert(sizeof(vt_ray) == 32, "vt_ray must be 32 bytes");
static_assert(sizeof(vt_hit) == 16, "vt_hit must be 16 bytes");
static_assert(sizeof(vt_bvh_node) == 32, "vt_bvh_node must be 32 bytes");
static_assert(sizeof(vt_node_pair) == 64, "vt_node_pair must be 64 bytes");
static_assert(sizeof(vt_tri64) == 64, "vt_tri64 must be 64 bytes");
static_assert(sizeof(vt_hit_attrs) == 64, "vt_hit_attrs must be 64 bytes");
static_assert(sizeof(vt_tri_attribs) == 48, "vt_tri_attribs must be 48 bytes");
static_assert(sizeof(vt_hit_shade) == 32, "vt_hit_shade must be 32 bytes");

// Mutation testing (scripts/mutants.sh for the device kernels, scripts/mutants_host.sh for the host walk): with -DVT_MUTANT=<k> ONE
// site answers `wrong`; the product build never defines VT_MUTANT, and every site is then the token `right` after preprocessing.
#ifndef VT_MUT
#ifdef VT_MUTANT
#define VT_MUT(k, wrong, right) ((VT_MUTANT == (k)) ? (wrong) : (right))
#else
#define VT_MUT(k, wrong, right) (right)
#endif
#endif

namespace vt {

// v1-layout tree (what bvh::Bvh<float> holds in the reference: source/objects/AccelStruct.h:67)
struct Bvh {
    std::vector<vt_bvh_node> nodes;        // nodes[0] = root; empty <=> no triangles
    std::vector<uint32_t>    prim_indices; // leaf slot -> original triangle index
};

// device-ready scene
struct HostScene {
    std::vector<vt_node_pair> pairs;   // depth-first order; pairs[0] = the root's children
    std::vector<vt_tri64>     tris;    // leaf order
    std::vector<uint32_t>     pair_depth; // depth of every pair (root's pair = 1), for level-wise refit
    uint32_t max_depth       = 0;      // deepest inner level = stack entries a ray can need
    uint32_t root_leaf_count = 0;      // != 0: the root itself is a leaf over tris[0..count)
    // host walk only (host_walk.cpp): alpha-test side data, as vt_scene_set_tri_attribs / vt_scene_set_alpha hold it on the device
    bool                           has_alpha = false;   // a triangle carries VT_TRI_ALPHATEST
    std::vector<vt_tri_attribs>    attribs;             // original triangle order
    std::vector<vt_alpha_material> alpha_mats;
    std::vector<uint8_t>           alpha_texels;
};

int  bvh_build(const vt_tri64* tris, uint32_t n, int nthreads, int builder, Bvh& out);
int  scene_linearise(const Bvh& bvh, const vt_tri64* tris, HostScene& out);
int  bvh_refit(Bvh& bvh, const vt_tri64* tris);
void tri_setup(const float p0[3], const float p1[3], const float p2[3], uint32_t prim,
               uint32_t flags, vt_tri64& out);

// memcpy split over a few host threads (staging copies of the host-buffer entry points)
void parallel_copy(void* dst, const void* src, size_t bytes);
// ... of n rays, returning the first ray whose range fails the checks of AccelStruct::Traverse (n = none)
uint64_t parallel_copy_checked(vt_ray* dst, const void* src, uint64_t n);

// value of a test-hook environment variable, NULL unless VT_ENABLE_TEST_HOOKS=1 (announced on stderr once per hook)
const char* test_hook(const char* name);

// fault injection (VT_TEST_FAIL_ALLOC / vt_test_fail_alloc, dead without VT_ENABLE_TEST_HOOKS=1): counts one allocation attempt of
// the library; true = this is the one that must fail.  Called by dev_malloc / pinned_malloc (engine_internal.h) only.
bool test_alloc_fails();
// ... and on every other HIP call the library checks (VT_TEST_FAIL_HIP / vt_test_fail_hip): counts one pass through a VT_HIP site
// (engine_internal.h); true = this one reports hipErrorUnknown and its call is NOT made.  Before it answers true it drains every
// device (the function registered with set_test_drain: engine.hip's), so that what the failing function leaves behind -- buffers
// it frees, events it never recorded -- meets an idle device, as after a real sticky error.
bool test_hip_fails();
void set_test_drain(void (*drain)());

void set_error(const std::string& msg);
int  fail(int code, const std::string& msg);

} // namespace vt

struct vt_bvh        { vt::Bvh bvh; };
// `stale` is shared with every vt_scene uploaded from this host scene: a device-side refit / skin refit sets it, the host walk
// refuses to answer from records the device no longer holds until vt_host_scene_sync has fetched the new ones
struct vt_host_scene {
    vt::HostScene hs;
    std::shared_ptr<std::atomic<int>> stale = std::make_shared<std::atomic<int>>(0);
};
