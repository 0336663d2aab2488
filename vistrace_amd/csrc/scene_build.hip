// scene_build.hip -- Rebuild's device half: from the CPU-built tree to the records the traversal kernels read.
//
// The reference constructs its intersector / traverser over the finished tree (source/objects/AccelStruct.cpp:772-773); this
// project re-packs the tree for the device first: inner nodes become 64-B sibling-pair records in depth-first (left-first) order,
// triangles are shuffled into leaf order, and two index tables are derived (triangle -> slot for refits, pairs by depth for the
// level-wise refit).  Round 1-4 did that on the host (linearise.cpp: a serial walk, 85-100 ms per million triangles) and uploaded
// the result; here the walk is replaced by five data-parallel kernels over the v1 node array, so a Rebuild's upload step is three
// plain copies (nodes, primitive indices, triangle records as the builder saw them) plus well under a millisecond of device work:
//
//   lin_parents   node i -> parent of its two children
//   lin_counts    bottom-up: inner nodes and triangles below every node (each leaf walks up; the second arrival at a node
//                 continues, and takes the first one's counts out of the 64-bit atomic they both add to: no fences)
//   lin_offsets   every node walks up to the root and sums what precedes it in the depth-first order: its pair index, the
//                 first triangle slot of its pair / of its leaf, its depth
//   lin_emit      inner node -> its pair record (children's `first` re-targeted); leaf -> its triangle records + prim_to_slot
//   radix sort    pairs by (max_depth - depth), stable: the level lists of the refit, deepest level first
//
// The result is BYTE-EQUAL to vt_scene_linearise + vt_scene_upload (tests/test_gpu_parity.py::test_device_linearise_*): same
// numbering (a pair's leaf children are emitted when the pair is numbered, left before right; then the left subtree, then the
// right), same level lists (ascending pair index inside a level).  vt_host_scene_download brings the records back for the callers
// that walk single rays on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <string.h>              // rocprim's texture iterator calls the global memset from host code
#include <rocprim/rocprim.hpp>

#include "engine_internal.h"

using namespace vt;

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kErrChild = 1, kErrDepth = 2, kErrPrim = 4;
constexpr int kMaxWalk = 1 << 16;             // steps a walk to the root may take before the tree is declared malformed

struct LinArgs {
    const vt_bvh_node* nodes;  uint32_t n_nodes;
    const uint32_t* prim_indices; uint32_t n_prims;
    const vt_tri64* tris_in;   uint32_t ntris;
    uint32_t *parent, *cnt, *tcount, *pidx, *tbase, *depth;
    unsigned long long* acc;   // per node: packed counts of the subtree walk(s) that have arrived (lin_counts)
    uint32_t* status;          // [0] error bits, [1] deepest pair, [2] a triangle carries VT_TRI_ALPHATEST
};

__global__ __launch_bounds__(256) void lin_alpha_flag(const vt_tri64* tris, uint32_t n, uint32_t* status)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && (tris[i].flags & VT_TRI_ALPHATEST)) status[2] = 1;
}

// parent[] arrives filled with kNone (a node no inner node points to stays a root of its own walk and the totals below the real
// root come out short: "malformed tree").  A child index that does not lie behind its parent, lies outside the array or is claimed
// by two parents is an error; the kernels behind this one do nothing once an error bit is up, so no walk ever follows a bad index.
__global__ __launch_bounds__(256) void lin_parents(LinArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_nodes) return;
    const vt_bvh_node me = a.nodes[i];
    if (me.prim_count != 0) {
        if (uint64_t(me.first) + me.prim_count > a.n_prims) atomicOr(&a.status[0], kErrPrim);
        return;
    }
    // children sit side by side behind their parent (v1 layout; both builders emit children at higher indices)
    if (me.first <= i || uint64_t(me.first) + 1 >= a.n_nodes) { atomicOr(&a.status[0], kErrChild); return; }
    if (atomicCAS(&a.parent[me.first], kNone, i) != kNone || atomicCAS(&a.parent[me.first + 1], kNone, i) != kNone)
        atomicOr(&a.status[0], kErrChild);
}

// Bottom-up counts without fences: the two walks that meet at a node exchange their counts THROUGH the atomic itself.  A walk adds
// its packed (inner nodes << 32 | triangles) to the node's 64-bit accumulator; the first to arrive sees 0 and ends, the second sees
// the sibling subtree's counts in the returned value, stores the node's totals and carries them upwards.  (A first version wrote
// the counts with plain stores and published them with __threadfence + a flag: an agent-scope release / acquire writes back and
// invalidates the XCD's L2 on this part -- 4.3 ms for 1.15 M nodes, 49 ms for 11.7 M; this form: see profiles/r5/notes.md.)
__global__ __launch_bounds__(256) void lin_counts(LinArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_nodes || a.status[0] != 0) return;        // (an error bit of lin_parents: the parent links are not to be trusted)
    const uint32_t pc = a.nodes[i].prim_count;
    if (pc == 0) return;                                   // the leaves start the walks
    unsigned long long mine = pc;                          // a leaf: no inner nodes, pc triangles (never 0: a packed value marks an arrival)
    uint32_t x = i;
    for (int guard = 0; guard < kMaxWalk; ++guard) {
        const uint32_t p = a.parent[x];
        if (p == kNone) return;
        const unsigned long long other = atomicAdd(&a.acc[p], mine);
        if (other == 0) return;                            // first at this node: the sibling's walk will carry on
        mine += other + (1ull << 32);                      // both subtrees + this inner node
        a.cnt[p] = uint32_t(mine >> 32);
        a.tcount[p] = uint32_t(mine);
        x = p;
    }
    atomicOr(&a.status[0], kErrDepth);
}

// Depth-first numbering without a walk from the root: node x below pair P (x = P's left or right child) comes behind
//   P itself, P's leaf children (emitted when P is numbered), and -- if x is the right child -- the whole left subtree.
__global__ __launch_bounds__(256) void lin_offsets(LinArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_nodes || a.status[0] != 0) return;
    const bool leaf = a.nodes[i].prim_count != 0;
    uint32_t x = i, ap = 0, at = 0, d = 0;
    bool first = true;
    for (;;) {
        const uint32_t P = a.parent[x];
        if (P == kNone) break;
        const uint32_t L = a.nodes[P].first;
        const bool right = x != L;
        const uint32_t pcL = a.nodes[L].prim_count, pcR = a.nodes[L + 1].prim_count;
        if (first && leaf) {
            at += (right && pcL != 0) ? pcL : 0u;          // a leaf sits in its pair's own block: behind the left sibling leaf
        } else {
            ap += 1u + ((VT_MUT(31, true, right) && pcL == 0) ? a.cnt[L] : 0u);      // (VT_MUT: mutation sites, vt_internal.h)
            at += pcL + pcR + ((right && pcL == 0) ? a.tcount[L] : 0u);   // pcX = 0 for an inner child: only leaf children count here
        }
        first = false;
        x = P;
        if (++d > uint32_t(kMaxWalk)) { atomicOr(&a.status[0], kErrDepth); return; }
    }
    a.tbase[i] = at;
    if (!leaf) {
        a.pidx[i] = ap;
        a.depth[i] = d + 1;                                // the root's pair has depth 1 (linearise.cpp)
        atomicMax(&a.status[1], d + 1);
    }
}

__global__ __launch_bounds__(256) void lin_emit(LinArgs a, vt_node_pair* pairs, vt_tri64* tris_out, uint32_t* prim_to_slot, uint32_t* keys,
                                                uint32_t* vals, uint32_t max_depth)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_nodes) return;
    const vt_bvh_node me = a.nodes[i];
    if (me.prim_count == 0) {
        const uint32_t p = a.pidx[i];
        vt_node_pair rec;
        for (int side = 0; side < 2; ++side) {
            const uint32_t c = me.first + uint32_t(side);
            vt_bvh_node ch = a.nodes[c];
            ch.first = ch.prim_count != 0 ? a.tbase[c] : a.pidx[c];
            rec.child[side] = ch;
        }
        pairs[p] = rec;
        keys[p] = max_depth - a.depth[i];                  // deepest level first
        vals[p] = p;
        return;
    }
    const uint32_t slot = a.tbase[i];
    for (uint32_t q = 0; q < me.prim_count; ++q) {
        const uint32_t idx = a.prim_indices[me.first + q];
        if (idx >= a.ntris) { atomicOr(&a.status[0], kErrPrim); continue; }
        vt_tri64 t = a.tris_in[idx];
        t.prim = idx;
        tris_out[slot + q] = t;
        prim_to_slot[idx] = slot + q;
    }
}

// where each level starts in the sorted keys (every level 0 .. max_depth-1 has at least one pair)
__global__ __launch_bounds__(256) void level_starts(const uint32_t* sorted_keys, uint32_t n, uint32_t* begin)
{
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    if (j == 0 || sorted_keys[j] != sorted_keys[j - 1]) begin[sorted_keys[j]] = VT_MUT(32, j + (j != 0), j);
}

// classic path (records linearised on the host): the two index tables from what is already on the device
__global__ __launch_bounds__(256) void slots_from_records(const vt_tri64* tris, uint32_t n, uint32_t* prim_to_slot, uint32_t* status)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t prim = tris[i].prim;
    if (prim >= n) { atomicOr(&status[0], kErrPrim); return; }
    prim_to_slot[prim] = i;
}

__global__ __launch_bounds__(256) void keys_from_depth(const uint32_t* depth, uint32_t n, uint32_t max_depth, uint32_t* keys, uint32_t* vals)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n) return;
    keys[p] = max_depth - depth[p];
    vals[p] = p;
}

constexpr uint32_t kLevelWord = 64, kRootWord = 16;   // words of the pinned read-back block: level starts from 64 on, the root pair at 16
constexpr unsigned kMaxDepthBits = 17;                // kMaxWalk + 1 levels at most
inline unsigned key_bits(uint32_t max_depth)          // radix-sort bits that cover the keys 0 .. max_depth - 1
{
    unsigned b = 1;
    while (b < 32 && (uint64_t(1) << b) < max_depth) ++b;
    return std::max(b, 8u);
}
inline size_t al256(size_t b) { return (b + 255) & ~size_t(255); }
inline uint32_t blocks_for(uint64_t n) { return uint32_t((n + 255) / 256); }
inline double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// pairs sorted by level (keys = max_depth - depth, vals = pair index: both in scratch) -> s->d_level_pairs, s->level_begin
int build_level_lists(vt_scene* s, uint32_t* keys, uint32_t* vals, uint32_t* keys_out, uint32_t* begin, void* sort_tmp, size_t sort_tmp_bytes,
                      uint32_t* h_begin /* pinned, max_depth entries */)
{
    vt_engine* e = s->engine;
    const uint32_t np = s->npairs, md = s->max_depth;
    size_t need = sort_tmp_bytes;
    VT_HIP(rocprim::radix_sort_pairs(sort_tmp, need, keys, keys_out, vals, s->d_level_pairs, np, 0, key_bits(md), e->stream));
    hipLaunchKernelGGL(level_starts, dim3(blocks_for(np)), dim3(256), 0, e->stream, keys_out, np, begin);
    VT_HIP(hipGetLastError());
    VT_HIP(hipMemcpyAsync(h_begin, begin, size_t(md) * 4, hipMemcpyDeviceToHost, e->stream));
    return VT_OK;
}

// sort scratch for np pairs, whatever the tree's depth turns out to be (8 key bits cover 255 levels, kMaxDepthBits the deepest
// tree the walks accept)
size_t sort_tmp_bytes_for(uint32_t np)
{
    size_t need = 0, need_deep = 0;
    uint32_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, need, nul, nul, nul, nul, np, 0, 8, hipStream_t(nullptr));
    (void)rocprim::radix_sort_pairs(nullptr, need_deep, nul, nul, nul, nul, np, 0, kMaxDepthBits, hipStream_t(nullptr));
    return al256(std::max(need, need_deep)) + 256;
}

// pinned read-back block of the engine: 64 words of status / counts / the root pair, then one word per tree level
int ensure_readback(vt_engine* e, uint32_t levels)
{
    const size_t need = std::max<size_t>(4096, (size_t(kLevelWord) + levels + 1) * 4);
    if (e->h_build && e->h_build_bytes >= need) return VT_OK;
    if (e->h_build) { (void)hipHostFree(e->h_build); e->h_build = nullptr; e->h_build_bytes = 0; }
    VT_HIP(pinned_malloc(reinterpret_cast<void**>(&e->h_build), need));
    e->h_build_bytes = need;
    return VT_OK;
}

} // namespace

namespace vt {

// Shared by vt_scene_upload (host-linearised records already copied into s->d_records) : prim_to_slot and the level lists on the
// device.  h_pair_depth: depth of every pair (HostScene::pair_depth).
int scene_index_tables(vt_scene* s, const uint32_t* h_pair_depth)
{
    vt_engine* e = s->engine;
    const uint32_t np = s->npairs, nt = s->ntris, md = s->max_depth;
    if (md > uint32_t(kMaxWalk)) return fail(VT_ERR_UNSUPPORTED, "vt_scene_upload: the tree is deeper than 65536 levels");
    const size_t sort_b = np ? sort_tmp_bytes_for(np) : 0;
    const size_t need = 5 * al256(size_t(np) * 4) + al256(size_t(md + 1) * 4) + sort_b + 512;
    int rc = ensure_bytes(&e->d_build, &e->d_build_bytes, need);
    if (rc == VT_OK) rc = ensure_readback(e, md);
    if (rc != VT_OK) return rc;
    char* base = static_cast<char*>(e->d_build);
    uint32_t* status = reinterpret_cast<uint32_t*>(base);
    char* cur = base + 256;
    auto take = [&](size_t bytes) { char* p = cur; cur += al256(bytes); return p; };
    uint32_t* depth = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* keys = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* vals = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* begin = reinterpret_cast<uint32_t*>(take(size_t(md + 1) * 4));
    void* sort_tmp = take(sort_b);
    VT_HIP(hipMemsetAsync(status, 0, 256, e->stream));
    if (nt) {
        VT_HIP(dev_malloc(reinterpret_cast<void**>(&s->d_prim_to_slot), size_t(nt) * 4));
        s->bytes += size_t(nt) * 4;
        hipLaunchKernelGGL(slots_from_records, dim3(blocks_for(nt)), dim3(256), 0, e->stream, s->d_tris, nt, s->d_prim_to_slot, status);
        VT_HIP(hipGetLastError());
    }
    uint32_t* h = reinterpret_cast<uint32_t*>(e->h_build);
    if (np) {
        VT_HIP(dev_malloc(reinterpret_cast<void**>(&s->d_level_pairs), size_t(np) * 4));
        s->bytes += size_t(np) * 4;
        VT_HIP(hipMemcpyAsync(depth, h_pair_depth, size_t(np) * 4, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(keys_from_depth, dim3(blocks_for(np)), dim3(256), 0, e->stream, depth, np, md, keys, vals);
        VT_HIP(hipGetLastError());
        rc = build_level_lists(s, keys, vals, keys_out, begin, sort_tmp, sort_b, h + kLevelWord);
        if (rc != VT_OK) return rc;
    }
    VT_HIP(hipMemcpyAsync(h, status, 16, hipMemcpyDeviceToHost, e->stream));
    VT_HIP(hipStreamSynchronize(e->stream));
    if (h[0] & kErrPrim) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload: bad prim index");
    s->level_begin.clear();
    if (np) {
        for (uint32_t k = 0; k < md; ++k) s->level_begin.push_back(h[kLevelWord + k]);
        s->level_begin.push_back(np);
    }
    return VT_OK;
}

} // namespace vt

extern "C" {

int vt_scene_upload_tree(vt_engine* e, const vt_bvh* bvhw, const vt_tri64* tris, uint32_t ntris, vt_scene** out)
{
    if (!e || !bvhw || !out) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: NULL argument");
    *out = nullptr;
    const Bvh& bvh = bvhw->bvh;
    const uint32_t N = uint32_t(bvh.nodes.size()), M = uint32_t(bvh.prim_indices.size());
    if (M != ntris) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: ntris differs from the tree's primitive count");
    if (N != 0 && !tris) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: tris is NULL");
    // an empty tree and a tree whose root is a leaf have nothing to number: the host path serves them (a handful of records)
    if (N == 0 || bvh.nodes[0].prim_count != 0) {
        vt_host_scene* hs = nullptr;
        int rc = vt_scene_linearise(bvhw, tris, &hs);
        if (rc == VT_OK) rc = vt_scene_upload(e, hs, out);
        vt_host_scene_free(hs);
        return rc;
    }
    if ((N & 1u) == 0) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: malformed tree (even node count)");
    const auto t_begin = std::chrono::steady_clock::now();
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_upload_tree: hipSetDevice failed");
    // (released before the scene goes to the group's other members: a call holds ONE member's lock at a time, or all of them
    // in the fixed order of update_every_member -- never a root's while it asks for a replica's)
    std::unique_lock<std::mutex> host_lock(e->host_mu);
    const uint32_t np = (N - 1) / 2;
    uint64_t gap = 0;                                        // test hook, as vt_scene_upload (include/vistrace_hip.h, "Test hooks")
    if (const char* env = test_hook("VT_TEST_RECORD_GAP")) gap = std::strtoull(env, nullptr, 10) & ~uint64_t(1);
    if (uint64_t(np) + gap + 2 * uint64_t(ntris) + 4 >= 0xFFFFFFFFull) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: scene too large");

    // ---- staging block: the tree as the builder left it, per-node scratch, sort scratch ---------------------------------------
    const size_t sort_b = sort_tmp_bytes_for(np);
    const size_t begin_b = (size_t(std::min<uint32_t>(np, uint32_t(kMaxWalk) + 1u)) + 1) * 4;      // one word per tree level, whatever the depth
    const size_t need = 512 + al256(size_t(N) * 32) + al256(size_t(M) * 4) + al256(size_t(ntris) * 64) + 6 * al256(size_t(N) * 4) + al256(size_t(N) * 8) +
                        3 * al256(size_t(np) * 4) + al256(begin_b) + sort_b;
    int rc = ensure_bytes(&e->d_build, &e->d_build_bytes, need);
    if (rc == VT_OK) rc = ensure_readback(e, 0);
    if (rc != VT_OK) return rc;
    char* base = static_cast<char*>(e->d_build);
    char* cur = base + 512;
    auto take = [&](size_t bytes) { char* p = cur; cur += al256(bytes); return p; };
    LinArgs a{};
    a.status = reinterpret_cast<uint32_t*>(base);
    vt_bvh_node* d_nodes = reinterpret_cast<vt_bvh_node*>(take(size_t(N) * 32));
    uint32_t* d_prims = reinterpret_cast<uint32_t*>(take(size_t(M) * 4));
    vt_tri64* d_tris_in = reinterpret_cast<vt_tri64*>(take(size_t(ntris) * 64));
    a.nodes = d_nodes; a.n_nodes = N; a.prim_indices = d_prims; a.n_prims = M; a.tris_in = d_tris_in; a.ntris = ntris;
    uint32_t** per_node[] = {&a.parent, &a.cnt, &a.tcount, &a.pidx, &a.tbase, &a.depth};
    for (uint32_t** pn : per_node) *pn = reinterpret_cast<uint32_t*>(take(size_t(N) * 4));
    a.acc = reinterpret_cast<unsigned long long*>(take(size_t(N) * 8));
    uint32_t* keys = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* vals = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(take(size_t(np) * 4));
    uint32_t* begin = reinterpret_cast<uint32_t*>(take(begin_b));
    void* sort_tmp = take(sort_b);
    const double alloc_ms = ms_since(t_begin);

    // ---- three plain copies, then the numbering ---------------------------------------------------------------------------------
    const auto t_copy = std::chrono::steady_clock::now();
    hipStream_t st = e->stream;
    VT_HIP(hipMemsetAsync(a.status, 0, 512, st));
    VT_HIP(hipMemsetAsync(a.acc, 0, size_t(N) * 8, st));
    VT_HIP(hipMemsetAsync(a.parent, 0xFF, size_t(N) * 4, st));                  // kNone: lin_parents claims children with a CAS
    VT_HIP(hipMemcpyAsync(d_nodes, bvh.nodes.data(), size_t(N) * 32, hipMemcpyHostToDevice, st));
    VT_HIP(hipMemcpyAsync(d_prims, bvh.prim_indices.data(), size_t(M) * 4, hipMemcpyHostToDevice, st));
    VT_HIP(hipMemcpyAsync(d_tris_in, tris, size_t(ntris) * 64, hipMemcpyHostToDevice, st));
    const double copy_issue_ms = ms_since(t_copy);
    const auto t_kern = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(lin_alpha_flag, dim3(blocks_for(ntris)), dim3(256), 0, st, d_tris_in, ntris, a.status);
    hipLaunchKernelGGL(lin_parents, dim3(blocks_for(N)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(lin_counts, dim3(blocks_for(N)), dim3(256), 0, st, a);
    hipLaunchKernelGGL(lin_offsets, dim3(blocks_for(N)), dim3(256), 0, st, a);
    VT_HIP(hipGetLastError());
    uint32_t* h = reinterpret_cast<uint32_t*>(e->h_build);
    VT_HIP(hipMemcpyAsync(h, a.status, 16, hipMemcpyDeviceToHost, st));
    VT_HIP(hipMemcpyAsync(h + 4, a.cnt, 4, hipMemcpyDeviceToHost, st));         // inner nodes and triangles below the root
    VT_HIP(hipMemcpyAsync(h + 5, a.tcount, 4, hipMemcpyDeviceToHost, st));
    VT_HIP(hipStreamSynchronize(st));
    const uint32_t err = h[0], max_depth = h[1];
    const bool has_alpha = h[2] != 0;
    if (err != 0 || h[4] != np || h[5] != ntris || max_depth == 0)
        return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: malformed tree");
    rc = ensure_readback(e, max_depth);          // (h is not read again before the level starts have come back)
    if (rc != VT_OK) return rc;
    h = reinterpret_cast<uint32_t*>(e->h_build);

    // ---- the scene: records (pairs | triangles | room for AlphaRecs), index tables ---------------------------------------------
    vt_scene* s = new vt_scene();
    s->engine = e;
    s->has_alpha = has_alpha;
    s->npairs = np;
    s->ntris = ntris;
    s->max_depth = max_depth;
    s->root_leaf_count = 0;
    s->tri_base = ((np + 1u) & ~1u) + uint32_t(gap);
    if (has_alpha) s->alpha_base = (s->tri_base + ntris + 1u) & ~1u;
    s->record_capacity = std::max<size_t>(has_alpha ? size_t(s->alpha_base) + ntris : size_t(s->tri_base) + ntris, 2);
    const size_t rec_bytes = s->record_capacity * 64;
    hipError_t herr = dev_malloc(reinterpret_cast<void**>(&s->d_records), rec_bytes);
    if (herr == hipSuccess) herr = dev_malloc(reinterpret_cast<void**>(&s->d_prim_to_slot), size_t(ntris) * 4);
    if (herr == hipSuccess) herr = dev_malloc(reinterpret_cast<void**>(&s->d_level_pairs), size_t(np) * 4);
    if (herr != hipSuccess) {
        track_scene(e, s);
        vt_scene_free(s);
        return fail(VT_ERR_HIP, std::string("vt_scene_upload_tree: ") + hipGetErrorString(herr));
    }
    track_scene(e, s);
    s->bytes = rec_bytes + size_t(ntris) * 4 + size_t(np) * 4;
    s->d_tris = reinterpret_cast<vt_tri64*>(s->d_records + size_t(s->tri_base) * 64);
    // only what the kernels do not write needs zeros: the padding record between pairs and triangles (+ the test gap), the AlphaRec room
    const size_t pair_end = size_t(np) * 64, tri_off = size_t(s->tri_base) * 64, tri_end = tri_off + size_t(ntris) * 64;
    herr = hipSuccess;
    if (tri_off > pair_end) herr = VT_TRY(hipMemsetAsync(s->d_records + pair_end, 0, tri_off - pair_end, st));
    if (herr == hipSuccess && rec_bytes > tri_end) herr = VT_TRY(hipMemsetAsync(s->d_records + tri_end, 0, rec_bytes - tri_end, st));
    if (herr != hipSuccess) { vt_scene_free(s); return fail(VT_ERR_HIP, std::string("vt_scene_upload_tree: ") + hipGetErrorString(herr)); }
    hipLaunchKernelGGL(lin_emit, dim3(blocks_for(N)), dim3(256), 0, st, a, reinterpret_cast<vt_node_pair*>(s->d_records), s->d_tris,
                       s->d_prim_to_slot, keys, vals, max_depth);
    rc = hipGetLastError() == hipSuccess ? VT_OK : fail(VT_ERR_HIP, "vt_scene_upload_tree: kernel launch failed");
    if (rc == VT_OK) rc = build_level_lists(s, keys, vals, keys_out, begin, sort_tmp, sort_b, h + kLevelWord);
    if (rc == VT_OK) {
        herr = VT_TRY(hipMemcpyAsync(h + 8, a.status, 4, hipMemcpyDeviceToHost, st));
        if (herr == hipSuccess) herr = VT_TRY(hipMemcpyAsync(h + kRootWord, s->d_records, sizeof(vt_node_pair), hipMemcpyDeviceToHost, st));   // the root pair: packet radius
        if (herr == hipSuccess) herr = VT_TRY(hipStreamSynchronize(st));
        if (herr != hipSuccess) rc = fail(VT_ERR_HIP, std::string("vt_scene_upload_tree: ") + hipGetErrorString(herr));
    }
    if (rc == VT_OK && h[8] != 0) rc = fail(VT_ERR_INVALID_ARG, "vt_scene_upload_tree: bad prim index");
    if (rc != VT_OK) { vt_scene_free(s); return rc; }
    for (uint32_t k = 0; k < max_depth; ++k) s->level_begin.push_back(h[kLevelWord + k]);
    s->level_begin.push_back(np);
    vt_node_pair root;
    std::memcpy(&root, h + kRootWord, sizeof(root));
    s->coherent_radius2 = scene_packet_radius2(root);
    s->upload_stats.alloc_ms = float(alloc_ms);
    s->upload_stats.copy_ms = float(copy_issue_ms);
    s->upload_stats.device_ms = float(ms_since(t_kern));
    s->upload_stats.total_ms = float(ms_since(t_begin));
    s->upload_stats.bytes_h2d = uint64_t(N) * 32 + uint64_t(M) * 4 + uint64_t(ntris) * 64;
    s->upload_stats.linearised_on_device = 1;

    host_lock.unlock();      // (tests/fake_group_check.py runs Rebuilds beside refits on a group from two threads)
    // a group's scene lives on every device: the finished records and index tables go from this device to the
    // others as device-to-device copies (engine.hip: scene_replicate) -- the host uploads the tree once
    rc = scene_replicate(s, [&](vt_engine* p, vt_scene** rep) { return vt_scene_upload_tree(p, bvhw, tris, ntris, rep); });
    if (rc != VT_OK) { vt_scene_free(s); return rc; }
    s->upload_stats.total_ms = float(ms_since(t_begin));
    *out = s;
    return VT_OK;
}

int vt_scene_upload_stats(const vt_scene* s, vt_upload_stats* out)
{
    if (!s || !out) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload_stats: NULL argument");
    *out = s->upload_stats;
    return VT_OK;
}

int vt_host_scene_download(vt_scene* s, vt_host_scene** out)
{
    if (!out) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_download: out is NULL");
    *out = nullptr;
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_download: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_download: the scene\'s engine has been closed");
    if (s->poisoned) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_download: the last refit left non-finite triangles; refit with finite data first");
    vt_host_scene* hsw = nullptr;
    try {
        hsw = new vt_host_scene();
        hsw->hs.pairs.resize(s->npairs);
        hsw->hs.tris.resize(s->ntris);
        hsw->hs.pair_depth.resize(s->npairs);
    } catch (const std::bad_alloc&) {
        delete hsw;
        return fail(VT_ERR_NOMEM, "vt_host_scene_download: out of host memory");
    }
    HostScene& hs = hsw->hs;
    hs.max_depth = s->max_depth;
    hs.root_leaf_count = s->root_leaf_count;
    int rc = vt_scene_read_records(s, hs.pairs.data(), hs.tris.data());
    if (rc == VT_OK && s->npairs) {
        // depth of every pair from the level lists: level k (deepest first) holds pairs of depth max_depth - k
        std::vector<uint32_t> order(s->npairs);
        DeviceGuard guard(s->engine->device);
        const hipError_t err = VT_TRY(hipMemcpy(order.data(), s->d_level_pairs, size_t(s->npairs) * 4, hipMemcpyDeviceToHost));
        if (err != hipSuccess) rc = fail(VT_ERR_HIP, std::string("vt_host_scene_download: ") + hipGetErrorString(err));
        for (size_t k = 0; rc == VT_OK && k + 1 < s->level_begin.size(); ++k)
            for (uint32_t j = s->level_begin[k]; j < s->level_begin[k + 1]; ++j) hs.pair_depth[order[j]] = s->max_depth - uint32_t(k);
    }
    if (rc != VT_OK) { delete hsw; return rc; }
    bool alpha = false;
    for (const vt_tri64& t : hs.tris) alpha |= (t.flags & VT_TRI_ALPHATEST) != 0;
    hs.has_alpha = alpha;
    // from now on a device-side refit marks this copy stale (vt_host_scene_sync refreshes it), on every member of a group
    s->add_host_copy(hsw->stale);
    for (vt_scene* rep : s->replicas) rep->add_host_copy(hsw->stale);
    *out = hsw;
    return VT_OK;
}

} // extern "C"
