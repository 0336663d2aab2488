// bvh_build.cpp -- CPU BVH build for Rebuild (stays on the host, as in the reference).
//
// Replaces the tail of AccelStruct::PopulateAccel, source/objects/AccelStruct.cpp:763-770:
//     LocallyOrderedClusteringBuilder<BVH, uint32_t> builder(mAccel);
//     compute_bounding_boxes_and_centers / compute_bounding_boxes_union
//     builder.build(...);  LeafCollapser(mAccel).collapse();
// The builder library (madmann91/bvh v1) is an empty submodule in the reference tree,
// so this is an independent implementation of the same published pipeline
// (Meister & Bittner, "Parallel Locally-Ordered Clustering for BVH Construction", 2018):
// 32-bit Morton sort of triangle centres, bottom-up nearest-neighbour merging inside a
// +-14 window, then SAH-driven collapsing of sibling leaves.  The output uses the v1
// node layout (32-B nodes, root at 0, siblings adjacent, parents before children).
// Tree shape only affects speed and tie-broken indices; the CPU oracle and the GPU
// kernel always walk the SAME tree produced here.
#include "vt_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <xmmintrin.h>

namespace vt {

namespace {

struct Box {
    float lo[3], hi[3];
};

inline Box box_union(const Box& a, const Box& b)
{
    Box r;
    for (int k = 0; k < 3; ++k) {
        r.lo[k] = a.lo[k] < b.lo[k] ? a.lo[k] : b.lo[k];
        r.hi[k] = a.hi[k] > b.hi[k] ? a.hi[k] : b.hi[k];
    }
    return r;
}

// half surface area: (dx + dy) * dz + dx * dy
inline float half_area(const Box& b)
{
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return (dx + dy) * dz + dx * dy;
}

inline Box node_box(const vt_bvh_node& n)
{
    return Box{{n.bounds[0], n.bounds[2], n.bounds[4]}, {n.bounds[1], n.bounds[3], n.bounds[5]}};
}

inline void set_node_box(vt_bvh_node& n, const Box& b)
{
    for (int k = 0; k < 3; ++k) {
        n.bounds[2 * k]     = b.lo[k];
        n.bounds[2 * k + 1] = b.hi[k];
    }
}

// Triangle::bounding_box()/center(), source/objects/Primitives.h:104-118: the box is
// taken over p0, p1() = p0 - e1 and p2() = p0 + e2 (the re-derived vertices).
inline void tri_box_center(const vt_tri64& t, Box& b, float c[3])
{
    for (int k = 0; k < 3; ++k) {
        float p0 = t.p0[k], p1 = t.p0[k] - t.e1[k], p2 = t.p0[k] + t.e2[k];
        float lo = p0, hi = p0;
        lo = p1 < lo ? p1 : lo;  hi = p1 > hi ? p1 : hi;
        lo = p2 < lo ? p2 : lo;  hi = p2 > hi ? p2 : hi;
        b.lo[k] = lo;  b.hi[k] = hi;
        c[k] = (p0 + p1 + p2) * (1.0f / 3.0f);
    }
}

// spread the low 10 bits of x so that there are two zero bits between each
inline uint32_t spread10(uint32_t x)
{
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8))  & 0x0300F00Fu;
    x = (x | (x << 4))  & 0x030C30C3u;
    x = (x | (x << 2))  & 0x09249249u;
    return x;
}

// stable LSD radix sort of (key, value) pairs; keys have 30 significant bits
void radix_sort_pairs(std::vector<uint32_t>& keys, std::vector<uint32_t>& vals)
{
    const size_t n = keys.size();
    std::vector<uint32_t> k2(n), v2(n);
    std::vector<size_t> hist(1025);
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * 10;
        std::fill(hist.begin(), hist.end(), size_t(0));
        for (size_t i = 0; i < n; ++i) ++hist[((keys[i] >> shift) & 1023u) + 1];
        for (int b = 0; b < 1024; ++b) hist[b + 1] += hist[b];
        for (size_t i = 0; i < n; ++i) {
            size_t d = hist[(keys[i] >> shift) & 1023u]++;
            k2[d] = keys[i];
            v2[d] = vals[i];
        }
        keys.swap(k2);
        vals.swap(v2);
    }
}

constexpr int   kSearchRadius  = 14;   // bvh v1 LocallyOrderedClusteringBuilder::search_radius
constexpr float kTraversalCost = 1.0f; // bvh v1 LeafCollapser::traversal_cost

// SAH leaf collapse (bvh v1 LeafCollapser): bottom-up, an inner node whose two children
// are (possibly already collapsed) leaves becomes a leaf when
//     half_area(node) * (n_left + n_right - traversal_cost)
//         <= half_area(left) * n_left + half_area(right) * n_right.
// Parents precede children in `nodes`, so a descending index sweep is bottom-up.
void collapse_leaves(std::vector<vt_bvh_node>& nodes, std::vector<uint32_t>& prim_indices)
{
    const size_t nc = nodes.size();
    if (nc == 0 || nodes[0].prim_count != 0) return;

    std::vector<uint32_t> pcount(nc);      // > 0: (collapsed) leaf with that many prims
    std::vector<uint8_t>  removed(nc, 0);
    for (size_t k = nc; k-- > 0;) {
        const vt_bvh_node& nd = nodes[k];
        if (nd.prim_count != 0) { pcount[k] = nd.prim_count; continue; }
        const uint32_t l = nd.first, r = nd.first + 1;
        pcount[k] = 0;
        if (pcount[l] > 0 && pcount[r] > 0) {
            const float total = float(pcount[l] + pcount[r]);
            const float collapse_cost = half_area(node_box(nd)) * (total - kTraversalCost);
            const float base_cost = half_area(node_box(nodes[l])) * float(pcount[l]) +
                                    half_area(node_box(nodes[r])) * float(pcount[r]);
            if (collapse_cost <= base_cost) {
                pcount[k]  = pcount[l] + pcount[r];
                removed[l] = removed[r] = 1;
            }
        }
    }
    // nodes below a removed node are removed as well
    for (size_t k = 0; k < nc; ++k)
        if (removed[k] && nodes[k].prim_count == 0) removed[nodes[k].first] = removed[nodes[k].first + 1] = 1;

    // new index = old index - (#removed before it); removed nodes go in sibling pairs,
    // so siblings stay adjacent and parents still precede children
    std::vector<uint32_t> new_index(nc);
    uint32_t kept = 0;
    for (size_t k = 0; k < nc; ++k) { new_index[k] = kept; kept += removed[k] ? 0u : 1u; }

    std::vector<vt_bvh_node> out_nodes(kept);
    std::vector<uint32_t> out_prims;
    out_prims.reserve(prim_indices.size());
    std::vector<uint32_t> stack;
    for (size_t k = 0; k < nc; ++k) {
        if (removed[k]) continue;
        vt_bvh_node nd = nodes[k];
        if (pcount[k] == 0) {
            nd.first = new_index[nd.first];
        } else {
            // gather the subtree's primitives, left before right
            const uint32_t first = uint32_t(out_prims.size());
            stack.assign(1, uint32_t(k));
            while (!stack.empty()) {
                const uint32_t s = stack.back();
                stack.pop_back();
                const vt_bvh_node& sn = nodes[s];
                if (sn.prim_count != 0) {
                    for (uint32_t q = 0; q < sn.prim_count; ++q) out_prims.push_back(prim_indices[sn.first + q]);
                } else {
                    stack.push_back(sn.first + 1);
                    stack.push_back(sn.first);
                }
            }
            nd.prim_count = pcount[k];
            nd.first      = first;
        }
        out_nodes[new_index[k]] = nd;
    }
    nodes.swap(out_nodes);
    prim_indices.swap(out_prims);
}

} // namespace

namespace {

// ---- top-down binned SAH (Wald 2007), task-parallel ---------------------------------------------------------------
// NOT the reference's algorithm (that is PLOC + leaf collapse below, kept as VT_BUILDER_PLOC).  Kernel time is
// proportional to the node steps per ray, and this tree needs 11 % (incoherent rays) to 37 % (camera rays) fewer of them
// than the PLOC tree on the synthetic scenes (profiles/r2/notes.md) for a comparable Rebuild time.  Same v1 node layout,
// same traversal, same hits (t, u, v; the tie-broken index may differ, as between any two trees).
// Parallel form: the calling thread splits the top of the tree (bins filled by all threads); every node of at most
// kTaskPrims triangles becomes an OpenMP task that builds its subtree serially into a vector of its own; the subtrees
// are appended in the order the tasks were created.  Bin sums are integer counts and min / max, and every range is
// partitioned by one thread: the result does not depend on the number of threads or on scheduling.
// Working set (round 5): one 40-B record per triangle -- box, centre, original index -- PERMUTED IN PLACE as ranges are partitioned,
// so every pass over a node's range streams through contiguous memory (rounds 2-4 permuted 4-B indices and fetched boxes and
// centres through them: two cache lines per triangle and pass, 2.8 s per million triangles on one core).  A node costs two passes:
// binning (boxes into 3 x 16 bins, SSE min / max) and the partition, which also accumulates the bounds of both children, so only
// the root (and the rare node split by index) is ever scanned for its bounds.  The partition is libstdc++'s bidirectional
// std::partition written out (same swaps for the same predicate values), min / max are exact and order-free: the tree is
// bit-identical to the one rounds 2-4 built (tests/test_host_build.py pins its hash).
struct Prim { float lo[3], hi[3], c[3]; uint32_t idx; };
static_assert(sizeof(Prim) == 40, "Prim layout: lo | hi | c | idx, read with two overlapping 16-B loads");

struct SahCtx {
    Prim* prims;                     // permuted in place; prims[i].idx becomes prim_indices[i]
    int nthreads;
};
constexpr int      kSahBins    = 16;
constexpr uint32_t kSahMaxLeaf = 4;
constexpr uint32_t kTaskPrims  = 16384;

struct SahSplit { int axis; int bin; float cost; };
struct Bounds { Box nb, cb; };                      // of a range: its triangles' boxes, its centres
const Box kEmptyBox{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};

// lanes 0..2 of two SSE registers = a Box; lane 3 carries whatever lies behind the three floats and is never read
struct Box4 { __m128 lo, hi; };
inline Box4 empty4() { return Box4{_mm_set1_ps(FLT_MAX), _mm_set1_ps(-FLT_MAX)}; }
inline void grow4(Box4& b, const Prim& p)           // box_union(b, p's box): same operand order as the scalar form
{
    b.lo = _mm_min_ps(b.lo, _mm_loadu_ps(p.lo));
    b.hi = _mm_max_ps(b.hi, _mm_loadu_ps(p.hi));
}
inline void grow4c(Box4& b, const Prim& p)          // ... with p's centre (a point)
{
    const __m128 c = _mm_loadu_ps(p.c);
    b.lo = _mm_min_ps(b.lo, c);
    b.hi = _mm_max_ps(b.hi, c);
}
inline void merge4(Box4& a, const Box4& b) { a.lo = _mm_min_ps(a.lo, b.lo); a.hi = _mm_max_ps(a.hi, b.hi); }
inline Box to_box(const Box4& b)
{
    alignas(16) float l[4], h[4];
    _mm_store_ps(l, b.lo); _mm_store_ps(h, b.hi);
    return Box{{l[0], l[1], l[2]}, {h[0], h[1], h[2]}};
}

// bins of all three axes in one pass over [begin, end); `par` = fill them with all threads.
// Only bins that received a triangle are initialised, converted and swept (an occupancy mask per axis): most nodes hold a handful
// of triangles, and 3 x 16 bins of set-up and sweep per node were three quarters of the build.  Skipping an empty bin changes
// nothing: the split candidate "behind an empty bin" has exactly the cost of the candidate behind the last occupied one (same
// left set, same right set, same float operations) and `cost < best` is strict, so the first of the two wins either way.
SahSplit sah_best_split(const SahCtx& c, uint32_t begin, uint32_t end, const Box& cb, bool par)
{
    float scale[3], lo[3];
    bool live[3];
    for (int a = 0; a < 3; ++a) {
        const float extent = cb.hi[a] - cb.lo[a];
        live[a] = extent > 0.0f;
        scale[a] = live[a] ? float(kSahBins) / extent : 0.0f;
        lo[a] = cb.lo[a];
    }
    Box4 bin_box[3][kSahBins];
    uint32_t bin_n[3][kSahBins];
    uint32_t occupied[3] = {0, 0, 0};
    auto fill = [&](Box4 (*bb)[kSahBins], uint32_t (*bn)[kSahBins], uint32_t* occ, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) {
            const Prim& p = c.prims[i];
            for (int a = 0; a < 3; ++a) {
                if (!live[a]) continue;
                int b = int((p.c[a] - lo[a]) * scale[a]);
                b = b < 0 ? 0 : (b >= kSahBins ? kSahBins - 1 : b);
                if (occ[a] >> b & 1u) {
                    grow4(bb[a][b], p);
                    ++bn[a][b];
                } else {                                     // first triangle of this bin: union with the empty box = the box itself
                    occ[a] |= 1u << b;
                    bb[a][b] = Box4{_mm_loadu_ps(p.lo), _mm_loadu_ps(p.hi)};
                    bn[a][b] = 1;
                }
            }
        }
    };
    if (par && c.nthreads > 1) {
#pragma omp parallel num_threads(c.nthreads)
        {
            Box4 lb[3][kSahBins];
            uint32_t ln[3][kSahBins];
            uint32_t locc[3] = {0, 0, 0};
            const int64_t total = int64_t(end) - begin, nt = omp_get_num_threads(), t = omp_get_thread_num();
            fill(lb, ln, locc, begin + total * t / nt, begin + total * (t + 1) / nt);
#pragma omp critical(vt_sah_bins)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < kSahBins; ++b) {
                    if (!(locc[a] >> b & 1u)) continue;
                    if (occupied[a] >> b & 1u) { merge4(bin_box[a][b], lb[a][b]); bin_n[a][b] += ln[a][b]; }
                    else { occupied[a] |= 1u << b; bin_box[a][b] = lb[a][b]; bin_n[a][b] = ln[a][b]; }
                }
        }
    } else {
        fill(bin_box, bin_n, occupied, begin, end);
    }
    SahSplit best{-1, 0, FLT_MAX};
    for (int a = 0; a < 3; ++a) {
        if (!live[a] || (occupied[a] & (occupied[a] - 1u)) == 0) continue;     // fewer than two occupied bins: nothing to split
        int used[kSahBins], nused = 0;
        Box bins[kSahBins];
        for (int b = 0; b < kSahBins; ++b)
            if (occupied[a] >> b & 1u) { used[nused] = b; bins[nused] = to_box(bin_box[a][b]); ++nused; }
        // right_area[k] / right_n[k]: everything from occupied bin k on
        float right_area[kSahBins];
        uint32_t right_n[kSahBins];
        Box acc = kEmptyBox;
        uint32_t cnt = 0;
        for (int k = nused - 1; k > 0; --k) {
            acc = box_union(acc, bins[k]);
            cnt += bin_n[a][used[k]];
            right_area[k] = half_area(acc);
            right_n[k] = cnt;
        }
        acc = kEmptyBox;
        cnt = 0;
        for (int k = 0; k < nused - 1; ++k) {
            acc = box_union(acc, bins[k]);
            cnt += bin_n[a][used[k]];
            const float cost = half_area(acc) * float(cnt) + right_area[k + 1] * float(right_n[k + 1]);
            if (cost < best.cost) best = SahSplit{a, used[k], cost};      // axis order, then bin order: deterministic ties
        }
    }
    return best;
}

// node bounds and centroid bounds of [begin, end): the root, and nodes whose parent was split by index
Bounds sah_bounds(const SahCtx& c, uint32_t begin, uint32_t end, bool par)
{
    Box4 nb = empty4(), cb = empty4();
    auto scan = [&](Box4& n, Box4& ce, int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) { grow4(n, c.prims[i]); grow4c(ce, c.prims[i]); }
    };
    if (par && c.nthreads > 1) {
#pragma omp parallel num_threads(c.nthreads)
        {
            Box4 ln = empty4(), lc = empty4();
            const int64_t total = int64_t(end) - begin, nt = omp_get_num_threads(), t = omp_get_thread_num();
            scan(ln, lc, begin + total * t / nt, begin + total * (t + 1) / nt);
#pragma omp critical(vt_sah_bounds)
            { merge4(nb, ln); merge4(cb, lc); }
        }
    } else {
        scan(nb, cb, begin, end);
    }
    return Bounds{to_box(nb), to_box(cb)};
}

// Decides node `self` over [begin, end) whose bounds are B: sets its box and returns 0 for a leaf, else the split position `mid`
// (begin < mid < end) after partitioning the range in place.  kids_known: kids[0 / 1] hold the bounds of [begin, mid) / [mid, end).
uint32_t sah_split_node(const SahCtx& c, vt_bvh_node& self, uint32_t begin, uint32_t end, bool par, const Bounds& B, Bounds kids[2],
                        bool& kids_known)
{
    const uint32_t count = end - begin;
    const Box& nb = B.nb;
    const Box& cb = B.cb;
    kids_known = false;
    set_node_box(self, nb);
    if (count <= 1) { self.prim_count = count; self.first = begin; return 0; }
    const SahSplit best = sah_best_split(c, begin, end, cb, par);
    const float leaf_cost = half_area(nb) * (float(count) - kTraversalCost);
    uint32_t mid;
    if (best.axis < 0 || (count <= kSahMaxLeaf && best.cost >= leaf_cost)) {
        if (count <= kSahMaxLeaf) { self.prim_count = count; self.first = begin; return 0; }
        mid = begin + count / 2;                        // coincident centroids: split by index
    } else {
        const float scale = float(kSahBins) / (cb.hi[best.axis] - cb.lo[best.axis]);
        const float lo = cb.lo[best.axis];
        const int axis = best.axis, bin = best.bin;
        Box4 knb[2] = {empty4(), empty4()}, kcb[2] = {empty4(), empty4()};
        // evaluates a triangle ONCE: which side it goes to, and that side's bounds grow by it
        auto goes_left = [&](const Prim& p) {
            int b = int((p.c[axis] - lo) * scale);
            b = b < 0 ? 0 : (b >= kSahBins ? kSahBins - 1 : b);
            const int side = b <= bin ? 0 : 1;
            grow4(knb[side], p); grow4c(kcb[side], p);
            return side == 0;
        };
        // libstdc++'s std::partition for bidirectional iterators, written out (it applies the predicate exactly once per
        // element): in place, and every range is partitioned by exactly one thread with this one algorithm, so the order it
        // leaves is as deterministic as a stable one -- and the same as when 4-B indices were partitioned
        Prim* first = c.prims + begin;
        Prim* last = c.prims + end;
        for (;;) {
            for (;;) {
                if (first == last) goto partitioned;
                if (goes_left(*first)) ++first; else break;
            }
            --last;
            for (;;) {
                if (first == last) goto partitioned;
                if (!goes_left(*last)) --last; else break;
            }
            std::swap(*first, *last);
            ++first;
        }
    partitioned:
        mid = uint32_t(first - c.prims);
        if (mid == begin || mid == end) mid = begin + count / 2;
        else {
            kids[0] = Bounds{to_box(knb[0]), to_box(kcb[0])};
            kids[1] = Bounds{to_box(knb[1]), to_box(kcb[1])};
            kids_known = true;
        }
    }
    self.prim_count = 0;
    return mid;
}

void refine_nodes(std::vector<vt_bvh_node>& N, int nthreads, int passes, float fraction);

// What a subtree task spends on its own subtree after building it (see refine_nodes): three passes over the worst 2 % of
// its inner nodes.  The subtree is cache resident and the tasks run side by side, so this does not show in the build
// time (1 M triangles, 8 threads: 0.41 s with and without), and it removes 3.3 % of the node steps per incoherent ray.
constexpr int   kLocalRefinePasses = 3;
constexpr float kLocalRefineFraction = 0.02f;

// serial build of the subtree over [begin, end) (bounds B) into `out` (out[0] = its root, children behind their parents)
void sah_build_subtree(const SahCtx& c, uint32_t begin, uint32_t end, const Bounds& B, std::vector<vt_bvh_node>& out)
{
    out.clear();
    out.reserve(size_t(end - begin));
    out.emplace_back();
    struct Task { uint32_t node, begin, end; bool known; Bounds b; };
    std::vector<Task> stack{{0u, begin, end, true, B}};
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        vt_bvh_node self{};
        Bounds kids[2];
        bool kids_known = false;
        const uint32_t mid = sah_split_node(c, self, t.begin, t.end, false, t.known ? t.b : sah_bounds(c, t.begin, t.end, false), kids, kids_known);
        if (mid != 0) {
            self.first = uint32_t(out.size());
            out.emplace_back();
            out.emplace_back();
            stack.push_back({self.first + 1, mid, t.end, kids_known, kids[1]});
            stack.push_back({self.first, t.begin, mid, kids_known, kids[0]});
        }
        out[t.node] = self;
    }
    refine_nodes(out, 1, kLocalRefinePasses, kLocalRefineFraction);
}

// Second level of the task tree: a REGION of at most kRegionPrims triangles is split on the thread that runs its task
// down to subtree tasks (above), and once those have finished, the stitched region is refined again as a whole --
// re-insertion across the borders of its subtrees, where the subtree tasks could not look.  Measured on S1M (262 144 bounce
// rays, oracle counters): 56.95 -> 55.40 steps per ray, 16 Mi bounce rays 4.30 -> 4.20 ms, camera rays 44.1 -> 41.8 steps, at an unchanged build time on 8 threads (profiles/r3/notes.md); the global pass of
// VT_BUILDER_BINNED_SAH_REFINED still adds a little on top (55.1).
#ifndef VT_REGION_PRIMS            // measurement overrides (make variant DEFS=-DVT_REGION_...): profiles/r3/notes.md
#define VT_REGION_PRIMS 65536
#endif
#ifndef VT_REGION_REFINE_PASSES
#define VT_REGION_REFINE_PASSES 3
#endif
#ifndef VT_REGION_REFINE_FRACTION
#define VT_REGION_REFINE_FRACTION 0.02f
#endif
constexpr uint32_t kRegionPrims = VT_REGION_PRIMS;
constexpr int      kRegionRefinePasses = VT_REGION_REFINE_PASSES;
constexpr float    kRegionRefineFraction = VT_REGION_REFINE_FRACTION;

// Splits [begin, end) on the calling thread until every piece holds at most piece_prims triangles, has every piece built by
// build_piece(begin, end, nodes) -- as OpenMP tasks when `tasks` (the caller is inside a parallel region), else by a
// parallel loop --, and stitches the result into `out` (v1 layout: out[0] = root, siblings adjacent, children behind their
// parents; pieces appended in creation order, so the tree does not depend on scheduling or on the number of threads).
template <class BuildPiece>
void sah_build_split(const SahCtx& c, uint32_t begin, uint32_t end, const Bounds& B, uint32_t piece_prims, bool par_bins, bool tasks,
                     const BuildPiece& build_piece, std::vector<vt_bvh_node>& out)
{
    struct Piece { uint32_t slot, begin, end; Bounds b; std::vector<vt_bvh_node> nodes; };
    std::vector<vt_bvh_node> top;
    top.emplace_back();
    std::vector<Piece> pieces;
    {
        struct Task { uint32_t node, begin, end; bool known; Bounds b; };
        std::vector<Task> stack{{0u, begin, end, true, B}};
        while (!stack.empty()) {
            const Task t = stack.back();
            stack.pop_back();
            const Bounds tb = t.known ? t.b : sah_bounds(c, t.begin, t.end, par_bins);
            if (t.end - t.begin <= piece_prims) { pieces.push_back(Piece{t.node, t.begin, t.end, tb, {}}); continue; }
            vt_bvh_node self{};
            Bounds kids[2];
            bool kids_known = false;
            const uint32_t mid = sah_split_node(c, self, t.begin, t.end, par_bins, tb, kids, kids_known);
            if (mid != 0) {
                self.first = uint32_t(top.size());
                top.emplace_back();
                top.emplace_back();
                stack.push_back({self.first + 1, mid, t.end, kids_known, kids[1]});
                stack.push_back({self.first, t.begin, mid, kids_known, kids[0]});
            }
            top[t.node] = self;
        }
    }
    if (tasks) {
        for (size_t k = 0; k < pieces.size(); ++k) {
            Piece* p = &pieces[k];
#pragma omp task firstprivate(p) shared(c, build_piece)
            build_piece(p->begin, p->end, p->b, p->nodes);
        }
#pragma omp taskwait
    } else {
#pragma omp parallel num_threads(c.nthreads)
#pragma omp single
        {
            for (size_t k = 0; k < pieces.size(); ++k) {
                Piece* p = &pieces[k];
#pragma omp task firstprivate(p) shared(c, build_piece)
                build_piece(p->begin, p->end, p->b, p->nodes);
            }
#pragma omp taskwait
        }
    }
    // stitch: a piece's root goes into its slot of the top, the rest behind everything emitted so far
    size_t total = top.size();
    for (const Piece& pc : pieces) total += pc.nodes.size() - 1;
    top.reserve(total);
    for (Piece& pc : pieces) {
        const uint32_t base = uint32_t(top.size()) - 1u;         // local index i >= 1 -> base + i
        for (vt_bvh_node& nd : pc.nodes)
            if (nd.prim_count == 0) nd.first += base;
        top[pc.slot] = pc.nodes[0];
        top.insert(top.end(), pc.nodes.begin() + 1, pc.nodes.end());
        std::vector<vt_bvh_node>().swap(pc.nodes);
    }
    out.swap(top);
}

int build_binned_sah(const vt_tri64* tris, uint32_t n, int nthreads, Bvh& out)
{
    std::vector<Prim> prims(n);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < int64_t(n); ++i) {
        Prim& p = prims[size_t(i)];
        Box b;
        tri_box_center(tris[i], b, p.c);
        // A triangle with a NaN / infinite vertex can never be hit (Primitives.h:173-189 yields NaN or -inf), but its box
        // would poison every box above it (NaN slab terms drop out of the reference's test, so rays would walk into it):
        // it gets the empty box (neutral in every union, fails every slab test) and a harmless centre.
        bool finite = true;
        for (int k = 0; k < 3; ++k)
            finite = finite && std::fabs(b.lo[k]) <= FLT_MAX && std::fabs(b.hi[k]) <= FLT_MAX;
        if (!finite) {
            b = kEmptyBox;
            p.c[0] = p.c[1] = p.c[2] = 0.0f;
        }
        for (int k = 0; k < 3; ++k) { p.lo[k] = b.lo[k]; p.hi[k] = b.hi[k]; }
        p.idx = uint32_t(i);
    }
    const SahCtx ctx{prims.data(), nthreads};

    // top of the tree on this thread (bins filled by all threads); regions of <= kRegionPrims triangles as tasks, each of
    // which splits itself into subtree tasks of <= kTaskPrims triangles and refines the stitched region afterwards
    const auto build_subtree = [&ctx](uint32_t b, uint32_t e, const Bounds& B, std::vector<vt_bvh_node>& nodes) { sah_build_subtree(ctx, b, e, B, nodes); };
    const auto build_region = [&ctx, &build_subtree](uint32_t b, uint32_t e, const Bounds& B, std::vector<vt_bvh_node>& nodes) {
        if (e - b <= kTaskPrims) { sah_build_subtree(ctx, b, e, B, nodes); return; }
        sah_build_split(ctx, b, e, B, kTaskPrims, false, true, build_subtree, nodes);
        refine_nodes(nodes, 1, kRegionRefinePasses, kRegionRefineFraction);
    };
    sah_build_split(ctx, 0u, n, sah_bounds(ctx, 0u, n, true), kRegionPrims, true, false, build_region, out.nodes);
    out.prim_indices.resize(n);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < int64_t(n); ++i) out.prim_indices[size_t(i)] = prims[size_t(i)].idx;
    return VT_OK;
}

int build_ploc(const vt_tri64* tris, uint32_t n, int nthreads, Bvh& out);

// ---- refinement by re-insertion: inside every subtree task of the SAH builder (above), and once more over the whole
// ---- finished tree as the opt-in VT_BUILDER_BINNED_SAH_REFINED ------------------------------------------------------
// Insertion-based optimisation after Bittner, Hapala, Havran, "Fast insertion-based optimization of bounding volume
// hierarchies" (CGF 2013): the inner nodes whose boxes are largest for what they hold are taken out, and their two
// subtrees are put back where they enlarge the tree least (branch-and-bound search over the tree; the cost of a
// position = the area of the new parent + the growth of every box above it).  Leaves and the triangle order stay as
// they are, only inner nodes move.  Over the whole tree (two passes over the worst 1 % each; 1 M triangles +0.09 s, most of
// it array set-up and the re-layout of all nodes) it reaches 5.4 % fewer node steps per incoherent ray; inside the builder's
// subtree tasks it is free and reaches 3.3 % (profiles/r2/notes.md).  Serial per call and deterministic.
constexpr int   kRefinePasses = 2;
constexpr float kRefineFraction = 0.01f;

void refine_nodes(std::vector<vt_bvh_node>& N, int nthreads, int passes, float fraction)
{
    if (N.size() < 7 || N.size() > size_t(0x3FFFFFFF) || N[0].prim_count != 0) return;   // node ids are ints here
    const int nc = int(N.size());
    const int root = 0;
    std::vector<int> parent(size_t(nc), -1), left(size_t(nc), -1), right(size_t(nc), -1);
    std::vector<Box> box(static_cast<size_t>(nc));
    std::vector<float> area(size_t(nc), 0.0f);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int k = 0; k < nc; ++k) {                        // every child has one parent: the writes are disjoint
        box[size_t(k)] = node_box(N[size_t(k)]);
        area[size_t(k)] = half_area(box[size_t(k)]);
        if (N[size_t(k)].prim_count == 0) {
            left[size_t(k)] = int(N[size_t(k)].first); right[size_t(k)] = left[size_t(k)] + 1;
            parent[size_t(left[size_t(k)])] = k; parent[size_t(right[size_t(k)])] = k;
        }
    }
    auto refit_up = [&](int k) {
        while (k >= 0) {
            const Box nb = box_union(box[size_t(left[size_t(k)])], box[size_t(right[size_t(k)])]);
            if (std::memcmp(&nb, &box[size_t(k)], sizeof(Box)) == 0) break;
            box[size_t(k)] = nb;
            area[size_t(k)] = half_area(nb);
            k = parent[size_t(k)];
        }
    };
    auto replace_child = [&](int p, int oldc, int newc) {
        if (left[size_t(p)] == oldc) left[size_t(p)] = newc; else right[size_t(p)] = newc;
        parent[size_t(newc)] = p;
    };
    struct Entry { float cost; int node; bool operator<(const Entry& o) const { return cost > o.cost || (cost == o.cost && node > o.node); } };
    std::vector<Entry> heap;
    // the node next to which a subtree with box xb costs least (never the root: its place is fixed)
    auto find_best = [&](const Box& xb, float xa) {
        float best_cost = FLT_MAX; int best = left[size_t(root)];
        heap.clear();
        heap.push_back({0.0f, root});
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end());
            const Entry e = heap.back(); heap.pop_back();
            if (!(e.cost + xa < best_cost)) break;                      // nothing below can beat the best position
            // subtrees of non-finite triangles carry the empty box (area +inf): never a position, never searched -- an
            // induced cost of -inf would otherwise draw every re-insertion into them and give them a real box
            if (!(area[size_t(e.node)] <= FLT_MAX)) continue;
            const float direct = half_area(box_union(box[size_t(e.node)], xb));
            const float total = e.cost + direct;
            if (e.node != root && total < best_cost) { best_cost = total; best = e.node; }
            const float induced = total - area[size_t(e.node)];         // what every position below pays at least
            if (left[size_t(e.node)] >= 0 && induced + xa < best_cost) {
                heap.push_back({induced, left[size_t(e.node)]}); std::push_heap(heap.begin(), heap.end());
                heap.push_back({induced, right[size_t(e.node)]}); std::push_heap(heap.begin(), heap.end());
            }
        }
        return best;
    };
    std::vector<std::pair<float, int>> cand;
    std::vector<char> touched(static_cast<size_t>(nc));
    std::vector<float> score(static_cast<size_t>(nc));
    const auto worse = [](const std::pair<float, int>& x, const std::pair<float, int>& y) {
        return x.first > y.first || (x.first == y.first && x.second < y.second);
    };
    for (int pass = 0; pass < passes; ++pass) {
        cand.clear();
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int k = 0; k < nc; ++k) {
            score[size_t(k)] = -1.0f;
            if (left[size_t(k)] < 0 || k == root || parent[size_t(k)] < 0 || parent[size_t(k)] == root) continue;
            const float a = area[size_t(k)], al = area[size_t(left[size_t(k)])], ar = area[size_t(right[size_t(k)])];
            const float mn = al < ar ? al : ar, mx = al < ar ? ar : al;
            // area x (area / smaller child) x (area / mean child): large boxes over small or lopsided content first
            const float pr = a * (a / mn) * (a / (0.5f * (al + ar)));
            if (!(mn > 0.0f) || !(mx <= FLT_MAX) || !(pr <= FLT_MAX)) continue;     // degenerate / empty boxes stay
            score[size_t(k)] = pr;
        }
        for (int k = 0; k < nc; ++k)
            if (score[size_t(k)] >= 0.0f) cand.push_back({score[size_t(k)], k});
        const size_t take = std::min(cand.size(), std::max<size_t>(1, size_t(float(cand.size()) * fraction)));
        if (take < cand.size()) std::nth_element(cand.begin(), cand.begin() + long(take), cand.end(), worse);
        cand.resize(take);
        std::sort(cand.begin(), cand.end(), worse);
        std::fill(touched.begin(), touched.end(), 0);
        for (const auto& c : cand) {
            const int n = c.second, P = parent[size_t(n)];
            // a node that an earlier re-insertion of this pass moved, or whose parent it moved, waits for the next pass
            if (touched[size_t(n)] || touched[size_t(P)] || P == root || parent[size_t(P)] < 0) continue;
            const int G = parent[size_t(P)], S = left[size_t(P)] == n ? right[size_t(P)] : left[size_t(P)];
            int X[2] = {left[size_t(n)], right[size_t(n)]};
            if (area[size_t(X[0])] < area[size_t(X[1])]) std::swap(X[0], X[1]);      // the larger subtree first
            replace_child(G, P, S);                                                   // n and P leave the tree
            refit_up(G);
            const int spare[2] = {n, P};
            touched[size_t(n)] = touched[size_t(P)] = 1;
            for (int q = 0; q < 2; ++q) {
                const int x = X[q], Q = spare[q];
                const int y = find_best(box[size_t(x)], area[size_t(x)]);
                const int py = parent[size_t(y)];
                replace_child(py, y, Q);                                              // Q takes y's place over y and x
                left[size_t(Q)] = y; right[size_t(Q)] = x; parent[size_t(y)] = Q; parent[size_t(x)] = Q;
                box[size_t(Q)] = box_union(box[size_t(y)], box[size_t(x)]);
                area[size_t(Q)] = half_area(box[size_t(Q)]);
                touched[size_t(Q)] = 1;
                refit_up(py);
            }
        }
    }
    // back to the v1 layout: root at 0, siblings adjacent, parents before their children
    std::vector<vt_bvh_node> out;
    out.reserve(size_t(nc));
    out.emplace_back();
    std::vector<std::pair<int, uint32_t>> todo{{root, 0u}};
    while (!todo.empty()) {
        const int o = todo.back().first; const uint32_t d = todo.back().second;
        todo.pop_back();
        vt_bvh_node nd = N[size_t(o)];
        set_node_box(nd, box[size_t(o)]);
        if (left[size_t(o)] >= 0) {
            nd.prim_count = 0;
            nd.first = uint32_t(out.size());
            out.emplace_back(); out.emplace_back();
            todo.push_back({right[size_t(o)], nd.first + 1}); todo.push_back({left[size_t(o)], nd.first});
        }
        out[d] = nd;
    }
    N.swap(out);
}

} // namespace

int bvh_build(const vt_tri64* tris, uint32_t n, int nthreads, int builder, Bvh& out)
{
    out.nodes.clear();
    out.prim_indices.clear();
    if (n == 0) return VT_OK;
    if (n > 0x7FFFFFFFu) return fail(VT_ERR_INVALID_ARG, "vt_bvh_build: more than 2^31-1 triangles");
    if (!tris) return fail(VT_ERR_INVALID_ARG, "vt_bvh_build: tris is NULL");
#ifdef _OPENMP
    // default: up to 16 threads.  The PLOC rounds are many short parallel regions; measured on a 2 x 64-core
    // host (scripts/build_rate.py, 1 M triangles): 1 thread 0.60 s, 16 threads 0.20 s, 128 threads 0.52 s,
    // 256 threads 3.4 s -- more threads only add fork/join and cross-socket traffic.
    if (nthreads <= 0) nthreads = std::min(omp_get_max_threads(), 16);
    // a small scene is built faster by few threads than it takes to wake many (10 k triangles: 30 ms on one thread, 50 on two,
    // more on eight); the tree does not depend on the thread count
    nthreads = std::max(1, std::min(nthreads, int(n / 8192u)));
#else
    nthreads = 1;
#endif
    if (builder == VT_BUILDER_PLOC) return build_ploc(tris, n, nthreads, out);
    if (builder == VT_BUILDER_BINNED_SAH) return build_binned_sah(tris, n, nthreads, out);
    if (builder == VT_BUILDER_BINNED_SAH_REFINED) {
        const int rc = build_binned_sah(tris, n, nthreads, out);
        if (rc == VT_OK) refine_nodes(out.nodes, nthreads, kRefinePasses, kRefineFraction);
        return rc;
    }
    return fail(VT_ERR_INVALID_ARG, "vt_bvh_build_ex: unknown builder");
}

// Refit: same topology, bounds recomputed bottom-up.  Parents precede children in `nodes`, so a
// descending index sweep visits children first.
int bvh_refit(Bvh& bvh, const vt_tri64* tris)
{
    if (bvh.nodes.empty()) return VT_OK;
    if (!tris) return fail(VT_ERR_INVALID_ARG, "vt_bvh_refit: tris is NULL");
    for (size_t k = bvh.nodes.size(); k-- > 0;) {
        vt_bvh_node& nd = bvh.nodes[k];
        Box b;
        if (nd.prim_count != 0) {
            float c[3];
            tri_box_center(tris[bvh.prim_indices[nd.first]], b, c);
            for (uint32_t q = 1; q < nd.prim_count; ++q) {
                Box t;
                tri_box_center(tris[bvh.prim_indices[nd.first + q]], t, c);
                b = box_union(b, t);
            }
        } else {
            b = box_union(node_box(bvh.nodes[nd.first]), node_box(bvh.nodes[nd.first + 1]));
        }
        set_node_box(nd, b);
    }
    return VT_OK;
}

namespace {

int build_ploc(const vt_tri64* tris, uint32_t n, int nthreads, Bvh& out)
{

    // 1. boxes, centres, scene box
    std::vector<Box> boxes(n);
    std::vector<float> centers(size_t(n) * 3);
#pragma omp parallel for schedule(static) num_threads(nthreads)
    for (int64_t i = 0; i < int64_t(n); ++i) tri_box_center(tris[i], boxes[i], &centers[size_t(i) * 3]);

    Box global = boxes[0];
    for (uint32_t i = 1; i < n; ++i) global = box_union(global, boxes[i]);

    // 2. Morton codes on a 1024^3 grid over the scene box, stable sort
    std::vector<uint32_t> codes(n), order(n);
    {
        const float dim = 1024.0f;
        float scale[3], offset[3];
        for (int k = 0; k < 3; ++k) {
            scale[k]  = dim * (1.0f / (global.hi[k] - global.lo[k]));
            offset[k] = -global.lo[k] * scale[k];
        }
#pragma omp parallel for schedule(static) num_threads(nthreads)
        for (int64_t i = 0; i < int64_t(n); ++i) {
            uint32_t g[3];
            for (int k = 0; k < 3; ++k) {
                float p = centers[size_t(i) * 3 + k] * scale[k] + offset[k];
                float q = p > 0.0f ? p : 0.0f; // also maps NaN (flat scene axis) to cell 0
                q = q < dim - 1.0f ? q : dim - 1.0f;
                g[k] = uint32_t(q);
            }
            codes[i] = spread10(g[0]) | (spread10(g[1]) << 1) | (spread10(g[2]) << 2);
            order[i] = uint32_t(i);
        }
        radix_sort_pairs(codes, order);
    }

    // 3. PLOC.  `nodes` is filled from the back: each round puts the children it merges
    // in front of those emitted by earlier rounds, so the root lands at index 0 and
    // every parent precedes its two adjacent children.
    const size_t node_count = size_t(2) * n - 1;
    std::vector<vt_bvh_node> nodes(node_count);
    std::vector<vt_bvh_node> cur(n), next;
    next.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        set_node_box(cur[i], boxes[order[i]]);
        cur[i].prim_count = 1;
        cur[i].first      = i; // slot in `order`, which becomes prim_indices
    }
    std::vector<Box>().swap(boxes);
    std::vector<float>().swap(centers);
    std::vector<uint32_t>().swap(codes);

    size_t tail = node_count;
    std::vector<uint32_t> nbr(n), slot(n);
    while (cur.size() > 1) {
        const int64_t m = int64_t(cur.size());
        // nearest neighbour = smallest union half-area within [i-14, i+14]; lowest index wins ties
#pragma omp parallel for schedule(static) num_threads(nthreads) if (m > 4096)
        for (int64_t i = 0; i < m; ++i) {
            const int64_t b = i > kSearchRadius ? i - kSearchRadius : 0;
            const int64_t e = i + kSearchRadius + 1 < m ? i + kSearchRadius + 1 : m;
            const Box bi = node_box(cur[i]);
            float best = std::numeric_limits<float>::max();
            int64_t best_j = -1;
            for (int64_t j = b; j < e; ++j) {
                if (j == i) continue;
                const float d = half_area(box_union(bi, node_box(cur[j])));
                if (d < best) { best = d; best_j = j; }
            }
            if (best_j < 0) best_j = (i + 1 < m) ? i + 1 : i - 1; // every distance inf/NaN
            nbr[i] = uint32_t(best_j);
        }
        // leaders (lower index of a mutual pair) get consecutive child slots
        size_t merged = 0;
        for (int64_t i = 0; i < m; ++i) {
            const uint32_t j = nbr[i];
            if (nbr[j] == uint32_t(i) && uint32_t(i) < j) slot[i] = uint32_t(merged++);
        }
        if (merged == 0) return fail(VT_ERR_INVALID_ARG, "vt_bvh_build: clustering made no progress");
        const size_t children_begin = tail - 2 * merged;
        next.clear();
        for (int64_t i = 0; i < m; ++i) {
            const uint32_t j = nbr[i];
            if (nbr[j] != uint32_t(i)) { next.push_back(cur[i]); continue; } // unmerged: carried over
            if (uint32_t(i) < j) continue;                                    // leader: emitted at j
            // follower: the parent takes this position
            const uint32_t lead = j;
            const size_t fc = children_begin + 2 * size_t(slot[lead]);
            nodes[fc]     = cur[lead];
            nodes[fc + 1] = cur[i];
            vt_bvh_node parent;
            set_node_box(parent, box_union(node_box(cur[i]), node_box(cur[lead])));
            parent.prim_count = 0;
            parent.first      = uint32_t(fc);
            next.push_back(parent);
        }
        tail = children_begin;
        cur.swap(next);
    }
    nodes[0] = cur[0]; // tail == 1 here

    // 4. SAH leaf collapse
    collapse_leaves(nodes, order);

    out.nodes.swap(nodes);
    out.prim_indices.swap(order);
    return VT_OK;
}

} // namespace

} // namespace vt
