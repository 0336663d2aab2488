// linearise.cpp -- triangle set-up and re-packing of the v1 tree for the device.
//
// tri_setup      : TriangleBackfaceCull ctor + ComputeNormalAndLoD,
//                  source/objects/Primitives.h:75-102 (only the fields intersect() reads).
// scene_linearise: stands where the reference constructs its intersector/traverser over
//                  the finished tree (source/objects/AccelStruct.cpp:772-773).  Produces
//                    pairs[] : one 64-B record per inner node = its two children, in
//                              depth-first (left-first) order, so a ray that descends into
//                              the left child reads the very next record;
//                    tris[]  : 64-B triangle records pre-shuffled into leaf order (the
//                              reference indirects through primitive_indices per test),
//                              each carrying the original index it must be reported as.
#include "vt_internal.h"

#include <utility>
#include <vector>

namespace vt {

void tri_setup(const float p0[3], const float p1[3], const float p2[3], uint32_t prim,
               uint32_t flags, vt_tri64& out)
{
    for (int k = 0; k < 3; ++k) {
        out.p0[k] = p0[k];
        out.e1[k] = p0[k] - p1[k]; // Primitives.h:82
        out.e2[k] = p2[k] - p0[k]; // Primitives.h:82
    }
    // n = cross(e1, e2)  (LeftHandedNormal = true, Primitives.h:93)
    out.n[0] = out.e1[1] * out.e2[2] - out.e1[2] * out.e2[1];
    out.n[1] = out.e1[2] * out.e2[0] - out.e1[0] * out.e2[2];
    out.n[2] = out.e1[0] * out.e2[1] - out.e1[1] * out.e2[0];
    out.prim   = prim;
    out.flags  = flags;
    out.pad[0] = out.pad[1] = 0;
}

int scene_linearise(const Bvh& bvh, const vt_tri64* tris, HostScene& out)
{
    out = HostScene();
    const auto& nodes = bvh.nodes;
    if (nodes.empty()) return VT_OK; // empty scene: every ray misses
    if (!tris) return fail(VT_ERR_INVALID_ARG, "vt_scene_linearise: tris is NULL");

    out.tris.reserve(bvh.prim_indices.size());
    auto emit_leaf = [&](const vt_bvh_node& leaf) -> uint32_t {
        const uint32_t first = uint32_t(out.tris.size());
        for (uint32_t q = 0; q < leaf.prim_count; ++q) {
            const uint32_t idx = bvh.prim_indices[leaf.first + q];
            vt_tri64 t = tris[idx];
            t.prim = idx;
            out.has_alpha = out.has_alpha || (t.flags & VT_TRI_ALPHATEST) != 0;
            out.tris.push_back(t);
        }
        return first;
    };

    if (nodes[0].prim_count != 0) {
        out.root_leaf_count = nodes[0].prim_count;
        emit_leaf(nodes[0]);
        return VT_OK;
    }

    const size_t npairs = (nodes.size() - 1) / 2;
    out.pairs.resize(npairs);
    out.pair_depth.resize(npairs);
    // Numbering = the order in which pairs are visited below: depth-first, left subtree first, so a ray that descends to
    // the left reads the next record.  A pair's children keep their sides (left = child[0]: the walk's tie rule depends on
    // it).  The two leaves of a pair are emitted when the pair is numbered (left before right), which keeps sibling leaves
    // contiguous in the triangle array.  (Round 3 measured larger-child-first, a breadth-first top and triangles interleaved
    // with their pairs: fabric traffic -5 %, kernel time unchanged on S1M and -2.4 % on S10M -- profiles/r3/notes.md; not kept.)
    struct Item { uint32_t fc, depth, parent, side; }; // fc = v1 index of the pair's left node
    uint32_t next_pair = 0;
    bool malformed = false;
    // numbers one pair, copies its two children, emits their leaves; returns the inner children in visiting order
    auto visit = [&](const Item& it, Item kids[2]) -> int {
        const uint32_t me = next_pair++;
        if (me >= npairs) { malformed = true; return 0; }
        if (it.parent != 0xFFFFFFFFu) out.pairs[it.parent].child[it.side].first = me;
        if (it.depth > out.max_depth) out.max_depth = it.depth;
        out.pair_depth[me] = it.depth;
        vt_node_pair& P = out.pairs[me];
        for (int side = 0; side < 2; ++side) {
            const vt_bvh_node& c = nodes[it.fc + side];
            P.child[side] = c;
            if (c.prim_count != 0) P.child[side].first = emit_leaf(c);
        }
        int nk = 0;
        for (int side = 0; side < 2; ++side) {
            const vt_bvh_node& c = nodes[it.fc + side];
            if (c.prim_count == 0) kids[nk++] = {c.first, it.depth + 1, me, uint32_t(side)};
        }
        return nk;
    };
    std::vector<Item> stack{{nodes[0].first, 1, 0xFFFFFFFFu, 0}};
    while (!stack.empty() && !malformed) {
        const Item it = stack.back();
        stack.pop_back();
        Item kids[2];
        const int nk = visit(it, kids);
        for (int q = nk - 1; q >= 0; --q) stack.push_back(kids[q]);   // left child popped first
    }
    if (malformed) return fail(VT_ERR_INVALID_ARG, "vt_scene_linearise: malformed tree");
    if (next_pair != npairs) return fail(VT_ERR_INVALID_ARG, "vt_scene_linearise: malformed tree");
    return VT_OK;
}

} // namespace vt
