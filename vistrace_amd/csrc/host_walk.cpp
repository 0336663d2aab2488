// host_walk.cpp -- the single-ray latency path of the C ABI (vt_host_scene_trace_*).
//
// BASELINE config 1 / SURVEY.md 8(b): one `accel:Traverse(origin, dir)` call from GLua is one ray.  A lone ray on the
// GPU is launch bound (~20 us: one launch + ~15 dependent record fetches over PCIe-visible memory) while the same walk
// on a host core takes 1-2 us, so the host class keeps the linearised scene it uploaded and answers single rays (and
// batches below the measured crossover) here.  This is NOT a fallback: vt_trace_closest / vt_trace_any and every *_dev
// entry never route to this file, and an AccelStruct cannot be built without a HIP device.
//
// The walk is the reference's, on the device layout: one iteration of bvh v1 SingleRayTraverser::traverse per sibling
// pair (source/objects/AccelStruct.cpp:818) with FastNodeIntersector's unfused slab arithmetic, and
// TriangleBackfaceCull::intersect (source/objects/Primitives.h:168-215) per leaf triangle, in the same order as the
// device kernel (vistrace_amd/csrc/trace_kernels.hip) -- results are bit-identical to it (tests/test_gpu_parity.py).
// Built with -ffp-contract=off: no FMA contraction, IEEE divide.
// VT_MUT(k, wrong, right): mutation sites with the numbers of trace_kernels.hip's list (1 swap on >=, 2 t < tmax, 4 leaf slots from
// the back, 5 plain 1/x, 6 no tmin term, 7 first < second, 8 / 14 / 15 u, v, w > 0, 11 t > tmin, 12 w = 1 - (u + v), 13 front faces
// culled); compiled out of the product.  scripts/mutants_host.sh runs the CPU tests against each -- no GPU needed.
#include "vt_internal.h"

#include <string>

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace vt {

namespace {

inline float safe_inverse(float x)            // bvh v1 utilities.hpp
{
    return VT_MUT(5, 1.0f / x, std::fabs(x) <= FLT_EPSILON ? std::copysign(1.0f / FLT_EPSILON, x) : 1.0f / x);
}
inline float robust_max(float a, float b) { return a > b ? a : b; }
inline float robust_min(float a, float b) { return a < b ? a : b; }

inline uint32_t wrap_index(float f, uint32_t n)
{
    const long long i = (long long)f;            // f is integral and |f| < 1e9
    const long long m = i % (long long)n;
    return uint32_t(m < 0 ? m + (long long)n : m);
}

// Primitives.h:196-208 with the texel lookup defined at vt_alpha_material (include/vistrace_hip.h)
bool alpha_pass(const HostScene& hs, uint32_t prim, float u, float v)
{
    const vt_tri_attribs& A = hs.attribs[prim];
    if (A.material >= hs.alpha_mats.size()) return true;
    const vt_alpha_material& M = hs.alpha_mats[A.material];
    const float w = 1.0f - u - v;
    const float tx = (w * A.uv[0][0] + u * A.uv[1][0]) + v * A.uv[2][0];
    const float ty = (w * A.uv[0][1] + u * A.uv[1][1]) + v * A.uv[2][1];
    const float s = ((tx * M.tex_mat[0][0] + ty * M.tex_mat[0][1]) + (M.tex_mat[0][2] + M.tex_mat[0][3])) * M.tex_scale;
    const float t = ((tx * M.tex_mat[1][0] + ty * M.tex_mat[1][1]) + (M.tex_mat[1][2] + M.tex_mat[1][3])) * M.tex_scale;
    float alpha = 1.0f;
    if (M.width != 0 && M.height != 0) {
        const uint8_t* img = hs.alpha_texels.data() + M.offset;
        float x = s * float(M.width), y = t * float(M.height);
        if (!(std::fabs(x) < 1.0e9f)) x = 0.0f;
        if (!(std::fabs(y) < 1.0e9f)) y = 0.0f;
        if (M.filter == 0) {
            const uint32_t xi = wrap_index(std::floor(x), M.width), yi = wrap_index(std::floor(y), M.height);
            alpha = float(img[size_t(yi) * M.width + xi]) / 255.0f;
        } else {
            const float fx = x - 0.5f, fy = y - 0.5f;
            const float x0 = std::floor(fx), y0 = std::floor(fy);
            const float ax = fx - x0, ay = fy - y0;
            const uint32_t i0 = wrap_index(x0, M.width), i1 = wrap_index(x0 + 1.0f, M.width);
            const uint32_t j0 = wrap_index(y0, M.height), j1 = wrap_index(y0 + 1.0f, M.height);
            const float a00 = float(img[size_t(j0) * M.width + i0]), a10 = float(img[size_t(j0) * M.width + i1]);
            const float a01 = float(img[size_t(j1) * M.width + i0]), a11 = float(img[size_t(j1) * M.width + i1]);
            const float top = a00 * (1.0f - ax) + a10 * ax;
            const float bot = a01 * (1.0f - ax) + a11 * ax;
            alpha = (top * (1.0f - ay) + bot * ay) / 255.0f;
        }
    }
    return !(alpha < M.alpha_ref);
}

struct Walk {
    const HostScene& hs;
    bool alpha;
    float ox, oy, oz, dx, dy, dz, tmin, tmax;
    uint32_t prim = VT_MISS;
    float u = 0.f, v = 0.f;

    // intersect_leaf: ascending slot order, best = hit, tmax = t (any_hit: stop at the first)
    template <bool ANY_HIT> bool leaf(uint32_t first, uint32_t count)
    {
        for (uint32_t k = first; k < first + count; ++k) {
            const uint32_t q = VT_MUT(4, first + count - 1 - (k - first), k);
            const vt_tri64& T = hs.tris[q];
            const float nDotDir = (T.n[0] * dx + T.n[1] * dy) + T.n[2] * dz;                      // :173
            const bool culled = (T.flags & VT_TRI_CULL_BACKFACE) && VT_MUT(13, nDotDir < 0.0f, nDotDir > 0.0f);   // :174
            const float cx = T.p0[0] - ox, cy = T.p0[1] - oy, cz = T.p0[2] - oz;                  // :176
            const float rx = dy * cz - dz * cy;                                                   // :177
            const float ry = dz * cx - dx * cz;
            const float rz = dx * cy - dy * cx;
            const float inv_det = 1.0f / nDotDir;                                                 // :178
            const float uu = ((rx * T.e2[0] + ry * T.e2[1]) + rz * T.e2[2]) * inv_det;            // :180
            const float vv = ((rx * T.e1[0] + ry * T.e1[1]) + rz * T.e1[2]) * inv_det;            // :181
            const float w = VT_MUT(12, 1.0f - (uu + vv), 1.0f - uu - vv);                           // :182
            const float t = ((T.n[0] * cx + T.n[1] * cy) + T.n[2] * cz) * inv_det;                // :188
            bool hit = !culled && VT_MUT(8, uu > 0.0f, uu >= 0.0f) && VT_MUT(14, vv > 0.0f, vv >= 0.0f) && VT_MUT(15, w > 0.0f, w >= 0.0f) &&
                       VT_MUT(11, t > tmin, t >= tmin) && VT_MUT(2, t < tmax, t <= tmax);           // :187-189
            if (hit && alpha && (T.flags & VT_TRI_ALPHATEST)) hit = alpha_pass(hs, T.prim, uu, vv); // :196-208
            if (hit) {
                prim = T.prim; u = uu; v = vv; tmax = t;
                if (ANY_HIT) return true;
            }
        }
        return false;
    }

    template <bool ANY_HIT> void run(uint32_t* stack)
    {
        if (hs.root_leaf_count != 0) { leaf<ANY_HIT>(0, hs.root_leaf_count); return; }
        if (hs.pairs.empty()) return;
        const int ox_ = std::signbit(dx), oy_ = std::signbit(dy), oz_ = std::signbit(dz);   // octant
        const float ix = safe_inverse(dx), iy = safe_inverse(dy), iz = safe_inverse(dz);
        const float sx = -ox * ix, sy = -oy * iy, sz = -oz * iz;
        uint32_t sp = 0, node = 0;
        for (;;) {
            const vt_node_pair& P = hs.pairs[node];
            float first[2], second[2];
            for (int c = 0; c < 2; ++c) {        // both children before either leaf (the right test sees the old tmax)
                const float* b = P.child[c].bounds;
                const float e0 = b[0 + ox_] * ix + sx, e1 = b[2 + oy_] * iy + sy, e2 = b[4 + oz_] * iz + sz;
                const float x0 = b[1 - ox_] * ix + sx, x1 = b[3 - oy_] * iy + sy, x2 = b[5 - oz_] * iz + sz;
                first[c] = robust_max(e0, robust_max(e1, robust_max(e2, VT_MUT(6, e2, tmin))));
                second[c] = robust_min(x0, robust_min(x1, robust_min(x2, tmax)));
            }
            bool go[2];
            for (int c = 0; c < 2; ++c) {
                go[c] = false;
                if (VT_MUT(7, first[c] < second[c], first[c] <= second[c])) {
                    if (P.child[c].prim_count != 0) {
                        if (leaf<ANY_HIT>(P.child[c].first, P.child[c].prim_count)) return;
                    } else {
                        go[c] = true;
                    }
                }
            }
            if (go[0] && go[1]) {
                const bool swap = VT_MUT(1, first[0] >= first[1], first[0] > first[1]);   // near child first, ties keep left
                stack[sp++] = P.child[swap ? 0 : 1].first;
                node = P.child[swap ? 1 : 0].first;
            } else if (go[0]) {
                node = P.child[0].first;
            } else if (go[1]) {
                node = P.child[1].first;
            } else if (sp != 0) {
                node = stack[--sp];
            } else {
                return;
            }
        }
    }
};

template <bool ANY_HIT>
int trace(const vt_host_scene* hsw, const vt_ray* rays, uint64_t n, vt_hit* hits, uint8_t* occluded)
{
    if (!hsw) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_trace: scene is NULL");
    if (n == 0) return VT_OK;
    if (!rays || (!hits && !occluded)) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_trace: NULL buffer");
    const HostScene& hs = hsw->hs;
    bool alpha = false;
    if (hs.has_alpha) {
        if (hs.attribs.size() != hs.tris.size() || hs.alpha_mats.empty())
            return fail(VT_ERR_UNSUPPORTED, "the scene holds alpha-tested triangles (Primitives.h:196-208): call "
                                            "vt_host_scene_set_alpha before tracing it on the host");
        alpha = true;
    }
    uint32_t small[128];
    std::vector<uint32_t> big;
    uint32_t* stack = small;
    if (hs.max_depth + 1 > 128) { big.resize(hs.max_depth + 1); stack = big.data(); }
    for (uint64_t i = 0; i < n; ++i) {
        const vt_ray& r = rays[i];
        Walk w{hs, alpha, r.org[0], r.org[1], r.org[2], r.dir[0], r.dir[1], r.dir[2], r.tmin, r.tmax};
        // non-finite origin / direction: can never hit (see trace_kernels.hip start_ray) -- answered without the walk
        // the reference performs; NaN range: every comparison of the reference fails
        const bool finite = std::fabs(w.ox) <= FLT_MAX && std::fabs(w.oy) <= FLT_MAX && std::fabs(w.oz) <= FLT_MAX &&
                            std::fabs(w.dx) <= FLT_MAX && std::fabs(w.dy) <= FLT_MAX && std::fabs(w.dz) <= FLT_MAX;
        if (finite && w.tmin == w.tmin && w.tmax == w.tmax) w.template run<ANY_HIT>(stack);
        if (ANY_HIT) occluded[i] = w.prim != VT_MISS ? 1 : 0;
        else hits[i] = vt_hit{w.prim, w.prim != VT_MISS ? w.tmax : 0.f, w.u, w.v};
    }
    return VT_OK;
}

} // namespace

} // namespace vt

using namespace vt;

extern "C" {

static int stale_error(const char* who)
{
    return fail(VT_ERR_INVALID_ARG, std::string(who) + ": the device scene uploaded from this host scene has been refitted since; "
                                    "call vt_host_scene_sync first (the host copy would answer for the old geometry)");
}

int vt_host_scene_trace_closest(const vt_host_scene* hs, const vt_ray* rays, uint64_t n, vt_hit* hits)
{
    if (hs && hs->stale->load(std::memory_order_acquire) != 0) return stale_error("vt_host_scene_trace_closest");
    return trace<false>(hs, rays, n, hits, nullptr);
}

int vt_host_scene_trace_any(const vt_host_scene* hs, const vt_ray* rays, uint64_t n, uint8_t* occluded)
{
    if (hs && hs->stale->load(std::memory_order_acquire) != 0) return stale_error("vt_host_scene_trace_any");
    return trace<true>(hs, rays, n, nullptr, occluded);
}

int vt_host_scene_set_alpha(vt_host_scene* hsw, const vt_tri_attribs* attribs, uint32_t ntris, const vt_alpha_material* mats,
                            uint32_t nmats, const uint8_t* texels, uint64_t ntexels)
{
    if (!hsw) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_set_alpha: scene is NULL");
    HostScene& hs = hsw->hs;
    if (ntris != hs.tris.size()) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_set_alpha: ntris differs from the scene's triangle count");
    if ((ntris && !attribs) || nmats == 0 || !mats) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_set_alpha: NULL argument");
    for (uint32_t i = 0; i < nmats; ++i) {
        const vt_alpha_material& m = mats[i];
        if (m.filter > 1) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_set_alpha: filter must be 0 (nearest) or 1 (bilinear)");
        if ((m.width == 0) != (m.height == 0)) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_set_alpha: width and height must both be 0 or both be set");
        if (m.width && (m.offset > ntexels || uint64_t(m.width) * m.height > ntexels - m.offset || !texels))
            return fail(VT_ERR_INVALID_ARG, "vt_host_scene_set_alpha: an alpha plane lies outside the texel array");
    }
    try {
        hs.attribs.assign(attribs, attribs + ntris);
        hs.alpha_mats.assign(mats, mats + nmats);
        hs.alpha_texels.assign(texels, texels + (texels ? ntexels : 0));
    } catch (const std::bad_alloc&) {
        return fail(VT_ERR_NOMEM, "vt_host_scene_set_alpha: out of host memory");
    }
    return VT_OK;
}

} // extern "C"
