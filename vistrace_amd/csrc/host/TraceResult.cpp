// TraceResult.cpp -- see TraceResult.h.  fp32 throughout, evaluation order as in the reference.
#include "TraceResult.h"

#include <cmath>

// mutation sites (scripts/mutants_host.sh: tests/cpp/test_trace_result built with -DVT_MUTANT=<k>); the product never defines VT_MUTANT
#ifdef VT_MUTANT
#define VT_MUT(k, wrong, right) ((VT_MUTANT == (k)) ? (wrong) : (right))
#else
#define VT_MUT(k, wrong, right) (right)
#endif

namespace vistrace {

int TraceResult::id = -1;

static inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// glm::cross / glm::normalize in their scalar forms (normalize: v * inversesqrt(dot(v, v)), inversesqrt = 1 / sqrt)
static inline Vec3 cross(const Vec3& a, const Vec3& b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline Vec3 normalize(const Vec3& a)
{
    const float inv = 1.0f / std::sqrt(dot(a, a));
    return Vec3{a.x * inv, a.y * inv, a.z * inv};
}
static inline Vec3 weighted(const Vec3& uvw, const Vec3 a[3])     // uvw[2] * a[0] + uvw[0] * a[1] + uvw[1] * a[2]
{
    return Vec3{(uvw.z * a[0].x + uvw.x * a[1].x) + uvw.y * a[2].x, (uvw.z * a[0].y + uvw.x * a[1].y) + uvw.y * a[2].y,
                (uvw.z * a[0].z + uvw.x * a[1].z) + uvw.y * a[2].z};
}

TraceResult::TraceResult(const Vec3& direction, float dist, float cw, float ca, const Triangle& tri, size_t prim,
                         const Vec2& uv, const Entity& ent, const Material& mat)
    : distance(dist), primitiveIndex(prim), coneWidth(cw), coneAngle(ca), mipOverride(cw < 0.f || VT_MUT(44, ca < 0.f, ca <= 0.f)),
      materialFlags(mat.flags), surfFlags(mat.surfFlags), water(mat.water)
{
    // caller passes glm::normalize(direction) in the reference (AccelStruct.cpp:826); wo = -direction (:56)
    const float inv = 1.0f / std::sqrt(dot(direction, direction));
    wo = Vec3{-(direction.x * inv), -(direction.y * inv), -(direction.z * inv)};

    // v[0] = p0, v[1] = p1() = p0 - e1, v[2] = p2() = p0 + e2 with e1 = p0 - p1, e2 = p2 - p0
    // (Primitives.h:82,104-105; TraceResult.cpp:65-68) -- the re-derived vertices, not the inputs
    const Vec3 e1{tri.p0.x - tri.p1.x, tri.p0.y - tri.p1.y, tri.p0.z - tri.p1.z};
    const Vec3 e2{tri.p2.x - tri.p0.x, tri.p2.y - tri.p0.y, tri.p2.z - tri.p0.z};
    v[0] = tri.p0;
    v[1] = Vec3{tri.p0.x - e1.x, tri.p0.y - e1.y, tri.p0.z - e1.z};
    v[2] = Vec3{tri.p0.x + e2.x, tri.p0.y + e2.y, tri.p0.z + e2.z};

    for (int i = 0; i < 3; ++i) {                                                   // :58-62
        vN[i] = tri.normals[i];
        vT[i] = tri.tangents[i];
        vB[i] = cross(vT[i], vN[i]);
    }

    uvw = Vec3{uv.x, uv.y, VT_MUT(41, 1.f - (uv.x + uv.y), 1.f - uv.x - uv.y)};    // :70
    // geometricNormal = nNorm = n / |n|, n = cross(e1, e2)   (Primitives.h:93-100, TraceResult.cpp:71)
    const Vec3 n{e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
    const float len = std::sqrt(dot(n, n));
    geometricNormal = Vec3{n.x / len, n.y / len, n.z / len};
    // tri.lod (Primitives.h:97-103): 0.5 * log2(triUVArea / length(n))
    const float uv10x = tri.uvs[1].x - tri.uvs[0].x, uv10y = tri.uvs[1].y - tri.uvs[0].y;
    const float uv20x = tri.uvs[2].x - tri.uvs[0].x, uv20y = tri.uvs[2].y - tri.uvs[0].y;
    lodOffset = 0.5f * std::log2(std::fabs(uv10x * uv20y - uv20x * uv10y) / len);

    blendFactor = uvw.z * tri.alphas[0] + uvw.x * tri.alphas[1] + uvw.y * tri.alphas[2];   // :73
    texUV = Vec2{uvw.z * tri.uvs[0].x + VT_MUT(46, uvw.y, uvw.x) * tri.uvs[1].x + VT_MUT(46, uvw.x, uvw.y) * tri.uvs[2].x,       // :74
                 uvw.z * tri.uvs[0].y + uvw.x * tri.uvs[1].y + uvw.y * tri.uvs[2].y};
    entIdx = ent.id;                                                                // :76
    rawEnt = ent.rawEntity;
    submatIdx = uint32_t(tri.material);
    hitSky = (mat.surfFlags & SURF_SKY) != SURF_NONE;                               // :83
    frontFacing = VT_MUT(42, dot(wo, geometricNormal) > 0.f, dot(wo, geometricNormal) >= 0.f);   // :85
}

const Vec3& TraceResult::GetPos()                                                   // :255-262
{
    if (!posSet) {
        pos = Vec3{(uvw.z * v[0].x + uvw.x * v[1].x) + uvw.y * v[2].x,
                   (uvw.z * v[0].y + uvw.x * v[1].y) + uvw.y * v[2].y,
                   (uvw.z * v[0].z + uvw.x * v[1].z) + uvw.y * v[2].z};
        posSet = true;
    }
    return pos;
}

void TraceResult::CalcFootprint()                                                   // :89-103 (Ray Tracing Gems cone)
{
    if (textureLodSet || mipOverride) return;
    coneWidth = coneAngle * distance + coneWidth;                                   // the cone at the hit point
    const float normalTerm = dot(wo, geometricNormal);
    textureLodInfo = Vec2{lodOffset, (coneWidth * coneWidth) / (normalTerm * normalTerm)};
    textureLodSet = true;
}

void TraceResult::CalcTBN()                                                         // :132-186, material.normalMap == nullptr
{
    if (tbnSet) return;
    normal = normalize(weighted(uvw, vN));
    tangent = normalize(weighted(uvw, vT));
    binormal = normalize(weighted(uvw, vB));

    const float kCosThetaThreshold = 0.1f;                                          // :175
    const float cosTheta = std::fabs(dot(wo, normal));
    if (VT_MUT(43, cosTheta < kCosThetaThreshold, cosTheta <= kCosThetaThreshold)) {
        float t = cosTheta * (1.f / kCosThetaThreshold);                            // saturate
        t = t < 0.f ? 0.f : t; t = t > 1.f ? 1.f : t;
        const float s = 1.f - t;                                                    // lerp(x, y, a) = x * (1 - a) + y * a
        if (VT_MUT(48, true, false)) t = s;                                         // (mutant 48: both weights the geometric normal's)
        normal = normalize(Vec3{geometricNormal.x * s + normal.x * t, geometricNormal.y * s + normal.y * t,
                                geometricNormal.z * s + normal.z * t});
        const float tn = dot(tangent, normal);
        tangent = normalize(Vec3{tangent.x - normal.x * tn, tangent.y - normal.y * tn, tangent.z - normal.z * tn});
        binormal = VT_MUT(45, cross(normal, tangent), cross(tangent, normal));
    }
    tbnSet = true;
}

} // namespace vistrace
