// TraceResult.cpp -- see TraceResult.h.  fp32 throughout, evaluation order as in the reference.
#include "TraceResult.h"

#include <cmath>

namespace vistrace {

int TraceResult::id = -1;

static inline float dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

TraceResult::TraceResult(const Vec3& direction, float dist, float cw, float ca, const Triangle& tri, size_t prim,
                         const Vec2& uv, const Entity& ent, const Material& mat)
    : distance(dist), primitiveIndex(prim), coneWidth(cw), coneAngle(ca), mipOverride(cw < 0.f || ca <= 0.f),
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

    uvw = Vec3{uv.x, uv.y, 1.f - uv.x - uv.y};                                     // :70
    // geometricNormal = nNorm = n / |n|, n = cross(e1, e2)   (Primitives.h:93-100, TraceResult.cpp:71)
    const Vec3 n{e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x};
    const float len = std::sqrt(dot(n, n));
    geometricNormal = Vec3{n.x / len, n.y / len, n.z / len};

    blendFactor = uvw.z * tri.alphas[0] + uvw.x * tri.alphas[1] + uvw.y * tri.alphas[2];   // :73
    texUV = Vec2{uvw.z * tri.uvs[0].x + uvw.x * tri.uvs[1].x + uvw.y * tri.uvs[2].x,       // :74
                 uvw.z * tri.uvs[0].y + uvw.x * tri.uvs[1].y + uvw.y * tri.uvs[2].y};
    entIdx = ent.id;                                                                // :76
    rawEnt = ent.rawEntity;
    submatIdx = uint32_t(tri.material);
    hitSky = (mat.surfFlags & SURF_SKY) != SURF_NONE;                               // :83
    frontFacing = dot(wo, geometricNormal) >= 0.f;                                  // :85
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

} // namespace vistrace
