// TraceResult.h -- the hit-record part of the reference's TraceResult
// (source/objects/TraceResult.h:55-72, ctor source/objects/TraceResult.cpp:45-86,
// GetPos :255-262).  Shading data (TraceResult.cpp:89-253: cone footprint, TBN, texture
// sampling) is out of scope.
#pragma once

#include "Scene.h"

namespace vistrace {

class TraceResult {
public:
    static int id;                        // Lua user-type id (TraceResult.cpp:9)

    float    distance;
    Vec3     wo;                          // -normalize(direction)
    Vec3     geometricNormal;
    Vec3     uvw;                         // (u, v, 1-u-v)
    Vec2     texUV;
    float    blendFactor;
    uint32_t entIdx;
    void*    rawEnt;
    uint32_t submatIdx;
    bool     hitSky = false;              // (material.surfFlags & SURF::SKY) != NONE   (TraceResult.cpp:83)
    bool     frontFacing;
    size_t   primitiveIndex;              // hit->primitive_index (AccelStruct.cpp:821), kept for batch users

    // `mat` = mMaterials[tri.material] (AccelStruct.cpp:823); the reference copies the whole Material, the getters
    // on this path read three of its fields
    TraceResult(const Vec3& direction, float distance, float coneWidth, float coneAngle, const Triangle& tri,
                size_t primitiveIndex, const Vec2& uv, const Entity& ent, const Material& mat);

    const Vec3& GetPos();
    uint32_t GetMaterialFlags() const { return materialFlags; }   // TraceResult.cpp:309
    uint32_t GetSurfFlags() const { return surfFlags; }           // :310
    bool     HitWater() const { return water; }                   // :311

private:
    Vec3  v[3];
    bool  posSet = false;
    Vec3  pos;
    float coneWidth, coneAngle;
    bool  mipOverride;
    uint32_t materialFlags, surfFlags;
    bool  water;
};

} // namespace vistrace
