// TraceResult.h -- the hit-record part of the reference's TraceResult
// (source/objects/TraceResult.h:55-72, ctor source/objects/TraceResult.cpp:45-86,
// GetPos :255-262) and the texture-free part of its shading frame: CalcFootprint (:89-103) and CalcTBN
// (:132-186) for a material WITHOUT a normal map.  Everything that samples a texture (normal maps, :139-173;
// CalcBlendFactor, CalcShadingData) needs the absent VTFParser submodule and stays out of scope.
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
    const Vec3& GetNormal()   { CalcTBN(); return normal; }     // TraceResult.cpp:264-268
    const Vec3& GetTangent()  { CalcTBN(); return tangent; }    // :269-273
    const Vec3& GetBinormal() { CalcTBN(); return binormal; }   // :274-278
    // textureLodInfo of CalcFootprint (:98-101): {triangle lod, coneWidth^2 / dot(wo, geometricNormal)^2}; false when
    // the cone is switched off (mipOverride, :54) -- the texture LoD itself needs a texture's size (Utils.h:75-78)
    bool GetTextureLodInfo(Vec2& out) { CalcFootprint(); out = textureLodInfo; return textureLodSet; }
    uint32_t GetMaterialFlags() const { return materialFlags; }   // TraceResult.cpp:309
    uint32_t GetSurfFlags() const { return surfFlags; }           // :310
    bool     HitWater() const { return water; }                   // :311

private:
    void CalcFootprint();
    void CalcTBN();

    Vec3  v[3];
    Vec3  vN[3], vT[3], vB[3];            // :58-61
    float lodOffset;                      // tri.lod (Primitives.h:103)
    Vec2  textureLodInfo;
    bool  textureLodSet = false;
    bool  tbnSet = false;
    Vec3  normal, tangent, binormal;
    bool  posSet = false;
    Vec3  pos;
    float coneWidth, coneAngle;
    bool  mipOverride;
    uint32_t materialFlags, surfFlags;
    bool  water;
};

} // namespace vistrace
