// Scene.h -- host-side scene records that the Tracing API keeps per acceleration structure.
// Mirrors the members the hot path touches in the reference:
//   Triangle  <- TriangleBackfaceCull<float>  source/objects/Primitives.h:44-118
//   Material  <- Material / MaterialFlags     source/objects/Material.h:31-72 (flag bits only)
//   Entity    <- struct Entity                source/objects/AccelStruct.h:33-40
//   World     <- class World                  source/objects/AccelStruct.h:42-60
// Asset parsing, skinning and texture data are out of scope: entity
// geometry arrives already in world space through IEntityMeshSource.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace vistrace {

struct Vec2 { float x = 0, y = 0; };
struct Vec3 { float x = 0, y = 0, z = 0; };

enum MaterialFlags : uint32_t {           // values of source/objects/Material.h:31-64 used here
    MATFLAG_NONE      = 0,
    MATFLAG_ALPHATEST = 256,              // Material.h:42
    MATFLAG_NOCULL    = 8192,             // Material.h:47
};

// BSPEnums::SURF bits (BSPParser is an absent submodule; values are the Source engine's bspflags.h): only SKY is
// read on this path (source/objects/TraceResult.cpp:83)
enum SurfFlags : uint32_t { SURF_NONE = 0, SURF_SKY = 0x4 };

struct Material {
    std::string path;
    uint32_t    flags = MATFLAG_NONE;
    uint32_t    surfFlags = SURF_NONE;                          // Material.h:120 (BSPEnums::SURF, set for world brushes: AccelStruct.cpp:392)
    bool        water = false;                                  // Material.h (TraceResult::HitWater, TraceResult.cpp:311)
    // what the alpha test of Primitives.h:196-208 reads (Material.h:81-122).  The decoded mip-0 alpha plane of
    // the base texture stands in for `const IVTFTexture* baseTexture`: VTF decoding stays with the module.
    float    baseTexMat[2][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}};   // glm::mat2x4 baseTexMat: row r = transform[r]
    float    texScale = 1.f;
    float    alphatestreference = 0.5f;                         // Material.h:122
    uint32_t alphaWidth = 0, alphaHeight = 0;                   // 0 x 0: no texture, the hit is kept
    bool     alphaBilinear = false;                             // lookup: nearest texel, or bilinear
    std::vector<uint8_t> baseAlpha;                             // alphaWidth x alphaHeight, row-major
};

struct Triangle {
    Vec3     p0, p1, p2;                  // world-space vertices (the device record re-derives e1, e2, n)
    Vec3     normals[3];                  // Primitives.h:62
    Vec3     tangents[3];                 // Primitives.h:63
    Vec2     uvs[3];
    float    alphas[3] = {1, 1, 1};
    bool     oneSided = false;            // Primitives.h:57 (world brushes: AccelStruct.cpp:408-410)
    size_t   material = 0;
    uint16_t entIdx = 0;                  // Primitives.h:60
};

struct Entity {
    void*    rawEntity = nullptr;
    uint32_t id = 0;
};

// What an AccelStruct keeps per build; shared (read-only after the build) with the TraceResultBatch objects traced from
// it, which may outlive a Rebuild or the accel itself.
struct SceneTables {
    std::vector<Triangle> triangles;
    std::vector<Entity>   entities;
    std::vector<Material> materials;
};

struct World {
    std::vector<Triangle> triangles;
    std::vector<Entity>   entities;
    std::vector<Material> materials;
};

// Engine-coupled scene ingest (source/objects/AccelStruct.cpp:567-758: model lookup, bones,
// skinning) lives behind this hook; the module supplies the real one, tests a fake.
class IEntityMeshSource {
public:
    virtual ~IEntityMeshSource() {}
    // Append the entity's world-space triangles and materials. `entityUserData` is what the
    // Lua stack holds for the entity. Triangle::material indexes the materials appended here
    // (relative); the caller rebases material and entIdx. Return false to skip the entity.
    virtual bool AppendEntity(void* entityUserData, Entity& outEntity, std::vector<Triangle>& tris,
                              std::vector<Material>& materials) = 0;
};

} // namespace vistrace
