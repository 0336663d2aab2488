// AccelStruct.h -- host class behind the GLua type `AccelStruct`, same public surface as the
// reference (source/objects/AccelStruct.h:61-86): PopulateAccel / Traverse / GetMaterial.
// The CPU BVH objects (mAccel, mpIntersector, mpTraverser) are replaced by a device scene
// reached through the C ABI of include/vistrace_hip.h; TraverseBatch is the additive entry
// that lets a script hand over N rays at once.
#pragma once

#include <memory>
#include <vector>

#include "LuaShim.h"
#include "Scene.h"
#include "vistrace_hip.h"

namespace vistrace {

class TraceResult;

class AccelStruct {
public:
    AccelStruct();
    ~AccelStruct();
    AccelStruct(const AccelStruct&) = delete;
    AccelStruct& operator=(const AccelStruct&) = delete;

    // Lua stack on entry: the entity table on top (source/VisTrace.cpp:776-788).  Pops it.
    void PopulateAccel(GarrysMod::Lua::ILuaBase* LUA, const World* pWorld = nullptr);
    // accel:Traverse(origin, direction, tMin=0, tMax=FLT_MAX, coneWidth=-1, coneAngle=-1)
    int Traverse(GarrysMod::Lua::ILuaBase* LUA);
    // accel:TraverseBatch(rays) -- additive, does not alter Traverse.  Two forms:
    //   rays = array of {origin, direction[, tMin[, tMax]]} tables -> array with a TraceResult or false per ray;
    //   rays = string of N packed 32-byte records {origin xyz, direction xyz, tMin, tMax} (fp32, the layout of vt_ray)
    //          -> ONE TraceResultBatch userdata: the batch stays on the device, getters take a ray index and fetch the
    //          array they need once (TraceResultBatch.h).  No per-ray table parsing, no per-hit allocation.
    //          accel:TraverseBatch(buffer, imageWidth): the rays are an image of imageWidth rays per row (camera rays) --
    //          a hint for the device's scheduling, results are the same with and without it.
    //   rays = array of such strings[, array of image widths] -> array of TraceResultBatch, all traced by ONE launch (a frame's
    //          small ray sets share one grid start and one drain: vt_batch_trace_closest_set).
    int TraverseBatch(GarrysMod::Lua::ILuaBase* LUA);

    const Material& GetMaterial(size_t i) const;

    // Non-Lua batch entry for native callers (extensions): closest hits for n rays (host walk below
    // kDeviceBatchMin rays, the device above).
    int TraceClosest(const vt_ray* rays, uint64_t n, vt_hit* hits) const;
    // batches of fewer rays are walked on the host: a launch-bound tiny batch costs 45-60 us on the device (idle
    // clocks, dependent fetches), a host-walked ray 0.5 us -- measured crossover ~110 rays on S10k-sized scenes
    // (tests/cpp/test_binding --bench prints both sides for 1 .. 4096 rays)
    static constexpr uint64_t kDeviceBatchMin = 128;
    // the two sides of that choice, callable directly (bench / tests)
    int TraceClosestHost(const vt_ray* rays, uint64_t n, vt_hit* hits) const;
    int TraceClosestDevice(const vt_ray* rays, uint64_t n, vt_hit* hits) const;

    static void SetEntityMeshSource(IEntityMeshSource* src);   // module-wide hook (see Scene.h)
    size_t TriangleCount() const { return mT->triangles.size(); }
    bool   IsBuilt() const { return mAccelBuilt; }

private:
    bool mAccelBuilt;
    vt_scene* mpScene;                      // device-resident linearised BVH + triangles
    mutable vt_host_scene* mpHostScene;     // the same records on the host: single rays are walked here (config 1); fetched on first use
    std::vector<vt_ray> mBatchRays;         // TraverseBatch scratch (members: a Lua error must not skip a destructor)
    std::vector<vt_hit> mBatchHits;
    std::vector<vt_batch*> mBatchSet;       // TraverseBatch({buffers}) scratch: the handles on their way into Lua userdata
    std::shared_ptr<SceneTables> mT;        // mTriangles / mEntities / mMaterials of the reference (AccelStruct.h:64-66); replaced per build

    void ReleaseDevice();
    bool AppendEntity(void* entityUserData);   // false: the entity table is full (65535)
    int  BuildAndUpload(vt_engine* eng);
    int  EnsureHostScene() const;
    int  UploadSideTables(const std::vector<uint8_t>& flags);
    int  TraverseBatchBuffer(GarrysMod::Lua::ILuaBase* LUA);
    int  TraverseBatchBuffers(GarrysMod::Lua::ILuaBase* LUA);
    TraceResult* MakeResult(const vt_ray& ray, const vt_hit& hit, float coneWidth, float coneAngle) const;
};

} // namespace vistrace
