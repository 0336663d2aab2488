// TraceResultBatch.h -- what `accel:TraverseBatch(buffer)` hands back: ONE userdata for the whole batch.
//
// The reference answers one ray per Lua call with one heap-allocated TraceResult (source/objects/AccelStruct.cpp:
// 825-831, ctor source/objects/TraceResult.cpp:45-86).  Carried over to a batch that is N table parses, N constructor
// calls and N allocations on the host around a trace that takes a fraction of a millisecond per million rays.  Here
// the batch stays where it was traced (vt_batch, include/vistrace_hip.h): hit records, the TraceResult core
// (Pos, Distance, GeometricNormal, Barycentric, Incident, FrontFacing: vt_hit_attrs) and the shading part (TextureUV,
// blend factor, entity id, material: vt_hit_shade; Normal, Tangent, Binormal: vt_hit_tbn) are materialised by device kernels, and a getter downloads the one
// array it reads on first use.  Getters mirror TraceResult's, with the ray's 1-based index as their argument.
#pragma once

#include <memory>
#include <vector>

#include "Scene.h"
#include "vistrace_hip.h"

namespace vistrace {

class TraceResult;

class TraceResultBatch {
public:
    static int id;                                      // Lua user-type id

    // takes ownership of `batch`
    TraceResultBatch(vt_batch* batch, std::shared_ptr<const SceneTables> tables);
    ~TraceResultBatch();
    TraceResultBatch(const TraceResultBatch&) = delete;
    TraceResultBatch& operator=(const TraceResultBatch&) = delete;

    uint64_t Count() const { return vt_batch_count(mBatch); }
    // arrays of Count() records, fetched from the device on first use; NULL (and vt_last_error) on failure
    const vt_hit*       Hits();
    const vt_hit_attrs* Attrs();
    const vt_hit_shade* Shade();
    const vt_hit_tbn*   Tbn();                          // shading frame (no normal map), cone off
    const SceneTables&  Tables() const { return *mTables; }
    // the triangle / entity / material behind hit i (i must be a hit)
    const Triangle& TriangleOf(const vt_hit& h) const { return mTables->triangles[h.prim]; }
    const Entity&   EntityOf(const vt_hit& h) const;
    const Material& MaterialOf(const vt_hit& h) const;
    // the reference's per-ray object for ray i, or NULL for a miss (caller owns it)
    TraceResult* MakeResult(uint64_t i);

private:
    vt_batch* mBatch;
    std::shared_ptr<const SceneTables> mTables;
};

} // namespace vistrace
