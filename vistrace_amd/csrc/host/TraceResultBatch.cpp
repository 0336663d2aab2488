// TraceResultBatch.cpp -- see TraceResultBatch.h.
#include "TraceResultBatch.h"

#include "TraceResult.h"

namespace vistrace {

int TraceResultBatch::id = -1;

TraceResultBatch::TraceResultBatch(vt_batch* batch, std::shared_ptr<const SceneTables> tables)
    : mBatch(batch), mTables(std::move(tables)) {}

TraceResultBatch::~TraceResultBatch() { vt_batch_free(mBatch); }

const vt_hit* TraceResultBatch::Hits()
{
    const vt_hit* p = nullptr;
    return vt_batch_hits(mBatch, &p) == VT_OK ? p : nullptr;
}

const vt_hit_attrs* TraceResultBatch::Attrs()
{
    const vt_hit_attrs* p = nullptr;
    return vt_batch_attrs(mBatch, &p) == VT_OK ? p : nullptr;
}

const vt_hit_shade* TraceResultBatch::Shade()
{
    const vt_hit_shade* p = nullptr;
    return vt_batch_shade(mBatch, &p) == VT_OK ? p : nullptr;
}

const vt_hit_tbn* TraceResultBatch::Tbn()
{
    const vt_hit_tbn* p = nullptr;
    return vt_batch_tbn(mBatch, &p) == VT_OK ? p : nullptr;
}

const Entity& TraceResultBatch::EntityOf(const vt_hit& h) const
{
    static const Entity kNoEntity{};
    const Triangle& tri = mTables->triangles[h.prim];
    return tri.entIdx < mTables->entities.size() ? mTables->entities[tri.entIdx] : kNoEntity;       // AccelStruct.cpp:822
}

const Material& TraceResultBatch::MaterialOf(const vt_hit& h) const
{
    static const Material kNoMaterial{};
    const Triangle& tri = mTables->triangles[h.prim];
    return tri.material < mTables->materials.size() ? mTables->materials[tri.material] : kNoMaterial;   // :823
}

TraceResult* TraceResultBatch::MakeResult(uint64_t i)
{
    const vt_hit* hits = Hits();
    if (!hits || hits[i].prim == VT_MISS) return nullptr;
    const vt_ray* rays = nullptr;                       // the ray's direction comes back from the device (fetched once)
    if (vt_batch_rays(mBatch, &rays) != VT_OK || !rays) return nullptr;
    const vt_hit& h = hits[i];
    const vt_ray& r = rays[i];
    return new TraceResult(Vec3{r.dir[0], r.dir[1], r.dir[2]}, h.t, -1.f, -1.f, TriangleOf(h), h.prim, Vec2{h.u, h.v},
                           EntityOf(h), MaterialOf(h));                                             // :825-831
}

} // namespace vistrace
