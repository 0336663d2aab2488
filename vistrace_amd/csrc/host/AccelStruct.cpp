// AccelStruct.cpp -- see AccelStruct.h.  Follows source/objects/AccelStruct.cpp:533-838 for
// argument handling, defaults, error messages and result selection; the BVH build/traverse
// calls at :763-773 and :818 go through the C ABI (CPU build -> linearise -> upload; GPU trace).
#include "AccelStruct.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "TraceResult.h"
#include "TraceResultBatch.h"

// mutation sites (scripts/mutants_binding.sh: tests/cpp/test_binding built with -DVT_MUTANT=<k>); the product never defines VT_MUTANT
#ifdef VT_MUTANT
#define VT_MUT(k, wrong, right) ((VT_MUTANT == (k)) ? (wrong) : (right))
#else
#define VT_MUT(k, wrong, right) (right)
#endif

using namespace GarrysMod::Lua;

namespace vistrace {

namespace {

IEntityMeshSource* g_meshSource = nullptr;
vt_engine* g_engine = nullptr;

// one engine per process, opened on first use (device from VISTRACE_DEVICE, default 0)
// One engine per process, opened on first use.  VISTRACE_DEVICES="0,1,2,3" forms a single-process multi-GPU group
// (BVH replicated on every device, TraverseBatch's rays sharded across them); otherwise one device, VISTRACE_DEVICE
// (default 0).
vt_engine* Engine(ILuaBase* LUA)
{
    if (!g_engine) {
        int devs[64];
        int ndev = 0;
        if (const char* list = std::getenv("VISTRACE_DEVICES")) {
            for (const char* p = list; *p && ndev < 64;) {
                char* end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) break;
                devs[ndev++] = int(v);
                p = (*end == ',') ? end + 1 : end;
            }
        }
        if (ndev == 0) {
            const char* one = std::getenv("VISTRACE_DEVICE");
            devs[ndev++] = one ? std::atoi(one) : 0;
        }
        const int rc = ndev > 1 ? vt_engine_open_multi(devs, ndev, &g_engine) : vt_engine_open(devs[0], &g_engine);
        if (rc != VT_OK) {
            g_engine = nullptr;
            static thread_local char msg[512];
            std::snprintf(msg, sizeof(msg), "VisTrace: cannot open the HIP device: %s", vt_last_error());
            LUA->ThrowError(msg);
        }
    }
    return g_engine;
}

} // namespace

void AccelStruct::SetEntityMeshSource(IEntityMeshSource* src) { g_meshSource = src; }

AccelStruct::AccelStruct() : mAccelBuilt(false), mpScene(nullptr), mpHostScene(nullptr), mT(std::make_shared<SceneTables>()) {}

AccelStruct::~AccelStruct() { ReleaseDevice(); }                       // AccelStruct.cpp:525-531

void AccelStruct::ReleaseDevice()
{
    if (mpScene) vt_scene_free(mpScene);
    mpScene = nullptr;
    if (mpHostScene) vt_host_scene_free(mpHostScene);
    mpHostScene = nullptr;
}

// the records on the host (pairs, leaf-ordered triangles) for the rays that are walked here: fetched from the device the first time
// one is asked for -- a caller that only traces batches never pays for the copy
int AccelStruct::EnsureHostScene() const
{
    if (mpHostScene) return VT_OK;
    if (!mpScene) return VT_ERR_INVALID_ARG;
    return vt_host_scene_download(mpScene, &mpHostScene);
}

void AccelStruct::PopulateAccel(ILuaBase* LUA, const World* pWorld)
{
    // ThrowError does not return and (in the real module: longjmp) does not unwind C++ frames, so every raise below
    // happens while no local with a destructor is alive: entities are appended in AppendEntity(), the engine is opened
    // between the entity loop and the build, and the build runs in BuildAndUpload(), whose vectors are gone when its
    // status is turned into a Lua error.
    // tear down the previous build (AccelStruct.cpp:537-550)
    mAccelBuilt = false;
    ReleaseDevice();
    mT = std::make_shared<SceneTables>();      // batches traced from the previous build keep their tables

    if (pWorld) {                                                       // :552-555
        mT->triangles = pWorld->triangles;
        mT->entities = pWorld->entities;
        mT->materials = pWorld->materials;
    }

    // iterate the entity table on top of the stack (:567-758)
    LUA->PushNil();
    while (LUA->Next(-2) != 0) {
        if (!LUA->IsType(-1, Type::Entity)) LUA->ThrowError("Build list must only contain entities");   // :575
        // entIdx is a uint16_t: the limit counts the entities the mesh source actually delivers, not the ones it skips;
        // the raise happens here, after AppendEntity's locals are gone (ThrowError does not unwind C++ frames)
        if (g_meshSource && !AppendEntity(LUA->GetUserdataRaw(-1, Type::Entity)))
            LUA->ThrowError("Too many entities in build list");
        LUA->Pop();                                                     // pop value, keep key
    }
    LUA->Pop();                                                         // pop entity table (:760)

    // Build BVH (:762-775): CPU PLOC + leaf collapse, re-pack, upload once per Rebuild
    vt_engine* eng = Engine(LUA);
    if (BuildAndUpload(eng) != VT_OK) {
        static thread_local char msg[512];
        std::snprintf(msg, sizeof(msg), "VisTrace: acceleration structure build failed: %s", vt_last_error());
        LUA->ThrowError(msg);
    }
    mAccelBuilt = true;
}

bool AccelStruct::AppendEntity(void* entityUserData)
{
    Entity ent;
    std::vector<Triangle> tris;
    std::vector<Material> mats;
    if (!g_meshSource->AppendEntity(entityUserData, ent, tris, mats)) return true;     // skipped by the source: not counted
    if (mT->entities.size() >= 65535) return false;
    const size_t matBase = mT->materials.size();
    const uint16_t entIdx = uint16_t(mT->entities.size());
    for (Triangle& t : tris) { t.material += matBase; t.entIdx = entIdx; }
    mT->materials.insert(mT->materials.end(), mats.begin(), mats.end());
    mT->triangles.insert(mT->triangles.end(), tris.begin(), tris.end());
    mT->entities.push_back(ent);
    return true;
}

int AccelStruct::BuildAndUpload(vt_engine* eng)
{
    const uint32_t n = uint32_t(mT->triangles.size());
    std::vector<float> verts(size_t(n) * 9);
    std::vector<uint8_t> flags(n);
    for (uint32_t i = 0; i < n; ++i) {
        const Triangle& t = mT->triangles[i];
        const float v[9] = {t.p0.x, t.p0.y, t.p0.z, t.p1.x, t.p1.y, t.p1.z, t.p2.x, t.p2.y, t.p2.z};
        std::memcpy(&verts[size_t(i) * 9], v, sizeof(v));
        const uint32_t mflags = t.material < mT->materials.size() ? mT->materials[t.material].flags : 0u;
        uint8_t f = 0;
        if (t.oneSided && !(mflags & MATFLAG_NOCULL)) f |= VT_TRI_CULL_BACKFACE;   // Primitives.h:174
        if (mflags & MATFLAG_ALPHATEST) f |= VT_TRI_ALPHATEST;                     // Primitives.h:196
        flags[i] = f;
    }
    std::vector<vt_tri64> recs(n);
    vt_bvh* bvh = nullptr;
    int rc = vt_tris_setup(verts.data(), flags.data(), n, recs.data());
    if (rc == VT_OK) rc = vt_bvh_build(recs.data(), n, 0, &bvh);
    // the tree and the records go up as they are; the device re-packs them (pairs in depth-first order, triangles in leaf order,
    // index tables) on every device of the engine.  The host copy that single rays are walked on is fetched on first use.
    if (rc == VT_OK) rc = vt_scene_upload_tree(eng, bvh, recs.data(), n, &mpScene);
    if (bvh) vt_bvh_free(bvh);
    if (rc == VT_OK) rc = UploadSideTables(flags);
    if (rc != VT_OK) ReleaseDevice();
    return rc;
}

// Per-triangle side table (uvs, vertex alphas, entity id, material index): what the device needs for the shading part of
// a batch's results (TraceResult.cpp:73-78, vt_hit_shade) and the per-vertex normals / tangents of its shading frame
// (TraceResult.cpp:132-186, vt_hit_tbn) -- always.  The side data of the in-kernel alpha test
// (Primitives.h:196-208: per-material transform / reference / alpha plane) only when a material carries the flag.
int AccelStruct::UploadSideTables(const std::vector<uint8_t>& flags)
{
    bool any = false;
    for (uint8_t f : flags) any = any || (f & VT_TRI_ALPHATEST);
    std::vector<vt_tri_attribs> attribs(mT->triangles.size());
    for (size_t i = 0; i < mT->triangles.size(); ++i) {
        const Triangle& t = mT->triangles[i];
        vt_tri_attribs& a = attribs[i];
        for (int k = 0; k < 3; ++k) { a.uv[k][0] = t.uvs[k].x; a.uv[k][1] = t.uvs[k].y; a.alpha[k] = t.alphas[k]; }
        a.ent_id = t.entIdx < mT->entities.size() ? mT->entities[t.entIdx].id : 0u;
        a.material = uint32_t(t.material);
        a.pad = 0;
    }
    int rc = vt_scene_set_tri_attribs(mpScene, attribs.data(), uint32_t(attribs.size()));
    if (rc == VT_OK) {                                       // per-vertex normals / tangents: the shading frame of a batch's hits
        std::vector<vt_tri_frame> frames(mT->triangles.size());
        for (size_t i = 0; i < mT->triangles.size(); ++i) {
            const Triangle& t = mT->triangles[i];
            for (int k = 0; k < 3; ++k) {
                frames[i].normal[k][0] = t.normals[k].x; frames[i].normal[k][1] = t.normals[k].y; frames[i].normal[k][2] = t.normals[k].z;
                frames[i].tangent[k][0] = t.tangents[k].x; frames[i].tangent[k][1] = t.tangents[k].y; frames[i].tangent[k][2] = t.tangents[k].z;
            }
        }
        rc = vt_scene_set_tri_frames(mpScene, frames.data(), uint32_t(frames.size()));
    }
    if (rc != VT_OK || !any) return rc;
    std::vector<vt_alpha_material> mats(mT->materials.size());
    std::vector<uint8_t> texels;
    for (size_t i = 0; i < mT->materials.size(); ++i) {
        const Material& m = mT->materials[i];
        vt_alpha_material& o = mats[i];
        std::memcpy(o.tex_mat, m.baseTexMat, sizeof(o.tex_mat));
        o.tex_scale = m.texScale;
        o.alpha_ref = m.alphatestreference;
        const bool has_plane = (m.flags & MATFLAG_ALPHATEST) && m.alphaWidth && m.alphaHeight &&
                               m.baseAlpha.size() >= size_t(m.alphaWidth) * m.alphaHeight;
        o.width = has_plane ? m.alphaWidth : 0;
        o.height = has_plane ? m.alphaHeight : 0;
        o.filter = m.alphaBilinear ? 1u : 0u;
        o.pad = 0;
        o.offset = texels.size();
        if (has_plane) texels.insert(texels.end(), m.baseAlpha.begin(), m.baseAlpha.begin() + size_t(m.alphaWidth) * m.alphaHeight);
    }
    rc = vt_scene_set_alpha(mpScene, mats.data(), uint32_t(mats.size()), texels.data(), texels.size());
    if (rc == VT_OK) rc = EnsureHostScene();                // the same tables for the host walk: such a scene fetches its host copy now
    if (rc == VT_OK)
        rc = vt_host_scene_set_alpha(mpHostScene, attribs.data(), uint32_t(attribs.size()), mats.data(), uint32_t(mats.size()),
                                     texels.data(), texels.size());
    return rc;
}

TraceResult* AccelStruct::MakeResult(const vt_ray& ray, const vt_hit& hit, float coneWidth, float coneAngle) const
{
    const Triangle& tri = mT->triangles[hit.prim];                          // :821
    static const Entity kNoEntity{};
    const Entity& ent = tri.entIdx < mT->entities.size() ? mT->entities[tri.entIdx] : kNoEntity;   // :822
    static const Material kNoMaterial{};
    const Material& mat = tri.material < mT->materials.size() ? mT->materials[tri.material] : kNoMaterial;   // :823
    return new TraceResult(Vec3{ray.dir[0], ray.dir[1], ray.dir[2]}, hit.t, coneWidth, coneAngle, tri, hit.prim,
                           Vec2{hit.u, hit.v}, ent, mat);                // :825-831
}

int AccelStruct::Traverse(ILuaBase* LUA)
{
    if (!mAccelBuilt)
        LUA->ThrowError("Unable to perform traversal, acceleration structure invalid (use AccelStruct:Rebuild to rebuild it)");
    const int numArgs = LUA->Top();

    // (origin, direction) are mandatory Vectors; the four numbers are optional and nil keeps the default
    LUA->CheckType(2, Type::Vector);
    LUA->CheckType(3, Type::Vector);
    const Vector origin = LUA->GetVector(2), direction = LUA->GetVector(3);
    auto optional_number = [&](int pos, float dflt) {
        return (numArgs >= pos && !LUA->IsType(pos, Type::Nil)) ? static_cast<float>(LUA->CheckNumber(pos)) : dflt;
    };
    const float tMin = optional_number(4, 0.f);            // AccelStruct.cpp:790-791
    const float tMax = optional_number(5, FLT_MAX);        // :793-794
    const float coneWidth = optional_number(6, -1.f);      // :796-797 (negative: mip 0 only)
    const float coneAngle = optional_number(7, -1.f);      // :799-800

    // same checks, order and messages as :802-806
    if (VT_MUT(96, coneWidth > 0, coneWidth >= 0) && VT_MUT(91, coneAngle < 0.f, coneAngle <= 0.f)) LUA->ThrowError("Valid cone width but invalid cone angle passed");
    if (VT_MUT(92, coneWidth <= 0, coneWidth < 0) && coneAngle > 0.f) LUA->ThrowError("Valid cone angle but invalid cone width passed");
    if (tMin < 0.f) LUA->ArgError(4, "tMin cannot be less than 0");
    if (VT_MUT(94, tMax < tMin, tMax <= tMin)) LUA->ArgError(5, "tMax must be greater than tMin");

    LUA->Pop(LUA->Top());

    // One ray per call (the reference's `mpTraverser->traverse(ray, *mpIntersector)`, :818): walked on the host copy
    // of the linearised scene -- 1-2 us against ~20 us for a launch-bound lone ray on the device (BASELINE config 1).
    const vt_ray ray{{origin.x, origin.y, origin.z}, {direction.x, direction.y, direction.z}, tMin, tMax};
    vt_hit hit;
    if (EnsureHostScene() != VT_OK || vt_host_scene_trace_closest(mpHostScene, &ray, 1, &hit) != VT_OK) {
        static thread_local char msg[512];
        std::snprintf(msg, sizeof(msg), "VisTrace: traversal failed: %s", vt_last_error());
        LUA->ThrowError(msg);
    }
    if (hit.prim != VT_MISS) {
        LUA->PushUserType_Value(MakeResult(ray, hit, coneWidth, coneAngle), TraceResult::id);
        return 1;
    }
    return 0;
}

int AccelStruct::TraceClosest(const vt_ray* rays, uint64_t n, vt_hit* hits) const
{
    if (!mAccelBuilt) return VT_ERR_INVALID_ARG;
    // below the crossover a launch costs more than walking the rays on this thread (measured: tests/cpp --bench)
    return n < kDeviceBatchMin ? TraceClosestHost(rays, n, hits) : TraceClosestDevice(rays, n, hits);
}

int AccelStruct::TraceClosestHost(const vt_ray* rays, uint64_t n, vt_hit* hits) const
{
    if (!mAccelBuilt) return VT_ERR_INVALID_ARG;
    const int rc = EnsureHostScene();
    return rc != VT_OK ? rc : vt_host_scene_trace_closest(mpHostScene, rays, n, hits);
}

int AccelStruct::TraceClosestDevice(const vt_ray* rays, uint64_t n, vt_hit* hits) const
{
    if (!mAccelBuilt) return VT_ERR_INVALID_ARG;
    return vt_trace_closest(mpScene, rays, n, hits);
}

int AccelStruct::TraverseBatch(ILuaBase* LUA)
{
    if (!mAccelBuilt)
        LUA->ThrowError("Unable to perform traversal, acceleration structure invalid (use AccelStruct:Rebuild to rebuild it)");
    if (LUA->IsType(2, Type::String)) return TraverseBatchBuffer(LUA);
    LUA->CheckType(2, Type::Table);
    {   // a table of packed buffers = a SET of batches (one merged launch), a table of ray tables = the per-ray form below
        LUA->PushNumber(1.0);
        LUA->GetTable(2);
        const bool buffers = LUA->IsType(-1, Type::String);
        LUA->Pop();
        if (buffers) return TraverseBatchBuffers(LUA);
    }
    // rays[i] = { origin, direction, tMin?, tMax? } with the defaults and checks of Traverse.  Fields are read BY INDEX
    // (rays[i][k]): lua_next skips nil values and promises no order, so {o, d, nil, tMax} must not shift tMax to field 3.
    // First pass validates and counts (every raise happens before the ray array exists), second pass fills.
    size_t count = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) mBatchRays.resize(count);
        for (size_t i = 0;; ++i) {
            LUA->PushNumber(double(i + 1));
            LUA->GetTable(2);
            if (LUA->IsType(-1, Type::Nil)) { LUA->Pop(); if (pass == 0) count = i; break; }
            if (pass == 0) LUA->CheckType(-1, Type::Table);
            const int rt = LUA->Top();
            vt_ray r{{0, 0, 0}, {0, 0, 0}, 0.f, FLT_MAX};
            for (int k = 1; k <= 4; ++k) {
                LUA->PushNumber(double(k));
                LUA->GetTable(rt);
                if (k <= 2) {
                    if (pass == 0 && !LUA->IsType(-1, Type::Vector))
                        LUA->ThrowError("Each ray must be a table {origin, direction[, tMin[, tMax]]}");
                    const Vector v = LUA->GetVector(-1);
                    float* dst = k == 1 ? r.org : r.dir;
                    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
                } else if (!LUA->IsType(-1, Type::Nil)) {               // nil keeps the default, as in Traverse
                    const float f = static_cast<float>(pass == 0 ? LUA->CheckNumber(-1) : LUA->GetNumber(-1));
                    if (k == 3) r.tmin = f; else r.tmax = f;
                }
                LUA->Pop();
            }
            if (pass == 0) {
                if (r.tmin < 0.f) LUA->ThrowError("tMin cannot be less than 0");
                if (VT_MUT(95, r.tmax < r.tmin, r.tmax <= r.tmin)) LUA->ThrowError("tMax must be greater than tMin");
            } else {
                mBatchRays[i] = r;
            }
            LUA->Pop();
        }
    }
    LUA->Pop(LUA->Top());

    mBatchHits.resize(mBatchRays.size());
    if (TraceClosest(mBatchRays.data(), mBatchRays.size(), mBatchHits.data()) != VT_OK) {
        static thread_local char msg[512];
        std::snprintf(msg, sizeof(msg), "VisTrace: traversal failed: %s", vt_last_error());
        LUA->ThrowError(msg);
    }
    LUA->CreateTable();
    for (size_t i = 0; i < mBatchHits.size(); ++i) {
        LUA->PushNumber(double(i + 1));
        if (mBatchHits[i].prim != VT_MISS) LUA->PushUserType_Value(MakeResult(mBatchRays[i], mBatchHits[i], -1.f, -1.f), TraceResult::id);
        else LUA->PushBool(false);
        LUA->SetTable(-3);
    }
    return 1;
}

// accel:TraverseBatch(buffer): N packed vt_ray records in, ONE TraceResultBatch out (TraceResultBatch.h).  Same range
// checks as Traverse (AccelStruct.cpp:805-806) on every ray -- made by the engine's staging copy while the batch streams to
// the device (vt_batch_trace_closest_ex, VT_BATCH_CHECK_RANGES), not in a pass of their own; no per-ray Lua traffic.
int AccelStruct::TraverseBatchBuffer(ILuaBase* LUA)
{
    unsigned int len = 0;
    const char* bytes = LUA->GetString(2, &len);
    if (!bytes || len % sizeof(vt_ray) != 0) LUA->ArgError(2, "ray buffer must hold whole 32-byte records {origin, direction, tMin, tMax}");
    const size_t n = len / sizeof(vt_ray);
    // optional: the rays are an image in row-major order with this many rays per row (camera rays): the engine then walks them
    // as pixel tiles (vt_batch_desc::ray_image_width; scheduling only)
    uint32_t width = 0;
    if (LUA->Top() >= 3 && !LUA->IsType(3, Type::Nil)) {
        const double w = LUA->CheckNumber(3);
        if (!(w >= 0.0 && w <= 1048576.0) || w != std::floor(w)) LUA->ArgError(3, "imageWidth must be a whole number of rays per row");
        width = uint32_t(w);
    }
    // the rays go to the device straight from the string (still on the Lua stack); nothing is kept on the host.  The hit records
    // come back behind the trace (VT_BATCH_FETCH_HITS): a script that asks for a batch reads at least those.
    vt_batch* batch = nullptr;
    uint64_t bad = n;
    const int rc = vt_batch_trace_closest_ex(mpScene, reinterpret_cast<const vt_ray*>(bytes), n, width,
                                             VT_BATCH_CHECK_RANGES | VT_BATCH_FETCH_HITS, &bad, &batch);
    if (rc != VT_OK && bad < n) {                                // Lua strings carry no alignment promise: copy the two fields out
        float range[2];
        std::memcpy(range, bytes + bad * sizeof(vt_ray) + 24, sizeof(range));
        if (range[0] < 0.f) LUA->ThrowError("tMin cannot be less than 0");
        LUA->ThrowError("tMax must be greater than tMin");
    }
    if (rc != VT_OK) {
        static thread_local char msg[512];
        std::snprintf(msg, sizeof(msg), "VisTrace: traversal failed: %s", vt_last_error());
        LUA->ThrowError(msg);
    }
    LUA->Pop(LUA->Top());
    LUA->PushUserType_Value(new TraceResultBatch(batch, mT), TraceResultBatch::id);
    return 1;
}

// accel:TraverseBatch({buffer1, buffer2, ...}[, {imageWidth1, ...}]): several packed ray buffers (a frame's ray sets: per light,
// per tile, per entity) -> a table of TraceResultBatch objects, traced by ONE merged launch (vt_batch_trace_closest_set): a
// launch costs ~0.3 ms beyond its rays, which a script with 16 small sets would otherwise pay 16 times.
int AccelStruct::TraverseBatchBuffers(ILuaBase* LUA)
{
    // Buffer by buffer: a string is staged and uploaded while it sits on the stack, then popped (a C function owns few stack
    // slots, and only a string on the stack is certain to stay).  A Lua error must not leak the set: it is aborted first.
    vt_batch_set* set = nullptr;
    if (vt_batch_set_begin(mpScene, VT_BATCH_CHECK_RANGES | VT_BATCH_FETCH_HITS, &set) != VT_OK) {
        static thread_local char msg[512];
        std::snprintf(msg, sizeof(msg), "VisTrace: traversal failed: %s", vt_last_error());
        LUA->ThrowError(msg);
    }
    const bool have_widths = LUA->IsType(3, Type::Table);
    for (size_t i = 0;; ++i) {
        LUA->PushNumber(double(i + 1));
        LUA->GetTable(2);
        if (LUA->IsType(-1, Type::Nil)) { LUA->Pop(); break; }
        const char* problem = nullptr;
        int problem_arg = 2;
        unsigned int len = 0;
        const char* bytes = nullptr;
        uint32_t width = 0;
        if (!LUA->IsType(-1, Type::String)) problem = "a table of ray buffers must hold strings only";
        else {
            bytes = LUA->GetString(-1, &len);
            if (!bytes || len % sizeof(vt_ray) != 0) problem = "ray buffer must hold whole 32-byte records {origin, direction, tMin, tMax}";
        }
        if (!problem && have_widths) {
            LUA->PushNumber(double(i + 1));
            LUA->GetTable(3);
            if (!LUA->IsType(-1, Type::Nil)) {
                const double w = LUA->IsType(-1, Type::Number) ? LUA->GetNumber(-1) : -1.0;
                if (!(w >= 0.0 && w <= 1048576.0) || w != std::floor(w)) { problem = "imageWidth must be a whole number of rays per row"; problem_arg = 3; }
                else width = uint32_t(w);
            }
            LUA->Pop();
        }
        if (problem) { vt_batch_set_abort(set); LUA->ArgError(problem_arg, problem); }
        const uint64_t n = len / sizeof(vt_ray);
        uint64_t bad = n;
        const int rc = vt_batch_set_add(set, reinterpret_cast<const vt_ray*>(bytes), n, width, &bad);
        if (rc != VT_OK) {
            vt_batch_set_abort(set);
            if (bad < n) {                                       // Lua strings carry no alignment promise: copy the two fields out
                float range[2];
                std::memcpy(range, bytes + bad * sizeof(vt_ray) + 24, sizeof(range));
                if (range[0] < 0.f) LUA->ThrowError("tMin cannot be less than 0");
                LUA->ThrowError("tMax must be greater than tMin");
            }
            static thread_local char msg[512];
            std::snprintf(msg, sizeof(msg), "VisTrace: traversal failed: %s", vt_last_error());
            LUA->ThrowError(msg);
        }
        LUA->Pop();
    }
    std::vector<vt_batch*>& batches = mBatchSet;                 // a member: a Lua error must not skip a destructor
    batches.assign(vt_batch_set_count(set), nullptr);
    if (vt_batch_set_trace(set, batches.data()) != VT_OK) {      // consumes the set either way
        static thread_local char msg[512];
        std::snprintf(msg, sizeof(msg), "VisTrace: traversal failed: %s", vt_last_error());
        LUA->ThrowError(msg);
    }
    LUA->Pop(LUA->Top());
    LUA->CreateTable();
    for (size_t i = 0; i < batches.size(); ++i) {
        LUA->PushNumber(double(i + 1));
        LUA->PushUserType_Value(new TraceResultBatch(batches[i], mT), TraceResultBatch::id);
        LUA->SetTable(-3);
    }
    return 1;
}

const Material& AccelStruct::GetMaterial(size_t i) const { return mT->materials[i]; }     // :840-843

} // namespace vistrace
