// Binding.cpp -- Lua thunks of the Tracing API.  Each thunk mirrors the stack handling of its
// namesake in source/VisTrace.cpp (cited per function); the C++ objects they reach
// (AccelStruct, TraceResult) are the host classes in this directory.
#include "Binding.h"

#include <cmath>
#include <cstdint>
#include <cstring>

#include <cstdio>

#include "TraceResult.h"
#include "TraceResultBatch.h"

using namespace GarrysMod::Lua;

namespace vistrace {

int AccelStruct_id = -1;
static World* g_pWorld = nullptr;

void SetWorld(World* world) { g_pWorld = world; }

static inline Vector MakeVector(float x, float y, float z) { return Vector{x, y, z}; }    // Utils.h MakeVector

// ---- AccelStruct -----------------------------------------------------------------------------
LUA_FUNCTION(AccelStruct_gc)                                           // VisTrace.cpp:753-762
{
    LUA->CheckType(1, AccelStruct_id);
    AccelStruct* p = LUA->GetUserType<AccelStruct>(1, AccelStruct_id);
    LUA->SetUserType(1, nullptr);
    delete p;
    return 0;
}

// Both CreateAccel(ents, traceWorld) and accel:Rebuild(ents, traceWorld) accept nothing / nil / a table
// for `ents` and leave exactly [self,] table on the stack for PopulateAccel (VisTrace.cpp:776-788, 806-815).
// `tablePos` is where the table argument sits (1 for CreateAccel, 2 for Rebuild); `onBadType` runs before
// the type error is raised (the error does not unwind C++ frames in the real module).
template <class Cleanup>
static void NormaliseEntityListArg(ILuaBase* LUA, int tablePos, Cleanup onBadType)
{
    if (LUA->Top() < tablePos) {
        LUA->CreateTable();
    } else if (LUA->IsType(tablePos, Type::Nil)) {
        LUA->Pop(LUA->Top() - (tablePos - 1));
        LUA->CreateTable();
    } else {
        if (!LUA->IsType(tablePos, Type::Table)) {
            onBadType();
            LUA->CheckType(tablePos, Type::Table);   // raises the formatted type error
        }
        LUA->Pop(LUA->Top() - tablePos);
    }
}

static bool TraceWorldArg(ILuaBase* LUA, int pos) { return LUA->IsType(pos, Type::Bool) ? LUA->GetBool(pos) : true; }

LUA_FUNCTION(vistrace_CreateAccel)                                     // VisTrace.cpp:770-792
{
    const bool traceWorld = TraceWorldArg(LUA, 2);
    AccelStruct* pAccelStruct = new AccelStruct();
    NormaliseEntityListArg(LUA, 1, [&] { delete pAccelStruct; });
    try {
        pAccelStruct->PopulateAccel(LUA, traceWorld ? g_pWorld : nullptr);
    } catch (...) {                                 // test doubles throw instead of longjmp: do not leak
        delete pAccelStruct;
        throw;
    }
    LUA->PushUserType_Value(pAccelStruct, AccelStruct_id);
    return 1;
}

LUA_FUNCTION(AccelStruct_Rebuild)                                      // VisTrace.cpp:798-818
{
    LUA->CheckType(1, AccelStruct_id);
    const bool traceWorld = TraceWorldArg(LUA, 3);
    AccelStruct* pAccelStruct = LUA->GetUserType<AccelStruct>(1, AccelStruct_id);
    NormaliseEntityListArg(LUA, 2, [] {});
    pAccelStruct->PopulateAccel(LUA, traceWorld ? g_pWorld : nullptr);
    return 0;
}

LUA_FUNCTION(AccelStruct_Traverse)                                     // VisTrace.cpp:831-836
{
    LUA->CheckType(1, AccelStruct_id);
    return LUA->GetUserType<AccelStruct>(1, AccelStruct_id)->Traverse(LUA);
}

LUA_FUNCTION(AccelStruct_TraverseBatch)                                // additive
{
    LUA->CheckType(1, AccelStruct_id);
    return LUA->GetUserType<AccelStruct>(1, AccelStruct_id)->TraverseBatch(LUA);
}

LUA_FUNCTION(AccelStruct_tostring)                                     // VisTrace.cpp:838-842
{
    LUA->PushString("AccelStruct");
    return 1;
}

// ---- TraceResult -----------------------------------------------------------------------------
static TraceResult* Self(ILuaBase* LUA)
{
    LUA->CheckType(1, TraceResult::id);
    return LUA->GetUserType<TraceResult>(1, TraceResult::id);
}

LUA_FUNCTION(TraceResult_gc)                                           // VisTrace.cpp:458-467
{
    TraceResult* p = Self(LUA);
    LUA->SetUserType(1, nullptr);
    delete p;
    return 0;
}

LUA_FUNCTION(TraceResult_Pos)                                          // VisTrace.cpp:469-476
{
    const Vec3& p = Self(LUA)->GetPos();
    LUA->PushVector(MakeVector(p.x, p.y, p.z));
    return 1;
}

LUA_FUNCTION(TraceResult_Incident)                                     // VisTrace.cpp:478-485
{
    TraceResult* r = Self(LUA);
    LUA->PushVector(MakeVector(r->wo.x, r->wo.y, r->wo.z));
    return 1;
}

LUA_FUNCTION(TraceResult_Distance)                                     // VisTrace.cpp:487-493
{
    LUA->PushNumber(Self(LUA)->distance);
    return 1;
}

LUA_FUNCTION(TraceResult_Entity)                                       // VisTrace.cpp:495-513
{
    TraceResult* pResult = Self(LUA);

    LUA->PushSpecial(SPECIAL_GLOB);
    LUA->GetField(-1, "Entity");
    LUA->PushNumber(pResult->entIdx);
    LUA->Call(1, 1);

    // the entity behind that index may have been replaced since the build: hand out NULL instead
    void* pEnt = LUA->GetUserdataRaw(-1, Type::Entity);
    if (pEnt == nullptr || pEnt != pResult->rawEnt) {
        LUA->GetField(-2, "Entity");
        LUA->PushNumber(-1);
        LUA->Call(1, 1);
    }

    return 1;
}

LUA_FUNCTION(TraceResult_GeometricNormal)                              // VisTrace.cpp:515-522
{
    TraceResult* r = Self(LUA);
    LUA->PushVector(MakeVector(r->geometricNormal.x, r->geometricNormal.y, r->geometricNormal.z));
    return 1;
}

LUA_FUNCTION(TraceResult_Normal)                                       // VisTrace.cpp:524-531
{
    TraceResult* r = Self(LUA);
    const Vec3& v = r->GetNormal();
    LUA->PushVector(MakeVector(v.x, v.y, v.z));
    return 1;
}

LUA_FUNCTION(TraceResult_Tangent)                                      // VisTrace.cpp:533-540
{
    TraceResult* r = Self(LUA);
    const Vec3& v = r->GetTangent();
    LUA->PushVector(MakeVector(v.x, v.y, v.z));
    return 1;
}

LUA_FUNCTION(TraceResult_Binormal)                                     // VisTrace.cpp:542-549
{
    TraceResult* r = Self(LUA);
    const Vec3& v = r->GetBinormal();
    LUA->PushVector(MakeVector(v.x, v.y, v.z));
    return 1;
}

LUA_FUNCTION(TraceResult_Barycentric)                                  // VisTrace.cpp:552-559
{
    TraceResult* r = Self(LUA);
    LUA->PushVector(MakeVector(r->uvw.x, r->uvw.y, r->uvw.z));
    return 1;
}

LUA_FUNCTION(TraceResult_TextureUV)                                    // VisTrace.cpp:560-571
{
    TraceResult* r = Self(LUA);
    LUA->CreateTable();
    LUA->PushNumber(r->texUV.x);
    LUA->SetField(-2, "u");
    LUA->PushNumber(r->texUV.y);
    LUA->SetField(-2, "v");
    return 1;
}

LUA_FUNCTION(TraceResult_SubMaterialIndex)                             // VisTrace.cpp:573-580
{
    LUA->PushNumber(Self(LUA)->submatIdx + 1);
    return 1;
}

LUA_FUNCTION(TraceResult_MaterialFlags)                                // VisTrace.cpp:613-619
{
    LUA->PushNumber(static_cast<double>(Self(LUA)->GetMaterialFlags()));
    return 1;
}

LUA_FUNCTION(TraceResult_SurfaceFlags)                                 // VisTrace.cpp:620-626
{
    LUA->PushNumber(static_cast<double>(Self(LUA)->GetSurfFlags()));
    return 1;
}

LUA_FUNCTION(TraceResult_HitSky)                                       // VisTrace.cpp:628-634
{
    LUA->PushBool(Self(LUA)->hitSky);
    return 1;
}

LUA_FUNCTION(TraceResult_HitWater)                                     // VisTrace.cpp:635-641
{
    LUA->PushBool(Self(LUA)->HitWater());
    return 1;
}

LUA_FUNCTION(TraceResult_FrontFacing)                                  // VisTrace.cpp:643-649
{
    LUA->PushBool(Self(LUA)->frontFacing);
    return 1;
}

LUA_FUNCTION(TraceResult_tostring)                                     // VisTrace.cpp:742-746
{
    LUA->PushString("VisTraceResult");
    return 1;
}

// ---- TraceResultBatch: the getters of TraceResult with the ray's 1-based index as argument (TraceResultBatch.h) -------
// A miss yields no value (as Traverse returns nothing for a miss); batch:Hit(i) tells hits from misses.
static TraceResultBatch* BatchSelf(ILuaBase* LUA)
{
    LUA->CheckType(1, TraceResultBatch::id);
    TraceResultBatch* b = LUA->GetUserType<TraceResultBatch>(1, TraceResultBatch::id);
    if (!b) LUA->ThrowError("VisTraceResultBatch has been released");
    return b;
}

static uint64_t BatchIndex(ILuaBase* LUA, TraceResultBatch* b)
{
    const double d = LUA->CheckNumber(2);
    if (!(d >= 1.0 && d <= double(b->Count())) || d != std::floor(d)) LUA->ArgError(2, "ray index out of range");
    return uint64_t(d) - 1;
}

static void BatchFetchError(ILuaBase* LUA)
{
    static thread_local char msg[512];
    std::snprintf(msg, sizeof(msg), "VisTrace: batch results unavailable: %s", vt_last_error());
    LUA->ThrowError(msg);
}

static const vt_hit& BatchHit(ILuaBase* LUA, TraceResultBatch* b, uint64_t i)
{
    const vt_hit* h = b->Hits();
    if (!h) BatchFetchError(LUA);
    return h[i];
}

static const vt_hit_attrs* BatchAttrs(ILuaBase* LUA, TraceResultBatch* b, uint64_t i)   // NULL for a miss
{
    const vt_hit_attrs* a = b->Attrs();
    if (!a) BatchFetchError(LUA);
    return a[i].hit ? &a[i] : nullptr;
}

LUA_FUNCTION(TraceResultBatch_gc)
{
    LUA->CheckType(1, TraceResultBatch::id);
    TraceResultBatch* p = LUA->GetUserType<TraceResultBatch>(1, TraceResultBatch::id);
    LUA->SetUserType(1, nullptr);
    delete p;
    return 0;
}

LUA_FUNCTION(TraceResultBatch_tostring) { LUA->PushString("VisTraceResultBatch"); return 1; }

LUA_FUNCTION(TraceResultBatch_Count) { LUA->PushNumber(double(BatchSelf(LUA)->Count())); return 1; }

LUA_FUNCTION(TraceResultBatch_Hit)
{
    TraceResultBatch* b = BatchSelf(LUA);
    LUA->PushBool(BatchHit(LUA, b, BatchIndex(LUA, b)).prim != VT_MISS);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_Hits)                                    // the packed vt_hit records, for bulk consumers
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit* h = b->Hits();
    if (!h && b->Count()) BatchFetchError(LUA);
    if (b->Count() * sizeof(vt_hit) > 0xFFFFFFFFull)             // PushString takes an unsigned length
        LUA->ThrowError("VisTraceResultBatch:Hits: more than 4 GiB of hit records do not fit one Lua string (read them through the getters)");
    if (b->Count() == 0) { LUA->PushString(""); return 1; }       // an empty batch: there is no array (and length 0 would mean strlen)
    LUA->PushString(reinterpret_cast<const char*>(h), unsigned(b->Count() * sizeof(vt_hit)));
    return 1;
}

LUA_FUNCTION(TraceResultBatch_Get)                                     // the reference's per-ray object, or nothing
{
    TraceResultBatch* b = BatchSelf(LUA);
    const uint64_t i = BatchIndex(LUA, b);
    if (!b->Hits()) BatchFetchError(LUA);
    TraceResult* r = b->MakeResult(i);
    if (!r) return 0;
    LUA->PushUserType_Value(r, TraceResult::id);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_Pos)                                     // TraceResult.cpp:255-262 on the device
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit_attrs* a = BatchAttrs(LUA, b, BatchIndex(LUA, b));
    if (!a) return 0;
    LUA->PushVector(MakeVector(a->pos[0], a->pos[1], a->pos[2]));
    return 1;
}

LUA_FUNCTION(TraceResultBatch_Incident)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit_attrs* a = BatchAttrs(LUA, b, BatchIndex(LUA, b));
    if (!a) return 0;
    LUA->PushVector(MakeVector(a->wo[0], a->wo[1], a->wo[2]));
    return 1;
}

LUA_FUNCTION(TraceResultBatch_Distance)                                // needs the hit records only
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushNumber(h.t);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_GeometricNormal)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit_attrs* a = BatchAttrs(LUA, b, BatchIndex(LUA, b));
    if (!a) return 0;
    LUA->PushVector(MakeVector(a->ngeo[0], a->ngeo[1], a->ngeo[2]));
    return 1;
}

// Normal / Tangent / Binormal of ray i's hit: the frame the device materialised for the whole batch (vt_hit_tbn)
static int PushBatchFrameVector(ILuaBase* LUA, int which)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const uint64_t i = BatchIndex(LUA, b);
    if (BatchHit(LUA, b, i).prim == VT_MISS) return 0;
    const vt_hit_tbn* t = b->Tbn();
    if (!t) BatchFetchError(LUA);
    const float* v = which == 0 ? t[i].normal : which == 1 ? t[i].tangent : t[i].binormal;
    LUA->PushVector(MakeVector(v[0], v[1], v[2]));
    return 1;
}
LUA_FUNCTION(TraceResultBatch_Normal) { return PushBatchFrameVector(LUA, 0); }
LUA_FUNCTION(TraceResultBatch_Tangent) { return PushBatchFrameVector(LUA, 1); }
LUA_FUNCTION(TraceResultBatch_Binormal) { return PushBatchFrameVector(LUA, 2); }

LUA_FUNCTION(TraceResultBatch_Barycentric)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushVector(MakeVector(h.u, h.v, 1.f - h.u - h.v));            // TraceResult.cpp:70
    return 1;
}

LUA_FUNCTION(TraceResultBatch_FrontFacing)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit_attrs* a = BatchAttrs(LUA, b, BatchIndex(LUA, b));
    if (!a) return 0;
    LUA->PushBool(a->front != 0);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_TextureUV)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const uint64_t i = BatchIndex(LUA, b);
    if (BatchHit(LUA, b, i).prim == VT_MISS) return 0;
    const vt_hit_shade* s = b->Shade();
    if (!s) BatchFetchError(LUA);
    LUA->CreateTable();
    LUA->PushNumber(s[i].tex_uv[0]);
    LUA->SetField(-2, "u");
    LUA->PushNumber(s[i].tex_uv[1]);
    LUA->SetField(-2, "v");
    return 1;
}

LUA_FUNCTION(TraceResultBatch_SubMaterialIndex)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushNumber(double(b->TriangleOf(h).material) + 1);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_MaterialFlags)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushNumber(double(b->MaterialOf(h).flags));
    return 1;
}

LUA_FUNCTION(TraceResultBatch_SurfaceFlags)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushNumber(double(b->MaterialOf(h).surfFlags));
    return 1;
}

LUA_FUNCTION(TraceResultBatch_HitSky)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushBool((b->MaterialOf(h).surfFlags & SURF_SKY) != SURF_NONE);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_HitWater)
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    LUA->PushBool(b->MaterialOf(h).water);
    return 1;
}

LUA_FUNCTION(TraceResultBatch_Entity)                                  // as TraceResult:Entity, VisTrace.cpp:495-513
{
    TraceResultBatch* b = BatchSelf(LUA);
    const vt_hit& h = BatchHit(LUA, b, BatchIndex(LUA, b));
    if (h.prim == VT_MISS) return 0;
    const Entity& ent = b->EntityOf(h);

    LUA->PushSpecial(SPECIAL_GLOB);
    LUA->GetField(-1, "Entity");
    LUA->PushNumber(ent.id);
    LUA->Call(1, 1);

    void* pEnt = LUA->GetUserdataRaw(-1, Type::Entity);
    if (pEnt == nullptr || pEnt != ent.rawEntity) {
        LUA->GetField(-2, "Entity");
        LUA->PushNumber(-1);
        LUA->Call(1, 1);
    }
    return 1;
}

// ---- helpers ------------------------------------------------------------------------------------
LUA_FUNCTION(vistrace_CalcRayOrigin)                                   // VisTrace.cpp:1478-1519
{
    LUA->CheckType(1, Type::Vector);
    LUA->CheckType(2, Type::Vector);
    const Vector p = LUA->GetVector(1), nrm = LUA->GetVector(2);
    const float pos[3] = {p.x, p.y, p.z}, normal[3] = {nrm.x, nrm.y, nrm.z};

    const float origin = 1.f / 32.f;
    const float fScale = 1.f / 65536.f;
    const float iScale = 256.f;

    float out[3];
    for (int k = 0; k < 3; ++k) {
        // per-component integer offset to the bit representation of the fp32 position (:1500-1508)
        const int32_t iOff = static_cast<int32_t>(normal[k] * iScale);
        int32_t bits;
        std::memcpy(&bits, &pos[k], 4);
        bits += pos[k] < 0.f ? -iOff : iOff;
        float iPos;
        std::memcpy(&iPos, &bits, 4);
        // small fixed offset near the origin, the variable one elsewhere (:1511-1516)
        out[k] = std::fabs(pos[k]) < origin ? pos[k] + normal[k] * fScale : iPos;
    }
    LUA->PushVector(MakeVector(out[0], out[1], out[2]));
    return 1;
}

// ---- registration (VisTrace.cpp:1685-1752, 1817-1832) --------------------------------------------
static void Method(ILuaBase* LUA, const char* name, CFunc f)
{
    LUA->PushCFunction(f);
    LUA->SetField(-2, name);
}

void RegisterTracingApi(ILuaBase* LUA)
{
    TraceResult::id = LUA->CreateMetaTable("VisTraceResult");          // :1685-1740 (the getters on this path)
    LUA->Push(-1);
    LUA->SetField(-2, "__index");
    Method(LUA, "__tostring", TraceResult_tostring);
    Method(LUA, "__gc", TraceResult_gc);
    Method(LUA, "Pos", TraceResult_Pos);
    Method(LUA, "Incident", TraceResult_Incident);
    Method(LUA, "Distance", TraceResult_Distance);
    Method(LUA, "Entity", TraceResult_Entity);
    Method(LUA, "GeometricNormal", TraceResult_GeometricNormal);
    Method(LUA, "Normal", TraceResult_Normal);
    Method(LUA, "Tangent", TraceResult_Tangent);
    Method(LUA, "Binormal", TraceResult_Binormal);
    Method(LUA, "Barycentric", TraceResult_Barycentric);
    Method(LUA, "TextureUV", TraceResult_TextureUV);
    Method(LUA, "SubMaterialIndex", TraceResult_SubMaterialIndex);
    Method(LUA, "MaterialFlags", TraceResult_MaterialFlags);
    Method(LUA, "SurfaceFlags", TraceResult_SurfaceFlags);
    Method(LUA, "HitSky", TraceResult_HitSky);
    Method(LUA, "HitWater", TraceResult_HitWater);
    Method(LUA, "FrontFacing", TraceResult_FrontFacing);
    LUA->Pop();

    TraceResultBatch::id = LUA->CreateMetaTable("VisTraceResultBatch");   // additive: what TraverseBatch(buffer) returns
    LUA->Push(-1);
    LUA->SetField(-2, "__index");
    Method(LUA, "__tostring", TraceResultBatch_tostring);
    Method(LUA, "__gc", TraceResultBatch_gc);
    Method(LUA, "Count", TraceResultBatch_Count);
    Method(LUA, "Hit", TraceResultBatch_Hit);
    Method(LUA, "Hits", TraceResultBatch_Hits);
    Method(LUA, "Get", TraceResultBatch_Get);
    Method(LUA, "Pos", TraceResultBatch_Pos);
    Method(LUA, "Incident", TraceResultBatch_Incident);
    Method(LUA, "Distance", TraceResultBatch_Distance);
    Method(LUA, "Entity", TraceResultBatch_Entity);
    Method(LUA, "GeometricNormal", TraceResultBatch_GeometricNormal);
    Method(LUA, "Normal", TraceResultBatch_Normal);
    Method(LUA, "Tangent", TraceResultBatch_Tangent);
    Method(LUA, "Binormal", TraceResultBatch_Binormal);
    Method(LUA, "Barycentric", TraceResultBatch_Barycentric);
    Method(LUA, "TextureUV", TraceResultBatch_TextureUV);
    Method(LUA, "SubMaterialIndex", TraceResultBatch_SubMaterialIndex);
    Method(LUA, "MaterialFlags", TraceResultBatch_MaterialFlags);
    Method(LUA, "SurfaceFlags", TraceResultBatch_SurfaceFlags);
    Method(LUA, "HitSky", TraceResultBatch_HitSky);
    Method(LUA, "HitWater", TraceResultBatch_HitWater);
    Method(LUA, "FrontFacing", TraceResultBatch_FrontFacing);
    LUA->Pop();

    AccelStruct_id = LUA->CreateMetaTable("AccelStruct");              // :1742-1752
    LUA->Push(-1);
    LUA->SetField(-2, "__index");
    Method(LUA, "__tostring", AccelStruct_tostring);
    Method(LUA, "__gc", AccelStruct_gc);
    Method(LUA, "Traverse", AccelStruct_Traverse);
    Method(LUA, "Rebuild", AccelStruct_Rebuild);
    Method(LUA, "TraverseBatch", AccelStruct_TraverseBatch);            // additive
    LUA->Pop();

    LUA->PushSpecial(SPECIAL_GLOB);                                    // :1817-1832 (the entries on this path)
    LUA->CreateTable();
    Method(LUA, "CreateAccel", vistrace_CreateAccel);
    Method(LUA, "CalcRayOrigin", vistrace_CalcRayOrigin);
    LUA->SetField(-2, "vistrace");
    LUA->Pop();
}

} // namespace vistrace
