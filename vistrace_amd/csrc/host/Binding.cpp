// Binding.cpp -- Lua thunks of the Tracing API.  Each thunk mirrors the stack handling of its
// namesake in source/VisTrace.cpp (cited per function); the C++ objects they reach
// (AccelStruct, TraceResult) are the host classes in this directory.
#include "Binding.h"

#include "TraceResult.h"

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

// The reference's TraceResult:Entity() (VisTrace.cpp:495-513) calls the engine's global
// Entity(); the engine-independent part is the index it passes, exposed here.
LUA_FUNCTION(TraceResult_EntIndex)
{
    LUA->PushNumber(Self(LUA)->entIdx);
    return 1;
}

LUA_FUNCTION(TraceResult_GeometricNormal)                              // VisTrace.cpp:515-522
{
    TraceResult* r = Self(LUA);
    LUA->PushVector(MakeVector(r->geometricNormal.x, r->geometricNormal.y, r->geometricNormal.z));
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

LUA_FUNCTION(TraceResult_FrontFacing)
{
    LUA->PushBool(Self(LUA)->frontFacing);
    return 1;
}

LUA_FUNCTION(TraceResult_tostring)
{
    LUA->PushString("TraceResult");
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
    TraceResult::id = LUA->CreateMetaTable("VisTraceResult");
    LUA->Push(-1);
    LUA->SetField(-2, "__index");
    Method(LUA, "__gc", TraceResult_gc);
    Method(LUA, "__tostring", TraceResult_tostring);
    Method(LUA, "Pos", TraceResult_Pos);
    Method(LUA, "Incident", TraceResult_Incident);
    Method(LUA, "Distance", TraceResult_Distance);
    Method(LUA, "EntIndex", TraceResult_EntIndex);
    Method(LUA, "GeometricNormal", TraceResult_GeometricNormal);
    Method(LUA, "Barycentric", TraceResult_Barycentric);
    Method(LUA, "TextureUV", TraceResult_TextureUV);
    Method(LUA, "SubMaterialIndex", TraceResult_SubMaterialIndex);
    Method(LUA, "FrontFacing", TraceResult_FrontFacing);
    LUA->Pop();

    AccelStruct_id = LUA->CreateMetaTable("AccelStruct");
    LUA->Push(-1);
    LUA->SetField(-2, "__index");
    Method(LUA, "__gc", AccelStruct_gc);
    Method(LUA, "__tostring", AccelStruct_tostring);
    Method(LUA, "Rebuild", AccelStruct_Rebuild);
    Method(LUA, "Traverse", AccelStruct_Traverse);
    Method(LUA, "TraverseBatch", AccelStruct_TraverseBatch);
    LUA->Pop();
}

} // namespace vistrace
