// Binding.h -- registration of the Tracing API on a Lua state (subset of GMOD_MODULE_OPEN,
// source/VisTrace.cpp:1685-1752, 1817-1832).
#pragma once

#include "AccelStruct.h"
#include "LuaShim.h"

namespace vistrace {

extern int AccelStruct_id;
void SetWorld(World* world);          // g_pWorld of source/VisTrace.cpp:751

// thunks (same names and stack conventions as the reference's LUA_FUNCTIONs)
LUA_FUNCTION(vistrace_CreateAccel);
LUA_FUNCTION(vistrace_CalcRayOrigin);
LUA_FUNCTION(AccelStruct_Rebuild);
LUA_FUNCTION(AccelStruct_Traverse);
LUA_FUNCTION(AccelStruct_TraverseBatch);
LUA_FUNCTION(AccelStruct_gc);
LUA_FUNCTION(AccelStruct_tostring);
LUA_FUNCTION(TraceResult_gc);
LUA_FUNCTION(TraceResult_Pos);
LUA_FUNCTION(TraceResult_Incident);
LUA_FUNCTION(TraceResult_Distance);
LUA_FUNCTION(TraceResult_Entity);
LUA_FUNCTION(TraceResult_GeometricNormal);
LUA_FUNCTION(TraceResult_Barycentric);
LUA_FUNCTION(TraceResult_TextureUV);
LUA_FUNCTION(TraceResult_SubMaterialIndex);
LUA_FUNCTION(TraceResult_MaterialFlags);
LUA_FUNCTION(TraceResult_SurfaceFlags);
LUA_FUNCTION(TraceResult_HitSky);
LUA_FUNCTION(TraceResult_HitWater);
LUA_FUNCTION(TraceResult_FrontFacing);
LUA_FUNCTION(TraceResult_tostring);

// creates the two metatables (type ids) and the global `vistrace` table the way GMOD_MODULE_OPEN does
void RegisterTracingApi(GarrysMod::Lua::ILuaBase* LUA);

} // namespace vistrace
