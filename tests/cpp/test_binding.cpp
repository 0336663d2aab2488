// test_binding.cpp -- drives the Tracing API thunks (vistrace.CreateAccel, AccelStruct:Rebuild,
// :Traverse, :TraverseBatch, TraceResult getters) through a fake Lua state and checks argument
// defaults, validation messages (source/objects/AccelStruct.cpp:780-806), return counts and
// the hit-record getters.  `--cpu` runs the part that needs no GPU.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <chrono>
#include <malloc.h>
#include <cstring>
#include <functional>
#include <string>

#include "AccelStruct.h"
#include "Binding.h"
#include "FakeLua.h"
#include "TraceResult.h"
#include "TraceResultBatch.h"

using namespace vistrace;
using fakelua::LuaError;
using fakelua::State;
namespace LT = GarrysMod::Lua::Type;

static int g_fail = 0, g_run = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        ++g_run;                                                                         \
        if (!(cond)) { ++g_fail; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static std::string error_of(const std::function<void()>& f, int* arg = nullptr)
{
    try { f(); } catch (const LuaError& e) { if (arg) *arg = e.arg; return e.what(); }
    return "";
}
static bool contains(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

// floor triangle pair at z = 0 (one-sided world brush), entity quad at z = 5 (two-sided)
static Triangle make_tri(Vec3 a, Vec3 b, Vec3 c, bool oneSided, size_t material)
{
    Triangle t;
    t.p0 = a; t.p1 = b; t.p2 = c;
    t.oneSided = oneSided;
    t.material = material;
    t.uvs[0] = Vec2{0, 0}; t.uvs[1] = Vec2{1, 0}; t.uvs[2] = Vec2{0, 1};
    t.alphas[0] = 0.f; t.alphas[1] = 1.f; t.alphas[2] = 0.5f;
    // a bent vertex frame (unit vectors, tangent perpendicular to normal at every vertex)
    t.normals[0] = Vec3{0, 0, 1};  t.normals[1] = Vec3{0.6f, 0, 0.8f};   t.normals[2] = Vec3{0, 0.6f, 0.8f};
    t.tangents[0] = Vec3{1, 0, 0}; t.tangents[1] = Vec3{0.8f, 0, -0.6f}; t.tangents[2] = Vec3{1, 0, 0};
    return t;
}

struct FakeMeshSource : IEntityMeshSource {
    int calls = 0;
    bool AppendEntity(void* ud, Entity& ent, std::vector<Triangle>& tris, std::vector<Material>& mats) override
    {
        ++calls;
        ent.rawEntity = ud;
        ent.id = 42;
        mats.push_back(Material{"models/fake", MATFLAG_NONE});
        tris.push_back(make_tri({0, 0, 5}, {4, 0, 5}, {0, 4, 5}, false, 0));
        return true;
    }
};

static void test_cpu_side()
{
    State L;
    RegisterTracingApi(&L);
    CHECK(AccelStruct_id >= LT::Count && TraceResult::id >= LT::Count && AccelStruct_id != TraceResult::id);
    CHECK(L.Top() == 0);
    // the registered surface, by name: the reference's Tracing API (VisTrace.cpp:1693-1720, 1745-1751, 1820) + TraverseBatch
    for (const char* m : {"__gc", "__tostring", "Pos", "Incident", "Distance", "Entity", "GeometricNormal", "Normal", "Tangent", "Binormal", "Barycentric", "TextureUV",
                          "SubMaterialIndex", "MaterialFlags", "SurfaceFlags", "HitSky", "HitWater", "FrontFacing"})
        CHECK(L.find_method(TraceResult::id, m) != nullptr);
    for (const char* m : {"__gc", "__tostring", "Traverse", "Rebuild", "TraverseBatch"})
        CHECK(L.find_method(AccelStruct_id, m) != nullptr);
    CHECK(L.find_method(TraceResult::id, "EntIndex") == nullptr);       // not a reference name
    CHECK(L.find_global("vistrace", "CreateAccel") != nullptr && L.find_global("vistrace", "CreateAccel")->fn == vistrace_CreateAccel);
    CHECK(L.find_global("vistrace", "CalcRayOrigin") != nullptr);
    {   // vistrace.CalcRayOrigin (VisTrace.cpp:1478-1519): far from the origin the offset is an integer step on the bit pattern
        L.PushValue(State::Vec(100.f, -100.f, 0.01f));
        L.PushValue(State::Vec(0.f, 1.f, 1.f));
        CHECK(L.find_global("vistrace", "CalcRayOrigin")->fn(&L) == 1);
        ::Vector o = L.GetVector(-1);
        float y = -100.f; int32_t bits; std::memcpy(&bits, &y, 4); bits += -256; float ey; std::memcpy(&ey, &bits, 4);
        CHECK(o.x == 100.f && o.y == ey && o.z == 0.01f + 1.f / 65536.f);
        L.Pop(L.Top());
    }
    {
        fakelua::Value tr = State::User(nullptr, TraceResult::id);
        CHECK(L.find_method(TraceResult::id, "__tostring")->fn(&L) == 1 && L.stack.back().str == "VisTraceResult");
        L.Pop(L.Top());
    }

    // Traverse on an accel that was never built -> the reference's message (AccelStruct.cpp:780)
    {
        AccelStruct* a = new AccelStruct();
        L.PushUserType(a, AccelStruct_id);
        L.PushValue(State::Vec(0, 0, 1));
        L.PushValue(State::Vec(0, 0, -1));
        std::string e = error_of([&] { AccelStruct_Traverse(&L); });
        CHECK(e == "Unable to perform traversal, acceleration structure invalid (use AccelStruct:Rebuild to rebuild it)");
        L.Pop(L.Top());
        L.PushUserType(a, AccelStruct_id);
        CHECK(AccelStruct_gc(&L) == 0);
        L.Pop(L.Top());
    }
    // wrong self type
    {
        L.PushNumber(3);
        std::string e = error_of([&] { AccelStruct_Traverse(&L); });
        CHECK(contains(e, "AccelStruct expected"));
        L.Pop(L.Top());
    }
    // CreateAccel: first argument must be a table (or nil / nothing)
    {
        L.PushNumber(1);
        std::string e = error_of([&] { vistrace_CreateAccel(&L); });
        CHECK(contains(e, "table expected"));
        L.Pop(L.Top());
    }
    // CreateAccel: "Build list must only contain entities" (AccelStruct.cpp:575), before any device work
    {
        L.PushValue(State::Array({State::Num(7)}));
        L.PushBool(false);
        std::string e = error_of([&] { vistrace_CreateAccel(&L); });
        CHECK(e == "Build list must only contain entities");
        L.Pop(L.Top());
    }
    {
        CHECK(AccelStruct_tostring(&L) == 1);
        CHECK(L.stack.back().str == "AccelStruct");
        L.Pop(L.Top());
    }
    // TraceResult hit-record math (TraceResult.cpp:45-86, 255-262) on a hand-checked case
    {
        Triangle t = make_tri({0, 0, 0}, {1, 0, 0}, {0, 1, 0}, false, 3);
        Entity ent; ent.id = 9;
        Material mat{"brush/sky", MATFLAG_NOCULL};
        mat.surfFlags = SURF_SKY | 0x80u;
        mat.water = true;
        TraceResult r(Vec3{0, 0, -2}, 1.0f, -1, -1, t, 5, Vec2{0.25f, 0.5f}, ent, mat);
        CHECK(r.hitSky && r.HitWater() && r.GetMaterialFlags() == MATFLAG_NOCULL && r.GetSurfFlags() == (SURF_SKY | 0x80u));
        CHECK(r.uvw.x == 0.25f && r.uvw.y == 0.5f && r.uvw.z == 0.25f);
        CHECK(r.GetPos().x == 0.25f && r.GetPos().y == 0.5f && r.GetPos().z == 0.0f);
        CHECK(r.geometricNormal.x == 0 && r.geometricNormal.y == 0 && r.geometricNormal.z == -1);   // cross(e1,e2)/|n|
        CHECK(r.wo.z == 1.0f && r.frontFacing == false);
        CHECK(r.texUV.x == 0.25f && r.texUV.y == 0.5f);           // w*uv0 + u*uv1 + v*uv2
        CHECK(r.blendFactor == 0.25f * 0.f + 0.25f * 1.f + 0.5f * 0.5f);
        CHECK(r.entIdx == 9 && r.submatIdx == 3 && r.distance == 1.0f && r.primitiveIndex == 5);
    }
}

// obj:name(args...) the way a script calls it: the function is looked up BY NAME through the metatable that
// RegisterTracingApi created for the object's type; the stack holds self + args on entry.
static int call_method(State& L, const fakelua::Value& self, const char* name, std::vector<fakelua::Value> args = {})
{
    const fakelua::Value* f = L.find_method(self.type, name);
    if (!f || f->type != LT::Function || !f->fn) { ++g_fail; std::printf("FAIL: no method %s registered\n", name); return -1; }
    L.Pop(L.Top());
    L.PushValue(self);
    for (auto& v : args) L.PushValue(v);
    return f->fn(&L);
}

static int call_traverse(State& L, AccelStruct* a, std::vector<fakelua::Value> args)
{
    return call_method(L, State::User(a, AccelStruct_id), "Traverse", std::move(args));
}

// the engine's global Entity(index) as the fake game provides it: index -> entity userdata, anything else -> NULL entity
static void* g_entityByIndex[64] = {};
static int Fake_Entity(GarrysMod::Lua::ILuaBase* LUA)
{
    const int idx = int(LUA->GetNumber(1));
    void* p = (idx >= 0 && idx < 64) ? g_entityByIndex[idx] : nullptr;
    LUA->PushUserType(p, LT::Entity);
    return 1;
}
static void install_entity_global(State& L)
{
    L.PushSpecial(GarrysMod::Lua::SPECIAL_GLOB);
    L.PushCFunction(Fake_Entity);
    L.SetField(-2, "Entity");
    L.Pop();
}

static void test_gpu_side()
{
    State L;
    RegisterTracingApi(&L);
    install_entity_global(L);
    FakeMeshSource src;
    AccelStruct::SetEntityMeshSource(&src);

    World world;
    world.materials.push_back(Material{"brush/floor", MATFLAG_NONE});
    world.materials.push_back(Material{"brush/nocull", MATFLAG_NOCULL});
    world.materials[1].surfFlags = SURF_SKY;
    world.materials[1].water = true;
    world.entities.push_back(Entity{nullptr, 0});
    // one-sided floor: n = cross(e1,e2) with e1=p0-p1, e2=p2-p0 -> (0,0,-1); rays going -z have nDotDir > 0 => culled
    world.triangles.push_back(make_tri({0, 0, 0}, {10, 0, 0}, {0, 10, 0}, true, 0));
    // same winding, nocull material, shifted in x: never culled
    world.triangles.push_back(make_tri({20, 0, 0}, {30, 0, 0}, {20, 10, 0}, true, 1));
    SetWorld(&world);

    // vistrace.CreateAccel({ent}, true)
    int dummyEntity = 0;
    g_entityByIndex[42] = &dummyEntity;                    // Entity(42) is the entity the accel was built from
    L.PushValue(State::Array({State::User(&dummyEntity, LT::Entity)}));
    L.PushBool(true);
    CHECK(L.find_global("vistrace", "CreateAccel")->fn(&L) == 1);       // vistrace.CreateAccel, resolved by name
    CHECK(L.Top() >= 1 && L.GetType(-1) == AccelStruct_id);
    AccelStruct* accel = L.GetUserType<AccelStruct>(-1, AccelStruct_id);
    const fakelua::Value accelValue = L.stack.back();
    CHECK(accel && accel->IsBuilt() && accel->TriangleCount() == 3 && src.calls == 1);
    CHECK(accel->GetMaterial(1).flags == MATFLAG_NOCULL);

    // defaults: tMin 0, tMax FLT_MAX, cones -1: hit the entity quad from above
    CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1)}) == 1);
    CHECK(L.Top() == 1 && L.GetType(1) == TraceResult::id);
    {
        TraceResult* r = L.GetUserType<TraceResult>(1, TraceResult::id);
        CHECK(r->distance == 3.0f && r->entIdx == 42 && r->primitiveIndex == 2);
        const fakelua::Value res = L.stack[0];
        CHECK(call_method(L, res, "Pos") == 1);
        ::Vector p = L.GetVector(-1);
        CHECK(p.x == 1.f && p.y == 1.f && p.z == 5.f);
        CHECK(call_method(L, res, "Distance") == 1 && L.GetNumber(-1) == 3.0);
        CHECK(call_method(L, res, "Barycentric") == 1);
        ::Vector b = L.GetVector(-1);
        CHECK(b.x == 0.25f && b.y == 0.25f && b.z == 0.5f);
        // TraceResult:Entity() goes through the game's global Entity(entIdx) (VisTrace.cpp:495-513)
        CHECK(call_method(L, res, "Entity") == 1 && L.GetType(-1) == LT::Entity && L.GetUserdataRaw(-1, LT::Entity) == &dummyEntity);
        int other = 0;
        g_entityByIndex[42] = &other;                          // the index now names another entity -> NULL entity (Entity(-1))
        CHECK(call_method(L, res, "Entity") == 1 && L.GetType(-1) == LT::Entity && L.GetUserdataRaw(-1, LT::Entity) == nullptr);
        g_entityByIndex[42] = &dummyEntity;
        CHECK(call_method(L, res, "GeometricNormal") == 1 && L.GetVector(-1).z == -1.f);
        CHECK(call_method(L, res, "FrontFacing") == 1 && L.GetBool(-1) == false);
        {   // TraceResult:Normal / Tangent / Binormal (VisTrace.cpp:524-549; CalcTBN, TraceResult.cpp:132-186, no normal map):
            // uvw = (.25, .25, .5): normal = normalize(.5 n0 + .25 n1 + .25 n2) = normalize(.15, .15, .9); head-on ray: no correction
            const float l = std::sqrt(0.15f * 0.15f + 0.15f * 0.15f + 0.9f * 0.9f);
            auto near = [](float a, float b) { return std::fabs(a - b) < 2e-6f; };
            CHECK(call_method(L, res, "Normal") == 1);
            ::Vector n = L.GetVector(-1);
            CHECK(near(n.x, 0.15f / l) && near(n.y, 0.15f / l) && near(n.z, 0.9f / l));
            CHECK(call_method(L, res, "Tangent") == 1);
            ::Vector tg = L.GetVector(-1);
            const float tl = std::sqrt(0.95f * 0.95f + 0.15f * 0.15f);
            CHECK(near(tg.x, 0.95f / tl) && tg.y == 0.f && near(tg.z, -0.15f / tl));
            CHECK(call_method(L, res, "Binormal") == 1);
            ::Vector bn = L.GetVector(-1);          // vB = cross(vT, vN) per vertex: (0,-1,0), (0,-1,0), (0,-.8,.6) -> normalize(0, -.95, .15)
            CHECK(bn.x == 0.f && near(bn.y, -0.95f / tl) && near(bn.z, 0.15f / tl));
            Vec2 lod;
            CHECK(r->GetTextureLodInfo(lod) == false);           // cone off by default (coneWidth = coneAngle = -1)
        }
        CHECK(call_method(L, res, "SubMaterialIndex") == 1 && L.GetNumber(-1) == 3);   // 2 world mats + 0, 1-based
        CHECK(call_method(L, res, "MaterialFlags") == 1 && L.GetNumber(-1) == double(MATFLAG_NONE));
        CHECK(call_method(L, res, "SurfaceFlags") == 1 && L.GetNumber(-1) == 0.0);
        CHECK(call_method(L, res, "HitSky") == 1 && L.GetBool(-1) == false);
        CHECK(call_method(L, res, "HitWater") == 1 && L.GetBool(-1) == false);
        CHECK(call_method(L, res, "Incident") == 1 && L.GetVector(-1).z == 1.f);
        CHECK(call_method(L, res, "TextureUV") == 1 && L.GetType(-1) == LT::Table);
        CHECK(call_method(L, res, "__tostring") == 1 && L.stack.back().str == "VisTraceResult");
        CHECK(call_method(L, res, "__gc") == 0);
        CHECK(L.GetUserType<TraceResult>(1, TraceResult::id) == nullptr);
    }
    // un-normalised direction: t scales, Incident is normalised
    CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -2)}) == 1);
    { TraceResult* r = L.GetUserType<TraceResult>(1, TraceResult::id); CHECK(r->distance == 1.5f && r->wo.z == 1.f); delete r; }
    // miss -> 0 results (Lua sees nil), stack cleared
    CHECK(call_traverse(L, accel, {State::Vec(100, 100, 8), State::Vec(0, 0, 1)}) == 0);
    CHECK(L.Top() == 0);
    // one-sided floor seen from above (nDotDir > 0) is culled -> the ray passes the floor and misses;
    // from below it hits; the nocull-material floor is hit from both sides
    CHECK(call_traverse(L, accel, {State::Vec(3, 3, 3), State::Vec(0, 0, -1)}) == 0);
    CHECK(call_traverse(L, accel, {State::Vec(3, 3, -3), State::Vec(0, 0, 1)}) == 1);
    delete L.GetUserType<TraceResult>(1, TraceResult::id);
    CHECK(call_traverse(L, accel, {State::Vec(22, 2, 3), State::Vec(0, 0, -1)}) == 1);
    {   // the nocull floor's material carries SURF_SKY and water (VisTrace.cpp:613-641)
        const fakelua::Value res = L.stack[0];
        CHECK(call_method(L, res, "MaterialFlags") == 1 && L.GetNumber(-1) == double(MATFLAG_NOCULL));
        CHECK(call_method(L, res, "SurfaceFlags") == 1 && L.GetNumber(-1) == double(SURF_SKY));
        CHECK(call_method(L, res, "HitSky") == 1 && L.GetBool(-1) == true);
        CHECK(call_method(L, res, "HitWater") == 1 && L.GetBool(-1) == true);
        CHECK(call_method(L, res, "Entity") == 1 && L.GetUserdataRaw(-1, LT::Entity) == nullptr);   // world: Entity(0) is not rawEnt... NULL
        CHECK(call_method(L, res, "__gc") == 0);
    }
    // tMin / tMax window and nil placeholders
    CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1), State::Num(0), State::Num(2.5)}) == 0);
    CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1), State::Nil(), State::Num(3.0)}) == 1);
    delete L.GetUserType<TraceResult>(1, TraceResult::id);
    CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1), State::Num(3.5), State::Nil()}) == 0);

    {   // a valid cone (coneWidth 0.5, coneAngle 0.25) reaches CalcFootprint (TraceResult.cpp:89-103): at distance 3 the cone is
        // 0.25 * 3 + 0.5 = 1.25 wide, dot(wo, ngeo) = -1; tri.lod = 0.5 * log2(uv area 1 / |n| 16) = -2 (Primitives.h:97-103)
        CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1), State::Nil(), State::Nil(), State::Num(0.5), State::Num(0.25)}) == 1);
        TraceResult* r = L.GetUserType<TraceResult>(1, TraceResult::id);
        Vec2 lod;
        CHECK(r->GetTextureLodInfo(lod) && lod.x == -2.f && lod.y == 1.5625f);
        CHECK(r->GetTextureLodInfo(lod) && lod.y == 1.5625f);     // computed once (textureLodSet)
        delete r;
    }

    // validation messages and argument numbers (AccelStruct.cpp:802-806)
    int arg = 0;
    std::string e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Num(-1)}); }, &arg);
    CHECK(contains(e, "tMin cannot be less than 0") && arg == 4);
    e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Num(2), State::Num(2)}); }, &arg);
    CHECK(contains(e, "tMax must be greater than tMin") && arg == 5);
    e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Nil(), State::Nil(), State::Num(1), State::Num(0)}); });
    CHECK(e == "Valid cone width but invalid cone angle passed");
    e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Nil(), State::Nil(), State::Num(-1), State::Num(0.1)}); });
    CHECK(e == "Valid cone angle but invalid cone width passed");
    // ... and the branch points of the two cone checks (:802-803): a width of exactly 0 is a VALID width (scripts/mutants_binding.sh 92 / 96)
    e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Nil(), State::Nil(), State::Num(0), State::Num(0)}); });
    CHECK(e == "Valid cone width but invalid cone angle passed");
    e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Nil(), State::Nil(), State::Num(0), State::Num(-1)}); });
    CHECK(e == "Valid cone width but invalid cone angle passed");
    CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1), State::Nil(), State::Nil(), State::Num(0), State::Num(0.25)}) == 1);
    delete L.GetUserType<TraceResult>(1, TraceResult::id);
    e = error_of([&] { call_traverse(L, accel, {State::Num(1), State::Vec(0, 0, -1)}); }, &arg);
    CHECK(contains(e, "Vector expected") && arg == 2);
    e = error_of([&] { call_traverse(L, accel, {State::Vec(0, 0, 1), State::Vec(0, 0, -1), State::Str("x")}); }, &arg);
    CHECK(contains(e, "number expected") && arg == 4);

    // TraverseBatch: a small batch (host walk) and a batch above the device crossover (one GPU launch) agree with
    // Traverse ray by ray; fields are read by index, so a nil tMin keeps its default and tMax stays field 4
    {
        CHECK(call_method(L, accelValue, "TraverseBatch", {State::Array({
            State::Array({State::Vec(1, 1, 8), State::Vec(0, 0, -1)}),
            State::Array({State::Vec(100, 100, 8), State::Vec(0, 0, 1)}),
            State::Array({State::Vec(3, 3, -3), State::Vec(0, 0, 1), State::Num(0), State::Num(10)}),
            State::Array({State::Vec(3, 3, -3), State::Vec(0, 0, 1), State::Num(0), State::Num(2)}),
        })}) == 1);
        CHECK(L.Top() == 1 && L.GetType(1) == LT::Table);
        auto& kv = L.stack.back().tab->kv;
        CHECK(kv.size() == 4);
        CHECK(kv[0].second.type == TraceResult::id && kv[1].second.type == LT::Bool && kv[2].second.type == TraceResult::id &&
              kv[3].second.type == LT::Bool);
        TraceResult* r0 = static_cast<TraceResult*>(*kv[0].second.ud);
        TraceResult* r2 = static_cast<TraceResult*>(*kv[2].second.ud);
        CHECK(r0->distance == 3.0f && r0->entIdx == 42 && r2->distance == 3.0f && r2->primitiveIndex == 0);
        delete r0; delete r2;
        L.Pop(L.Top());

        // {o, d, nil, tMax}: a table with a hole at 3 (lua_next would skip it)
        fakelua::Value holed = State::NewTable();
        holed.tab->kv.push_back({State::Num(1), State::Vec(1, 1, 8)});
        holed.tab->kv.push_back({State::Num(2), State::Vec(0, 0, -1)});
        holed.tab->kv.push_back({State::Num(4), State::Num(2.5)});              // tMax 2.5 < t = 3 -> miss
        CHECK(call_method(L, accelValue, "TraverseBatch", {State::Array({holed})}) == 1);
        CHECK(L.stack.back().tab->kv.size() == 1 && L.stack.back().tab->kv[0].second.type == LT::Bool);
        std::string be = error_of([&] { call_method(L, accelValue, "TraverseBatch", {State::Array({State::Array({State::Vec(0, 0, 1)})})}); });
        CHECK(be == "Each ray must be a table {origin, direction[, tMin[, tMax]]}");
        be = error_of([&] { call_method(L, accelValue, "TraverseBatch", {State::Array({State::Array({State::Vec(0, 0, 1), State::Vec(0, 0, 1), State::Num(3), State::Num(2)})})}); });
        CHECK(be == "tMax must be greater than tMin");
        be = error_of([&] {                                   // tMax == tMin is refused too (:806 is <=)
            call_method(L, accelValue, "TraverseBatch", {State::Array({State::Array({State::Vec(1, 1, 8), State::Vec(0, 0, -1), State::Num(2), State::Num(2)})})});
        });
        CHECK(be == "tMax must be greater than tMin");

        // 400 rays over the three surfaces: above AccelStruct::kDeviceBatchMin -> the device; each must equal the
        // single-ray Traverse (host walk) bit for bit
        std::vector<fakelua::Value> many;
        std::vector<std::pair<::Vector, ::Vector>> od;
        for (int i = 0; i < 400; ++i) {
            const float x = float((i * 37) % 320) * 0.1f - 1.f, y = float((i * 53) % 120) * 0.1f - 1.f;
            const bool up = (i % 3) == 0;
            od.push_back({::Vector{x, y, up ? -4.f : 9.f}, ::Vector{0.01f * float(i % 7), 0.02f, up ? 1.f : -1.f}});
            many.push_back(State::Array({State::Vec(od.back().first.x, od.back().first.y, od.back().first.z),
                                         State::Vec(od.back().second.x, od.back().second.y, od.back().second.z)}));
        }
        CHECK(call_method(L, accelValue, "TraverseBatch", {State::Array(many)}) == 1);
        const fakelua::Value batch = L.stack.back();
        CHECK(batch.tab->kv.size() == 400);
        int nhit = 0;
        for (int i = 0; i < 400 && batch.tab->kv.size() == 400; ++i) {
            const int got = call_traverse(L, accel, {State::Vec(od[i].first.x, od[i].first.y, od[i].first.z),
                                                     State::Vec(od[i].second.x, od[i].second.y, od[i].second.z)});
            const fakelua::Value& bv = batch.tab->kv[size_t(i)].second;
            CHECK((got == 1) == (bv.type == TraceResult::id));
            if (got == 1 && bv.type == TraceResult::id) {
                TraceResult* a = L.GetUserType<TraceResult>(1, TraceResult::id);
                TraceResult* b = static_cast<TraceResult*>(*bv.ud);
                CHECK(a->primitiveIndex == b->primitiveIndex && a->distance == b->distance && a->uvw.x == b->uvw.x && a->uvw.y == b->uvw.y);
                ++nhit;
                delete a; delete b;
            }
        }
        CHECK(nhit > 10 && nhit < 400);
        L.Pop(L.Top());

        // The same 400 rays as ONE packed buffer -> ONE TraceResultBatch userdata: every getter with the ray's index
        // equals the per-ray TraceResult's getter (Pos: computed on the device, within 1e-5 relative)
        std::string packed(400 * sizeof(vt_ray), '\0');
        for (int i = 0; i < 400; ++i) {
            const vt_ray r{{od[i].first.x, od[i].first.y, od[i].first.z}, {od[i].second.x, od[i].second.y, od[i].second.z}, 0.f, FLT_MAX};
            std::memcpy(&packed[size_t(i) * sizeof(vt_ray)], &r, sizeof(r));
        }
        fakelua::Value bufv;
        bufv.type = LT::String; bufv.str = packed;
        CHECK(call_method(L, accelValue, "TraverseBatch", {bufv}) == 1);
        CHECK(L.Top() == 1 && L.GetType(1) == TraceResultBatch::id);
        const fakelua::Value rb = L.stack.back();
        CHECK(call_method(L, rb, "Count") == 1 && L.GetNumber(-1) == 400.0);
        CHECK(call_method(L, rb, "__tostring") == 1 && L.stack.back().str == "VisTraceResultBatch");
        CHECK(call_method(L, rb, "Hits") == 1 && L.stack.back().type == LT::String && L.stack.back().str.size() == 400 * sizeof(vt_hit));
        {   // the optional image-width hint (rays per row; scheduling only): same records, bad values are argument errors
            const std::string plain = L.stack.back().str;
            for (double w : {20.0, 16.0, 8.0, 7.0, 0.0}) {
                CHECK(call_method(L, accelValue, "TraverseBatch", {bufv, State::Num(w)}) == 1);
                const fakelua::Value hinted = L.stack.back();
                CHECK(call_method(L, hinted, "Hits") == 1 && L.stack.back().str == plain);
            }
            const std::string we = error_of([&] { call_method(L, accelValue, "TraverseBatch", {bufv, State::Num(2.5)}); });
            CHECK(we.find("imageWidth") != std::string::npos);
            const std::string wn = error_of([&] { call_method(L, accelValue, "TraverseBatch", {bufv, State::Num(-4)}); });
            CHECK(wn.find("imageWidth") != std::string::npos);
        }
        {   // a TABLE of buffers = a set of batches traced by one merged launch: each equals its single-buffer batch
            CHECK(call_method(L, rb, "Hits") == 1);
            const std::string whole = L.stack.back().str;
            std::vector<fakelua::Value> bufs;
            const size_t cuts[] = {0, 1, 130, 130, 256, 400};                // ragged, one empty
            for (size_t k = 0; k + 1 < sizeof(cuts) / sizeof(cuts[0]); ++k) {
                fakelua::Value v; v.type = LT::String; v.str = packed.substr(cuts[k] * sizeof(vt_ray), (cuts[k + 1] - cuts[k]) * sizeof(vt_ray));
                bufs.push_back(v);
            }
            CHECK(call_method(L, accelValue, "TraverseBatch", {State::Array(bufs), State::Array({State::Num(0), State::Num(8)})}) == 1);
            CHECK(L.Top() == 1 && L.GetType(1) == LT::Table);
            const fakelua::Value set = L.stack.back();
            CHECK(set.tab->kv.size() == bufs.size());
            std::string joined;
            for (auto& kv : set.tab->kv) {
                CHECK(kv.second.type == TraceResultBatch::id);
                CHECK(call_method(L, kv.second, "Hits") == 1);
                joined += L.stack.back().str;
            }
            CHECK(joined == whole);
            CHECK(call_method(L, set.tab->kv[4].second, "Count") == 1 && L.GetNumber(-1) == 144.0);
            CHECK(call_method(L, set.tab->kv[2].second, "Count") == 1 && L.GetNumber(-1) == 0.0);
            for (auto& kv : set.tab->kv) call_method(L, kv.second, "__gc");
            // a bad ray in the third buffer is reported with Traverse's message; a non-string member is an argument error
            std::vector<fakelua::Value> badset = bufs;
            vt_ray inv{{0, 0, 1}, {0, 0, -1}, 2.f, 2.f};
            std::memcpy(&badset[4].str[9 * sizeof(vt_ray)], &inv, sizeof(inv));
            CHECK(error_of([&] { call_method(L, accelValue, "TraverseBatch", {State::Array(badset)}); }) == "tMax must be greater than tMin");
            badset[4] = State::Num(3);
            int sarg = 0;
            CHECK(contains(error_of([&] { call_method(L, accelValue, "TraverseBatch", {State::Array(badset)}); }, &sarg), "strings only") && sarg == 2);
            L.Pop(L.Top());
        }
        int bhit = 0;
        auto close = [](float a, float b) { return std::fabs(a - b) <= 1e-5f * std::max(1.0f, std::fabs(b)); };
        for (int i = 0; i < 400; ++i) {
            const int got = call_traverse(L, accel, {State::Vec(od[i].first.x, od[i].first.y, od[i].first.z),
                                                     State::Vec(od[i].second.x, od[i].second.y, od[i].second.z)});
            const fakelua::Value one = got ? L.stack[0] : fakelua::Value();
            const fakelua::Value idx = State::Num(double(i + 1));
            CHECK(call_method(L, rb, "Hit", {idx}) == 1 && L.GetBool(-1) == (got == 1));
            if (got != 1) {
                CHECK(call_method(L, rb, "Pos", {idx}) == 0 && call_method(L, rb, "Distance", {idx}) == 0 && call_method(L, rb, "Get", {idx}) == 0);
                CHECK(call_method(L, rb, "Normal", {idx}) == 0 && call_method(L, rb, "Binormal", {idx}) == 0);
                continue;
            }
            ++bhit;
            TraceResult* a = static_cast<TraceResult*>(*one.ud);
            CHECK(call_method(L, rb, "Distance", {idx}) == 1 && float(L.GetNumber(-1)) == a->distance);
            CHECK(call_method(L, rb, "Pos", {idx}) == 1);
            const ::Vector bp = L.GetVector(-1);
            const Vec3& ap = a->GetPos();
            CHECK(close(bp.x, ap.x) && close(bp.y, ap.y) && close(bp.z, ap.z));
            CHECK(call_method(L, rb, "Barycentric", {idx}) == 1 && L.GetVector(-1).x == a->uvw.x && L.GetVector(-1).y == a->uvw.y);
            CHECK(call_method(L, rb, "GeometricNormal", {idx}) == 1 && close(L.GetVector(-1).z, a->geometricNormal.z));
            CHECK(call_method(L, rb, "Incident", {idx}) == 1 && close(L.GetVector(-1).z, a->wo.z) && close(L.GetVector(-1).x, a->wo.x));
            {   // the batch's frame comes from the device kernel, the single result's from the host class: same expression tree
                const Vec3 hn = a->GetNormal(), ht = a->GetTangent(), hb = a->GetBinormal();
                CHECK(call_method(L, rb, "Normal", {idx}) == 1);
                const ::Vector dn = L.GetVector(-1);
                CHECK(dn.x == hn.x && dn.y == hn.y && dn.z == hn.z);
                CHECK(call_method(L, rb, "Tangent", {idx}) == 1);
                const ::Vector dt = L.GetVector(-1);
                CHECK(dt.x == ht.x && dt.y == ht.y && dt.z == ht.z);
                CHECK(call_method(L, rb, "Binormal", {idx}) == 1);
                const ::Vector db = L.GetVector(-1);
                CHECK(db.x == hb.x && db.y == hb.y && db.z == hb.z);
            }
            CHECK(call_method(L, rb, "FrontFacing", {idx}) == 1 && L.GetBool(-1) == a->frontFacing);
            CHECK(call_method(L, rb, "SubMaterialIndex", {idx}) == 1 && L.GetNumber(-1) == double(a->submatIdx + 1));
            CHECK(call_method(L, rb, "MaterialFlags", {idx}) == 1 && L.GetNumber(-1) == double(a->GetMaterialFlags()));
            CHECK(call_method(L, rb, "SurfaceFlags", {idx}) == 1 && L.GetNumber(-1) == double(a->GetSurfFlags()));
            CHECK(call_method(L, rb, "HitSky", {idx}) == 1 && L.GetBool(-1) == a->hitSky);
            CHECK(call_method(L, rb, "HitWater", {idx}) == 1 && L.GetBool(-1) == a->HitWater());
            CHECK(call_method(L, rb, "TextureUV", {idx}) == 1 && L.GetType(-1) == LT::Table);
            {
                const auto& uvkv = L.stack.back().tab->kv;
                CHECK(uvkv.size() == 2 && close(float(uvkv[0].second.num), a->texUV.x) && close(float(uvkv[1].second.num), a->texUV.y));
            }
            CHECK(call_method(L, rb, "Entity", {idx}) == 1 && L.GetType(-1) == LT::Entity);
            void* be_ent = L.GetUserdataRaw(-1, LT::Entity);
            CHECK(call_method(L, one, "Entity") == 1 && L.GetUserdataRaw(-1, LT::Entity) == be_ent);
            CHECK(call_method(L, rb, "Get", {idx}) == 1 && L.GetType(-1) == TraceResult::id);
            TraceResult* g = static_cast<TraceResult*>(*L.stack.back().ud);
            CHECK(g->primitiveIndex == a->primitiveIndex && g->distance == a->distance && g->uvw.x == a->uvw.x && g->wo.z == a->wo.z);
            delete g;
            delete a;
            L.Pop(L.Top());
        }
        CHECK(bhit == nhit);
        // validation: whole records, the range checks of Traverse on every ray, index range
        std::string bad = packed.substr(0, 100);
        fakelua::Value badv; badv.type = LT::String; badv.str = bad;
        int barg = 0;
        std::string bmsg = error_of([&] { call_method(L, accelValue, "TraverseBatch", {badv}); }, &barg);
        CHECK(contains(bmsg, "whole 32-byte records") && barg == 2);
        vt_ray neg{{0, 0, 1}, {0, 0, -1}, -1.f, 5.f};
        std::memcpy(&packed[7 * sizeof(vt_ray)], &neg, sizeof(neg));
        bufv.str = packed;
        CHECK(error_of([&] { call_method(L, accelValue, "TraverseBatch", {bufv}); }) == "tMin cannot be less than 0");
        bmsg = error_of([&] { call_method(L, rb, "Pos", {State::Num(401)}); }, &barg);
        CHECK(contains(bmsg, "index out of range") && barg == 2);
        bmsg = error_of([&] { call_method(L, rb, "Pos", {State::Num(0)}); }, &barg);
        CHECK(contains(bmsg, "index out of range"));
        // the batch outlives a Rebuild of its accel (it shares the tables it was traced against) and its own __gc is final
        CHECK(call_method(L, accelValue, "Rebuild", {State::Nil(), State::Bool(false)}) == 0);
        CHECK(call_method(L, rb, "SubMaterialIndex", {State::Num(1)}) <= 1);
        CHECK(call_method(L, rb, "__gc") == 0);
        CHECK(contains(error_of([&] { call_method(L, rb, "Count"); }), "released"));
        CHECK(call_method(L, accelValue, "Rebuild", {State::Array({State::User(&dummyEntity, LT::Entity)})}) == 0);
        L.Pop(L.Top());
    }

    // Rebuild(nil, false): no world, no entities -> empty but valid accel, every trace misses
    {
        CHECK(call_method(L, accelValue, "Rebuild", {State::Nil(), State::Bool(false)}) == 0);
        CHECK(accel->IsBuilt() && accel->TriangleCount() == 0);
        CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1)}) == 0);
        // Rebuild({ent}) with the world again
        CHECK(call_method(L, accelValue, "Rebuild", {State::Array({State::User(&dummyEntity, LT::Entity), State::User(&dummyEntity, LT::Entity)})}) == 0);
        CHECK(accel->TriangleCount() == 4 && src.calls == 4);
        CHECK(call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1)}) == 1);
        delete L.GetUserType<TraceResult>(1, TraceResult::id);
    }
    // alpha-tested material (Primitives.h:196-208): a fence triangle whose 2x2 checker alpha plane [[0,255],[255,0]]
    // lets the ray through where alpha < alphatestreference and stops it elsewhere
    {
        Material fence{"brush/fence", MATFLAG_ALPHATEST};
        fence.alphaWidth = fence.alphaHeight = 2;
        fence.baseAlpha = {0, 255, 255, 0};
        world.materials.push_back(fence);
        world.triangles.push_back(make_tri({40, 0, 0}, {50, 0, 0}, {40, 10, 0}, false, 2));   // uvs (0,0) (1,0) (0,1)
        CHECK(call_method(L, accelValue, "Rebuild") == 0);
        CHECK(accel->IsBuilt());
        // barycentric (u,v) = (0.2,0.2) -> texel (0,0): alpha 0 -> the ray passes through the fence
        CHECK(call_traverse(L, accel, {State::Vec(42, 2, 8), State::Vec(0, 0, -1)}) == 0);
        // (0.7,0.2) -> texel (1,0): alpha 1 -> hit
        CHECK(call_traverse(L, accel, {State::Vec(47, 2, 8), State::Vec(0, 0, -1)}) == 1);
        delete L.GetUserType<TraceResult>(1, TraceResult::id);
        // a reference of 0 keeps every hit (alpha < 0 never holds)
        world.materials.back().alphatestreference = 0.f;
        CHECK(call_method(L, accelValue, "Rebuild") == 0);
        CHECK(call_traverse(L, accel, {State::Vec(42, 2, 8), State::Vec(0, 0, -1)}) == 1);
        delete L.GetUserType<TraceResult>(1, TraceResult::id);
        world.triangles.pop_back();
        world.materials.pop_back();
    }
    CHECK(call_method(L, accelValue, "__tostring") == 1 && L.stack.back().str == "AccelStruct");
    CHECK(call_method(L, accelValue, "__gc") == 0);
    SetWorld(nullptr);
    AccelStruct::SetEntityMeshSource(nullptr);
}

// BASELINE config 1 through the binding: single-ray accel:Traverse calls (10 k-triangle world mesh, 10 k seeded
// rays), the way a GLua script issues them; prints microseconds per call, TraceResult construction included.
static void bench_single_calls()
{
    State L;
    RegisterTracingApi(&L);
    World world;
    world.materials.push_back(Material{"brush/floor", MATFLAG_NONE});
    world.entities.push_back(Entity{nullptr, 0});
    const int k = 71;                                           // 71 x 71 x 2 = 10 082 triangles
    auto height = [](int i, int j) { return 6.0f * float((i * 7 + j * 13) % 11) / 11.0f; };
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            const float x0 = float(i) * 4 - 142, x1 = x0 + 4, y0 = float(j) * 4 - 142, y1 = y0 + 4;
            world.triangles.push_back(make_tri({x0, y0, height(i, j)}, {x1, y0, height(i + 1, j)}, {x0, y1, height(i, j + 1)}, false, 0));
            world.triangles.push_back(make_tri({x1, y0, height(i + 1, j)}, {x1, y1, height(i + 1, j + 1)}, {x0, y1, height(i, j + 1)}, false, 0));
        }
    SetWorld(&world);
    L.PushValue(State::Array({}));
    L.PushBool(true);
    CHECK(vistrace_CreateAccel(&L) == 1);
    AccelStruct* accel = L.GetUserType<AccelStruct>(-1, AccelStruct_id);
    CHECK(accel && accel->IsBuilt() && accel->TriangleCount() == size_t(2 * k * k));
    uint64_t rng = 0x5EEDull;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return float(rng >> 40) / 16777216.0f; };
    const int calls = 10000;
    int hits = 0;
    for (int pass = 0; pass < 2; ++pass) {                      // pass 0 warms up
        hits = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int c = 0; c < calls; ++c) {
            const float dx = next() * 2 - 1, dy = next() * 2 - 1, dz = -(0.2f + next());
            const int got = call_traverse(L, accel, {State::Vec(next() * 40 - 20, next() * 40 - 20, 60), State::Vec(dx, dy, dz)});
            if (got == 1) { ++hits; delete L.GetUserType<TraceResult>(1, TraceResult::id); }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
        if (pass) std::printf("config 1 through the binding: %d accel:Traverse calls, %d hits, %.1f us per call\n", calls, hits, us);
    }
    CHECK(hits > calls / 2);
    // host walk against the device for small batches: where does a launch start to pay?  (AccelStruct::kDeviceBatchMin)
    {
        std::vector<vt_ray> rays(4096);
        for (auto& r : rays) {
            r = vt_ray{{next() * 40 - 20, next() * 40 - 20, 60}, {next() * 2 - 1, next() * 2 - 1, -(0.2f + next())}, 0.f, FLT_MAX};
        }
        std::vector<vt_hit> h_host(rays.size()), h_dev(rays.size());
        for (uint64_t n : {1ull, 4ull, 8ull, 16ull, 32ull, 64ull, 256ull, 4096ull}) {
            double us[2] = {0, 0};
            for (int side = 0; side < 2; ++side) {
                const int reps = n >= 4096 ? 50 : 2000;
                for (int pass = 0; pass < 2; ++pass) {
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int r = 0; r < reps; ++r) {
                        const vt_ray* src = rays.data() + (uint64_t(r) * n) % (rays.size() - n + 1);
                        const int rc = side == 0 ? accel->TraceClosestHost(src, n, h_host.data()) : accel->TraceClosestDevice(src, n, h_dev.data());
                        if (rc != VT_OK) { ++g_fail; std::printf("FAIL trace rc %d: %s\n", rc, vt_last_error()); r = reps; }
                    }
                    us[side] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
                }
            }
            CHECK(accel->TraceClosestHost(rays.data(), n, h_host.data()) == VT_OK && accel->TraceClosestDevice(rays.data(), n, h_dev.data()) == VT_OK);
            CHECK(std::memcmp(h_host.data(), h_dev.data(), n * sizeof(vt_hit)) == 0);    // host walk == device, bit for bit
            std::printf("batch of %5llu rays: host walk %8.2f us, device %8.2f us per call\n", (unsigned long long)n, us[0], us[1]);
        }
    }
    L.Pop(L.Top());
    L.PushUserType(accel, AccelStruct_id);
    CHECK(AccelStruct_gc(&L) == 0);
    SetWorld(nullptr);
}

// The batch boundary end to end, through the fake Lua state: accel:TraverseBatch(table of ray tables) -> table of
// TraceResults (the per-ray cost of the reference's interface: N table parses, N constructors) against
// accel:TraverseBatch(packed buffer) -> one TraceResultBatch, then one getter over every ray.
static void bench_batch_forms()
{
    // The fake VM frees a Lua string the moment it leaves the stack; with glibc's defaults a 32-MB block is then unmapped page by
    // page (1-2 ms inside the timed call).  A Lua VM's allocator keeps such blocks: tell malloc to do the same.
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    State L;
    RegisterTracingApi(&L);
    World world;
    world.materials.push_back(Material{"brush/floor", MATFLAG_NONE});
    world.entities.push_back(Entity{nullptr, 0});
    const int k = 224;                                          // 224 x 224 x 2 = 100 352 triangles
    auto height = [](int i, int j) { return 6.0f * float((i * 7 + j * 13) % 11) / 11.0f; };
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            const float x0 = float(i) * 4 - 448, x1 = x0 + 4, y0 = float(j) * 4 - 448, y1 = y0 + 4;
            world.triangles.push_back(make_tri({x0, y0, height(i, j)}, {x1, y0, height(i + 1, j)}, {x0, y1, height(i, j + 1)}, false, 0));
            world.triangles.push_back(make_tri({x1, y0, height(i + 1, j)}, {x1, y1, height(i + 1, j + 1)}, {x0, y1, height(i, j + 1)}, false, 0));
        }
    SetWorld(&world);
    L.PushValue(State::Array({}));
    L.PushBool(true);
    CHECK(vistrace_CreateAccel(&L) == 1);
    const fakelua::Value accelValue = L.stack.back();
    uint64_t rng = 0xBA7C4ull;
    auto next = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return float(rng >> 40) / 16777216.0f; };
    const size_t nbuf = size_t(1) << 20, ntab = size_t(1) << 17;
    std::vector<vt_ray> rays(nbuf);
    for (vt_ray& r : rays) r = vt_ray{{next() * 800 - 400, next() * 800 - 400, 40.f}, {next() - 0.5f, next() - 0.5f, -1.f}, 0.f, FLT_MAX};
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };

    // buffer form: 1 Mi rays.  The fake VM copies a Value (std::string and all) wherever a Lua VM passes a reference, so the
    // 32-MB string is moved onto the stack by hand and a getter's two arguments are set in place.
    std::string packed(reinterpret_cast<const char*>(rays.data()), nbuf * sizeof(vt_ray));
    const fakelua::Value* traverse = L.find_method(accelValue.type, "TraverseBatch");
    {   // warm-up (first launch, allocations)
        fakelua::Value bufv; bufv.type = LT::String; bufv.str = packed;
        call_method(L, accelValue, "TraverseBatch", {bufv});
        const fakelua::Value w = L.stack.back();
        call_method(L, w, "__gc");
    }
    double best_call = 1e9, best_hits = 1e9, best_dist = 1e9, best_pos = 1e9;
    double sum = 0;
    for (int rep = 0; rep < 4; ++rep) {
        L.stack.clear();
        L.stack.push_back(accelValue);
        L.stack.emplace_back();
        L.stack.back().type = LT::String;
        L.stack.back().str.swap(packed);
        const auto t0 = clk::now();
        CHECK(traverse->fn(&L) == 1);
        const auto t1 = clk::now();
        const fakelua::Value rb = L.stack.back();
        packed.assign(reinterpret_cast<const char*>(rays.data()), nbuf * sizeof(vt_ray));
        // (a) the bulk consumer: every hit record as one packed string
        const fakelua::Value* hitsfn = L.find_method(rb.type, "Hits");
        if (rep == 3) {                                          // last repetition: where does Hits() spend its time?
            const fakelua::Value* d1 = L.find_method(rb.type, "Distance");
            L.stack.clear(); L.stack.push_back(rb); L.stack.push_back(State::Num(1));
            const auto a0 = clk::now();
            d1->fn(&L);                                          // waits for the kernels, downloads the hit records
            const auto a1 = clk::now();
            L.stack.clear(); L.stack.push_back(rb);
            hitsfn->fn(&L);                                      // now only the copy into a (fake) Lua string
            const auto a2 = clk::now();
            std::printf("  (Hits() = wait for the kernels + download of %zu MB: %.2f ms; + copy into the Lua string: %.2f ms)\n",
                        nbuf * sizeof(vt_hit) >> 20, secs(a0, a1) * 1e3, secs(a1, a2) * 1e3);
            L.stack.clear(); L.stack.push_back(rb);
        }
        L.stack.clear(); L.stack.push_back(rb);
        const auto t2 = clk::now();
        CHECK(hitsfn->fn(&L) == 1 && L.stack.back().str.size() == nbuf * sizeof(vt_hit));
        const auto t3 = clk::now();
        sum += double(static_cast<unsigned char>(L.stack.back().str[5]));
        // (b) one scalar per ray through the getter
        const fakelua::Value* dist = L.find_method(rb.type, "Distance");
        L.stack.clear(); L.stack.push_back(rb); L.stack.push_back(State::Num(1));
        const auto t4 = clk::now();
        for (size_t i = 0; i < nbuf; ++i) {
            L.stack[1].num = double(i + 1);
            if (dist->fn(&L) == 1) { sum += L.stack.back().num; L.stack.pop_back(); }
        }
        const auto t5 = clk::now();
        const fakelua::Value* pos = L.find_method(rb.type, "Pos");
        for (size_t i = 0; i < nbuf; ++i) {
            L.stack[1].num = double(i + 1);
            if (pos->fn(&L) == 1) { sum += L.stack.back().vec.z; L.stack.pop_back(); }
        }
        const auto t6 = clk::now();
        call_method(L, rb, "__gc");
        best_call = std::min(best_call, secs(t0, t1));
        best_hits = std::min(best_hits, secs(t0, t1) + secs(t2, t3));
        best_dist = std::min(best_dist, secs(t4, t5));
        best_pos = std::min(best_pos, secs(t5, t6));
    }
    std::printf("batch boundary, buffer form, %zu rays: TraverseBatch call %.2f ms (checks + upload; trace and result kernels enqueued); "
                "call + Hits() (all hit records back as one string) %.2f ms = %.1f Mrays/s end to end; Distance(i) for every ray +%.2f ms, "
                "Pos(i) for every ray +%.2f ms (fake VM calls included)\n",
                nbuf, best_call * 1e3, best_hits * 1e3, double(nbuf) / best_hits / 1e6, best_dist * 1e3, best_pos * 1e3);

    // table form: 128 Ki rays (a Lua table per ray in, a TraceResult per hit out)
    std::vector<fakelua::Value> many;
    many.reserve(ntab);
    for (size_t i = 0; i < ntab; ++i)
        many.push_back(State::Array({State::Vec(rays[i].org[0], rays[i].org[1], rays[i].org[2]), State::Vec(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2])}));
    const fakelua::Value tabv = State::Array(many);
    double best_tab = 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        const auto t0 = clk::now();
        CHECK(call_method(L, accelValue, "TraverseBatch", {tabv}) == 1);
        const fakelua::Value res = L.stack.back();
        for (auto& kv : res.tab->kv)
            if (kv.second.type == TraceResult::id) { TraceResult* r = static_cast<TraceResult*>(*kv.second.ud); sum += r->distance; }
        const auto t1 = clk::now();
        for (auto& kv : res.tab->kv)
            if (kv.second.type == TraceResult::id) delete static_cast<TraceResult*>(*kv.second.ud);
        best_tab = std::min(best_tab, secs(t0, t1));
        L.Pop(L.Top());
    }
    std::printf("batch boundary, table form, %zu rays: TraverseBatch + distance of every result %.2f ms = %.3f Mrays/s end to end (checksum %.3g)\n",
                ntab, best_tab * 1e3, double(ntab) / best_tab / 1e6, sum);
    CHECK(call_method(L, accelValue, "__gc") == 0);
    SetWorld(nullptr);
}

// ---- fault injection through the Lua surface (needs VT_ENABLE_TEST_HOOKS=1; tests/test_gpu_fault_injection.py) ---------------
// Every device / pinned allocation behind vistrace.CreateAccel, accel:Rebuild and accel:TraverseBatch is made to fail in turn
// (vt_test_fail_alloc).  The reference's convention on such paths is delete-before-throw (source/VisTrace.cpp:782-785,
// source/objects/AccelStruct.cpp:186-203, :780): the script sees a Lua error, never an abort, and the module stays usable.
// the hook under test: allocations (--fail-alloc) or every other checked HIP call (--fail-hip: vt_test_fail_hip)
static int (*g_arm)(uint64_t) = vt_test_fail_alloc;
static uint64_t (*g_passed)(void) = vt_test_alloc_count;

static void test_fail_alloc_side()
{
    const bool hip = g_arm == vt_test_fail_hip;
    State L;
    RegisterTracingApi(&L);
    install_entity_global(L);
    FakeMeshSource src;
    AccelStruct::SetEntityMeshSource(&src);
    World world;
    world.materials.push_back(Material{"brush/floor", MATFLAG_NONE});
    world.entities.push_back(Entity{nullptr, 0});
    world.triangles.push_back(make_tri({0, 0, 0}, {10, 0, 0}, {0, 10, 0}, true, 0));
    world.triangles.push_back(make_tri({20, 0, 0}, {30, 0, 0}, {20, 10, 0}, true, 0));
    SetWorld(&world);
    int dummyEntity = 0;
    g_entityByIndex[42] = &dummyEntity;
    if (vt_test_fail_alloc(0) != VT_OK || vt_test_fail_hip(0) != VT_OK) { ++g_fail; std::printf("FAIL: --fail-alloc / --fail-hip need VT_ENABLE_TEST_HOOKS=1\n"); return; }

    auto create = [&]() -> fakelua::Value {
        L.Pop(L.Top());
        L.PushValue(State::Array({State::User(&dummyEntity, LT::Entity)}));
        L.PushBool(true);
        if (L.find_global("vistrace", "CreateAccel")->fn(&L) != 1) return State::Nil();
        return L.stack.back();
    };
    auto works = [&](const fakelua::Value& accelValue) {       // the entity quad from above, through the single-ray path
        AccelStruct* accel = static_cast<AccelStruct*>(*accelValue.ud);
        if (call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1)}) != 1 || L.GetType(1) != TraceResult::id) return false;
        return L.GetUserType<TraceResult>(1, TraceResult::id)->distance == 3.0f;
    };
    // 400 rays: above AccelStruct::kDeviceBatchMin, so the batch goes to the device (staging pipeline, batch block, launch scratch)
    std::vector<fakelua::Value> many;
    for (int i = 0; i < 400; ++i)
        many.push_back(State::Array({State::Vec(float((i * 37) % 320) * 0.1f - 1.f, float((i * 53) % 120) * 0.1f - 1.f, 9.f), State::Vec(0.01f * float(i % 7), 0.02f, -1.f)}));
    auto batch_hits = [&](const fakelua::Value& accelValue) -> int {
        if (call_method(L, accelValue, "TraverseBatch", {State::Array(many)}) != 1) return -1;
        int nhit = 0;
        for (auto& kv : L.stack.back().tab->kv)
            if (kv.second.type == TraceResult::id) { ++nhit; delete static_cast<TraceResult*>(*kv.second.ud); }
        return nhit;
    };

    // a clean run counts the allocations of each call (the first CreateAccel also opens the engine)
    fakelua::Value ref = create();
    const uint64_t n_create = g_passed();
    CHECK(ref.type == AccelStruct_id && works(ref) && n_create >= 4);
    (void)g_arm(0);
    const int ref_hits = batch_hits(ref);
    const uint64_t n_batch = g_passed();
    CHECK(ref_hits > 0 && n_batch >= 1);
    int failed_create = 0, failed_rebuild = 0, failed_batch = 0;
    for (uint64_t k = 1; k <= n_create; ++k) {               // vistrace.CreateAccel with its k-th allocation failing
        (void)g_arm(k);
        fakelua::Value got;
        const std::string e = error_of([&] { got = create(); });
        (void)g_arm(0);
        if (!e.empty()) { ++failed_create; CHECK(contains(e.c_str(), "VisTrace:")); }
        else { CHECK(got.type == AccelStruct_id && works(got)); CHECK(call_method(L, got, "__gc") == 0); }
        CHECK(works(ref) && batch_hits(ref) == ref_hits);    // the module is still usable, the older accel untouched
    }
    for (uint64_t k = 1; k <= n_create; ++k) {               // accel:Rebuild: a failure leaves the accel invalid, the next Rebuild heals it
        fakelua::Value a = create();
        (void)g_arm(k);
        const std::string e = error_of([&] { call_method(L, a, "Rebuild", {State::Array({State::User(&dummyEntity, LT::Entity)}), State::Bool(true)}); });
        (void)g_arm(0);
        if (!e.empty()) {
            ++failed_rebuild;
            AccelStruct* accel = static_cast<AccelStruct*>(*a.ud);
            const std::string t = error_of([&] { call_traverse(L, accel, {State::Vec(1, 1, 8), State::Vec(0, 0, -1)}); });
            CHECK(contains(t.c_str(), "acceleration structure invalid"));
            CHECK(error_of([&] { call_method(L, a, "Rebuild", {State::Array({State::User(&dummyEntity, LT::Entity)}), State::Bool(true)}); }).empty());
        }
        CHECK(works(a));
        CHECK(call_method(L, a, "__gc") == 0);
    }
    // accel:TraverseBatch(packed buffer) -> vt_batch_trace_closest_ex.  The engine keeps what a batch allocated (staging pipeline,
    // spare device and pinned blocks), so a repeated call of the same size allocates nothing: every round asks for a batch 3 x
    // larger than the last, which needs a new device block and a new pinned block for its hit records -- allocations 1 and 2 of
    // the call (the staging pipeline and the launch scratch exist since the clean call above).
    // (--fail-hip: every checked HIP call of one TraverseBatch of 40 000 rays in turn, the size kept)
    size_t rays_n = 40000;
    uint64_t k_max = 2;
    int rounds = 2, injected_batch = 0;
    if (hip) {
        std::string packed(rays_n * sizeof(vt_ray), '\0');
        for (size_t i = 0; i < rays_n; ++i) {
            const vt_ray r{{float((i * 37) % 320) * 0.1f - 1.f, float((i * 53) % 120) * 0.1f - 1.f, 9.f}, {0.01f * float(i % 7), 0.02f, -1.f}, 0.f, FLT_MAX};
            std::memcpy(&packed[i * sizeof(vt_ray)], &r, sizeof(r));
        }
        fakelua::Value bufv;
        bufv.type = LT::String; bufv.str = packed;
        for (int warm = 0; warm < 2; ++warm) {
            (void)g_arm(0);
            CHECK(call_method(L, ref, "TraverseBatch", {bufv}) == 1 && L.GetType(1) == TraceResultBatch::id);
            const fakelua::Value rb = L.stack.back();
            CHECK(call_method(L, rb, "Hits") == 1);
            (void)call_method(L, rb, "__gc");
            k_max = g_passed();
        }
        rounds = 1;
    }
    for (uint64_t k = 1; k <= k_max; ++k) {
        for (int round = 0; round < rounds; ++round, rays_n *= (hip ? 1 : 3)) {
            ++injected_batch;
            std::string packed(rays_n * sizeof(vt_ray), '\0');
            for (size_t i = 0; i < rays_n; ++i) {
                const vt_ray r{{float((i * 37) % 320) * 0.1f - 1.f, float((i * 53) % 120) * 0.1f - 1.f, 9.f}, {0.01f * float(i % 7), 0.02f, -1.f}, 0.f, FLT_MAX};
                std::memcpy(&packed[i * sizeof(vt_ray)], &r, sizeof(r));
            }
            fakelua::Value bufv;
            bufv.type = LT::String; bufv.str = packed;
            auto hits_of = [&]() -> long {
                if (call_method(L, ref, "TraverseBatch", {bufv}) != 1 || L.GetType(1) != TraceResultBatch::id) return -1;
                const fakelua::Value rb = L.stack.back();
                if (call_method(L, rb, "Hits") != 1 || L.stack.back().str.size() != rays_n * sizeof(vt_hit)) return -1;
                const vt_hit* h = reinterpret_cast<const vt_hit*>(L.stack.back().str.data());
                long nhit = 0;
                for (size_t i = 0; i < rays_n; ++i) nhit += h[i].prim != VT_MISS;
                (void)call_method(L, rb, "__gc");
                return nhit;
            };
            (void)g_arm(k);
            long got = -2;
            const std::string e = error_of([&] { got = hits_of(); });
            (void)g_arm(0);
            const long clean = hits_of();                     // the same buffer once more: now it must work
            CHECK(clean > 0);
            if (!e.empty()) { ++failed_batch; CHECK(contains(e.c_str(), "VisTrace:")); }
            else CHECK(got == clean);
            CHECK(batch_hits(ref) == ref_hits && works(ref));
        }
    }
    std::printf("%s through Lua: CreateAccel %d of %llu (the first call also opened the engine: later ones pass fewer), Rebuild %d, "
                "TraverseBatch %d of %d injected failures surfaced as Lua errors\n", hip ? "fail-hip" : "fail-alloc",
                failed_create, (unsigned long long)n_create, failed_rebuild, failed_batch, injected_batch);
    CHECK(failed_create >= 3 && failed_rebuild >= 3 && failed_batch >= 2);
    CHECK(call_method(L, ref, "__gc") == 0);
    SetWorld(nullptr);
}

int main(int argc, char** argv)
{
    if (argc > 1 && (std::strcmp(argv[1], "--fail-alloc") == 0 || std::strcmp(argv[1], "--fail-hip") == 0)) {
        if (std::strcmp(argv[1], "--fail-hip") == 0) { g_arm = vt_test_fail_hip; g_passed = vt_test_hip_count; }
        test_fail_alloc_side();
        std::printf("binding (%s): %d checks, %d failed\n", argv[1] + 2, g_run, g_fail);
        return g_fail ? 1 : 0;
    }
    if (argc > 1 && std::strcmp(argv[1], "--bench") == 0) {
        bench_single_calls();
        bench_batch_forms();
        std::printf("binding (bench): %d checks, %d failed\n", g_run, g_fail);
        return g_fail ? 1 : 0;
    }
    const bool cpu_only = argc > 1 && std::strcmp(argv[1], "--cpu") == 0;
    test_cpu_side();
    if (!cpu_only) test_gpu_side();
    std::printf("%s: %d checks, %d failed\n", cpu_only ? "binding (cpu)" : "binding (cpu+gpu)", g_run, g_fail);
    return g_fail ? 1 : 0;
}
