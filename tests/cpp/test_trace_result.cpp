// test_trace_result.cpp -- the host-side TraceResult class (vistrace_amd/csrc/host/TraceResult.cpp: what one accel:Traverse call
// hands to Lua) against the CPU oracle, on the CPU: every field the constructor derives (source/objects/TraceResult.cpp:45-86),
// GetPos (:255-262), the shading frame (CalcTBN without a normal map, :132-186) and the cone footprint (CalcFootprint, :89-103)
// for random triangles, frames, uvs and hit points -- bit for bit (both sides are plain fp32 without contraction; log2 is the
// same libm call).  Test infrastructure: links oracle/vt_oracle.c, which the product never does.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "TraceResult.h"
extern "C" {
#include "vt_oracle.h"
}

using namespace vistrace;

static int g_fail = 0, g_run = 0;
#define CHECK(cond) do { ++g_run; if (!(cond)) { if (++g_fail < 20) std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

static bool same(float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (std::isnan(a) && std::isnan(b)); }
static bool same3(const Vec3& a, const float b[3]) { return same(a.x, b[0]) && same(a.y, b[1]) && same(a.z, b[2]); }

int main()
{
    std::mt19937 rng(20260401);
    std::uniform_real_distribution<float> pos(-50.f, 50.f), unit(0.f, 1.f), sym(-1.f, 1.f);
    int grazing = 0, cone_on = 0;
    for (int it = 0; it < 20000; ++it) {
        Triangle tri;
        tri.p0 = Vec3{pos(rng), pos(rng), pos(rng)};
        const float size = std::pow(10.f, sym(rng) * 1.5f);
        tri.p1 = Vec3{tri.p0.x + sym(rng) * size, tri.p0.y + sym(rng) * size, tri.p0.z + sym(rng) * size};
        tri.p2 = Vec3{tri.p0.x + sym(rng) * size, tri.p0.y + sym(rng) * size, tri.p0.z + sym(rng) * size};
        float normals[9], tangents[9], uvs[6];
        for (int k = 0; k < 3; ++k) {
            tri.normals[k] = Vec3{sym(rng), sym(rng), sym(rng)};
            tri.tangents[k] = Vec3{sym(rng), sym(rng), sym(rng)};
            tri.uvs[k] = Vec2{sym(rng) * 3.f, sym(rng) * 3.f};
            tri.alphas[k] = unit(rng);
            normals[k * 3] = tri.normals[k].x; normals[k * 3 + 1] = tri.normals[k].y; normals[k * 3 + 2] = tri.normals[k].z;
            tangents[k * 3] = tri.tangents[k].x; tangents[k * 3 + 1] = tri.tangents[k].y; tangents[k * 3 + 2] = tri.tangents[k].z;
            uvs[k * 2] = tri.uvs[k].x; uvs[k * 2 + 1] = tri.uvs[k].y;
        }
        tri.material = 3;
        const float p0[3] = {tri.p0.x, tri.p0.y, tri.p0.z}, p1[3] = {tri.p1.x, tri.p1.y, tri.p1.z}, p2[3] = {tri.p2.x, tri.p2.y, tri.p2.z};
        vto_tri ot;
        vto_tri_setup(p0, p1, p2, 0, &ot);
        // a hit point inside the triangle, a direction (every tenth almost inside the triangle's plane: the grazing branch)
        float u = unit(rng), v = unit(rng);
        if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }
        float dir[3] = {sym(rng), sym(rng), sym(rng)};
        if (it % 10 == 0) {
            const float e[3] = {tri.p1.x - tri.p0.x, tri.p1.y - tri.p0.y, tri.p1.z - tri.p0.z};
            for (int k = 0; k < 3; ++k) dir[k] = e[k] + 0.05f * sym(rng) * ot.n[k] / std::sqrt(ot.n[0] * ot.n[0] + ot.n[1] * ot.n[1] + ot.n[2] * ot.n[2] + 1e-30f);
        }
        const float dist = unit(rng) * 100.f;
        const bool cone = it % 3 != 0;
        const float cw = cone ? unit(rng) : -1.f, ca = cone ? 0.001f + unit(rng) * 0.05f : -1.f;
        cone_on += cone;
        Entity ent; ent.id = 77;
        Material mat;
        TraceResult r(Vec3{dir[0], dir[1], dir[2]}, dist, cw, ca, tri, 5, Vec2{u, v}, ent, mat);

        vto_attrs oa;
        vto_hit_attrs(&ot, dir, u, v, &oa);
        CHECK(same3(r.wo, oa.wo) && same3(r.geometricNormal, oa.ngeo) && same3(r.uvw, oa.uvw) && same3(r.GetPos(), oa.pos));
        CHECK(r.frontFacing == (oa.front != 0) && r.distance == dist && r.entIdx == 77 && r.submatIdx == 3 && r.primitiveIndex == 5);
        float tex[2], blend;
        vto_hit_shade(u, v, uvs, tri.alphas, tex, &blend);
        CHECK(same(r.texUV.x, tex[0]) && same(r.texUV.y, tex[1]) && same(r.blendFactor, blend));
        vto_tbn ob;
        vto_hit_tbn(&ot, dir, dist, u, v, normals, tangents, uvs, cw, ca, &ob);
        CHECK(same3(r.GetNormal(), ob.normal) && same3(r.GetTangent(), ob.tangent) && same3(r.GetBinormal(), ob.binormal));
        Vec2 lod;
        const bool set = r.GetTextureLodInfo(lod);
        CHECK(set == (ob.lod_set != 0) && set == cone);
        if (set) CHECK(same(lod.x, ob.lod_info[0]) && same(lod.y, ob.lod_info[1]));
        // which branch CalcTBN took (recomputed from the interpolated normal before the correction)
        const float w = 1.f - u - v;
        float n[3];
        for (int k = 0; k < 3; ++k) n[k] = (w * normals[k] + u * normals[3 + k]) + v * normals[6 + k];
        const float inv = 1.f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        grazing += std::fabs((oa.wo[0] * n[0] * inv + oa.wo[1] * n[1] * inv) + oa.wo[2] * n[2] * inv) <= 0.1f;
    }
    CHECK(grazing > 500 && cone_on > 10000);

    // ---- the branch points themselves (round 6: scripts/mutants_host.sh showed the random cases never sit ON one) ---------------
    {
        Triangle tri;
        tri.p0 = Vec3{0, 0, 0}; tri.p1 = Vec3{1, 0, 0}; tri.p2 = Vec3{0, 1, 0};          // n = cross(p0 - p1, p2 - p0) = (0, 0, -1)
        float normals[9], tangents[9], uvs[6] = {0, 0, 1, 0, 0, 1};
        for (int k = 0; k < 3; ++k) {
            tri.normals[k] = Vec3{0, 0, 1}; tri.tangents[k] = Vec3{1, 0, 0}; tri.uvs[k] = Vec2{uvs[2 * k], uvs[2 * k + 1]}; tri.alphas[k] = 0.5f;
            normals[3 * k] = 0; normals[3 * k + 1] = 0; normals[3 * k + 2] = 1; tangents[3 * k] = 1; tangents[3 * k + 1] = 0; tangents[3 * k + 2] = 0;
        }
        tri.material = 0;
        const float p0[3] = {0, 0, 0}, p1[3] = {1, 0, 0}, p2[3] = {0, 1, 0};
        vto_tri ot;
        vto_tri_setup(p0, p1, p2, 0, &ot);
        Entity ent; ent.id = 1;
        Material mat;
        auto agree = [&](const float dir[3], float cw, float ca, bool want_front, int want_lod) {
            TraceResult r(Vec3{dir[0], dir[1], dir[2]}, 2.f, cw, ca, tri, 0, Vec2{0.25f, 0.25f}, ent, mat);
            vto_attrs oa;
            vto_hit_attrs(&ot, dir, 0.25f, 0.25f, &oa);
            vto_tbn ob;
            vto_hit_tbn(&ot, dir, 2.f, 0.25f, 0.25f, normals, tangents, uvs, cw, ca, &ob);
            CHECK(r.frontFacing == (oa.front != 0) && r.frontFacing == want_front);
            CHECK(same3(r.GetNormal(), ob.normal) && same3(r.GetTangent(), ob.tangent) && same3(r.GetBinormal(), ob.binormal));
            Vec2 lod;
            const bool set = r.GetTextureLodInfo(lod);
            CHECK(set == (ob.lod_set != 0) && int(set) == want_lod);
            return r.wo.z;
        };
        // frontFacing = dot(wo, geometricNormal) >= 0 (TraceResult.cpp:85) AT zero: rays inside the triangle's plane
        const float flat[3][3] = {{1, 0, 0}, {0, -2, 0}, {3, 4, 0}};
        for (const float* d : flat) agree(d, -1.f, -1.f, true, 0);
        // mipOverride = coneWidth < 0 || coneAngle <= 0 (:54): an angle of exactly 0 switches the footprint off, a tiny one does not
        const float down[3] = {0.1f, 0.2f, -1.f};
        agree(down, 0.25f, 0.f, false, 0);
        agree(down, 0.25f, 1e-30f, false, 1);
        agree(down, 0.f, 0.004f, false, 1);                                               // a width of exactly 0 is a valid cone
        agree(down, -1e-30f, 0.004f, false, 0);
        // the grazing correction runs on cosTheta <= 0.1 (:175-176): a direction whose normalised z is 0.1f to the bit, found by search
        int found = 0;
        for (int i = -2000; i <= 2000 && !found; ++i) {
            float a = 0.99498743f;
            for (int s = 0; s < std::abs(i); ++s) a = std::nextafter(a, i < 0 ? 0.f : 2.f);
            const float d[3] = {a, 0.f, -0.1f};
            const float inv = 1.0f / std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            if (-(d[2] * inv) == 0.1f) {
                found = 1;
                // shading normal (0, 0, 1), wo.z = 0.1f exactly: cosTheta == the threshold -> corrected.  With a vertex tangent that is
                // NOT perpendicular to the normal the correction shows (tangent re-orthogonalised to (1, 0, 0), binormal = cross(tangent,
                // normal)); just above the threshold the frame stays as interpolated
                for (int k = 0; k < 3; ++k) {
                    tri.tangents[k] = Vec3{0.8f, 0.f, 0.6f};
                    tangents[3 * k] = 0.8f; tangents[3 * k + 1] = 0.f; tangents[3 * k + 2] = 0.6f;
                }
                CHECK(agree(d, -1.f, -1.f, false, 0) == 0.1f);
                {
                    TraceResult at(Vec3{d[0], d[1], d[2]}, 2.f, -1.f, -1.f, tri, 0, Vec2{0.25f, 0.25f}, ent, mat);
                    CHECK(at.GetTangent().x == 1.f && at.GetTangent().z == 0.f);
                }
                const float just_above[3] = {a, 0.f, -std::nextafter(0.1f, 1.f)};
                agree(just_above, -1.f, -1.f, false, 0);
            }
        }
        CHECK(found == 1);
    }
    std::printf("trace result (cpu): %d checks, %d failed; %d grazing hits\n", g_run, g_fail, grazing);
    return g_fail ? 1 : 0;
}
