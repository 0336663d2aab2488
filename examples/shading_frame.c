/* shading_frame.c -- TraceResult:Normal / Tangent / Binormal for a whole batch, and vertex frames that follow the bones, from
 * plain C (include/vistrace_hip.h).  A two-triangle quad carries smooth per-vertex normals (vt_scene_set_tri_frames); a batch
 * traced against it hands back the interpolated, normalised frame of every hit (vt_batch_tbn: TraceResult::CalcTBN without a
 * normal map, source/objects/TraceResult.cpp:132-186).  Then the quad is skinned by one bone turned a quarter about z
 * (vt_scene_set_skin + vt_scene_skin_refit: SkinTriangle, source/objects/AccelStruct.cpp:66-102): positions AND frames turn.
 *   gcc -std=c11 -Iinclude examples/shading_frame.c -Lvistrace_amd/lib -lvistrace_hip -lm -Wl,-rpath,$PWD/vistrace_amd/lib
 * Exit code 0 = everything agreed; 2 = no HIP device (the library has no CPU fallback). */
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "vistrace_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc__ = (call);                                                       \
        if (rc__ != VT_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, vt_last_error()); \
            return rc__ == VT_ERR_HIP ? 2 : 1;                                   \
        }                                                                        \
    } while (0)

static int near3(const float a[3], float x, float y, float z) { return fabsf(a[0] - x) < 1e-5f && fabsf(a[1] - y) < 1e-5f && fabsf(a[2] - z) < 1e-5f; }

int main(void)
{
    int ndev = 0;
    CHECK(vt_device_count(&ndev));
    if (ndev <= 0) { fprintf(stderr, "no HIP device (the library has no CPU fallback)\n"); return 2; }
    /* quad (0,0)-(2,2) in z = 0, two triangles; p0, p1, p2 per triangle */
    const float verts[18] = {0, 0, 0, 2, 0, 0, 0, 2, 0, /**/ 2, 0, 0, 2, 2, 0, 0, 2, 0};
    vt_tri64 recs[2];
    vt_bvh* bvh = NULL;
    vt_host_scene* hs = NULL;
    vt_engine* eng = NULL;
    vt_scene* scene = NULL;
    CHECK(vt_tris_setup(verts, NULL, 2, recs));
    CHECK(vt_bvh_build(recs, 2, 0, &bvh));
    CHECK(vt_scene_linearise(bvh, recs, &hs));
    CHECK(vt_engine_open(0, &eng));
    CHECK(vt_scene_upload(eng, hs, &scene));
    vt_bvh_free(bvh);

    /* side tables: uvs (only the cone footprint needs them: the triangle's lod is derived from the uvs) and the vertex frames --
     * normals lean outwards along x as on a gently curved panel: (-0.6, 0, 0.8) on the x = 0 edge, (+0.6, 0, 0.8) on the
     * x = 2 edge; tangents perpendicular to them */
    vt_tri_attribs attribs[2];
    vt_tri_frame frames[2];
    memset(attribs, 0, sizeof attribs);
    for (int t = 0; t < 2; ++t)
        for (int k = 0; k < 3; ++k) {
            const float x = verts[t * 9 + k * 3], y = verts[t * 9 + k * 3 + 1];
            const float lean = x < 1.f ? -0.6f : 0.6f;
            attribs[t].uv[k][0] = x * 0.5f; attribs[t].uv[k][1] = y * 0.5f; attribs[t].alpha[k] = 1.f;
            frames[t].normal[k][0] = lean; frames[t].normal[k][1] = 0.f; frames[t].normal[k][2] = 0.8f;
            frames[t].tangent[k][0] = 0.8f; frames[t].tangent[k][1] = 0.f; frames[t].tangent[k][2] = -lean;
        }
    CHECK(vt_scene_set_tri_attribs(scene, attribs, 2));
    CHECK(vt_scene_set_tri_frames(scene, frames, 2));

    /* three rays straight down: over the left edge, the middle and the right edge of the panel; one that misses */
    const vt_ray rays[4] = {{{0.01f, 1.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}, {{1.f, 0.5f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX},
                            {{1.99f, 1.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}, {{9.f, 9.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}};
    vt_batch* batch = NULL;
    const vt_hit* hits = NULL;
    const vt_hit_tbn* tbn = NULL;
    CHECK(vt_batch_trace_closest(scene, rays, 4, &batch));
    CHECK(vt_batch_hits(batch, &hits));
    CHECK(vt_batch_tbn(batch, &tbn));
    int ok = hits[0].prim != VT_MISS && hits[1].prim != VT_MISS && hits[2].prim != VT_MISS && hits[3].prim == VT_MISS;
    ok = ok && tbn[0].normal[0] < -0.55f && tbn[2].normal[0] > 0.55f;               /* the frame leans with the panel ...        */
    ok = ok && near3(tbn[1].normal, 0.f, 0.f, 1.f) && near3(tbn[1].tangent, 1.f, 0.f, 0.f) && near3(tbn[1].binormal, 0.f, -1.f, 0.f);
    ok = ok && tbn[3].normal[2] == 0.f && tbn[0].lod_set == 0;                      /* ... a miss reads zeros; cone off by default */
    printf("middle of the panel: normal (%.3f %.3f %.3f) tangent (%.3f %.3f %.3f) binormal (%.3f %.3f %.3f)\n", tbn[1].normal[0],
           tbn[1].normal[1], tbn[1].normal[2], tbn[1].tangent[0], tbn[1].tangent[1], tbn[1].tangent[2], tbn[1].binormal[0], tbn[1].binormal[1], tbn[1].binormal[2]);
    vt_batch_free(batch);

    /* one bone: a quarter turn about z (x -> y, y -> -x), bind = identity; glm::mat4, column-major */
    vt_skin_vertex skin[6];
    uint32_t matrix_base[2] = {0, 0};
    for (int k = 0; k < 6; ++k) { memset(&skin[k], 0, sizeof skin[k]); skin[k].weight[0] = 1.f; skin[k].num_bones = 1; }
    const float bone[16] = {0, 1, 0, 0, /**/ -1, 0, 0, 0, /**/ 0, 0, 1, 0, /**/ 0, 0, 0, 1};
    const float bind[16] = {1, 0, 0, 0, /**/ 0, 1, 0, 0, /**/ 0, 0, 1, 0, /**/ 0, 0, 0, 1};
    CHECK(vt_scene_set_skin(scene, verts, skin, matrix_base, 2));
    CHECK(vt_scene_skin_refit(scene, bone, bind, 1));
    /* the panel now spans x in [-2, 0], y in [0, 2]; its old x = 2 edge (normals leaning +x) lies along y = 2 and leans +y */
    const vt_ray after[2] = {{{-1.f, 1.99f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}, {{1.f, 1.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}};
    CHECK(vt_batch_trace_closest(scene, after, 2, &batch));
    CHECK(vt_batch_hits(batch, &hits));
    CHECK(vt_batch_tbn(batch, &tbn));
    ok = ok && hits[0].prim != VT_MISS && hits[1].prim == VT_MISS;                  /* the geometry moved ...                    */
    ok = ok && tbn[0].normal[1] > 0.55f && fabsf(tbn[0].normal[0]) < 1e-5f;          /* ... and the frames turned with it         */
    printf("after the quarter turn, near the y = 2 edge: normal (%.3f %.3f %.3f)\n", tbn[0].normal[0], tbn[0].normal[1], tbn[0].normal[2]);
    vt_batch_free(batch);

    vt_host_scene_free(hs);
    vt_scene_free(scene);
    vt_engine_close(eng);
    printf("%s\n", ok ? "ok" : "MISMATCH");
    return ok ? 0 : 1;
}
