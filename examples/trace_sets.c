/* trace_sets.c -- a frame's worth of small ray sets through ONE launch, from plain C (include/vistrace_hip.h):
 * three host buffers (a 64 x 64 camera image, a handful of probe rays, an empty set) are staged and uploaded one by one
 * (vt_batch_set_add: the range checks of AccelStruct::Traverse run inside the staging copy), traced by one merged launch
 * (vt_batch_set_trace) and read back as one vt_batch each.  The same sets are then traced one call at a time: same bytes.
 *   gcc -std=c11 -Iinclude examples/trace_sets.c -Lvistrace_amd/lib -lvistrace_hip -Wl,-rpath,$PWD/vistrace_amd/lib
 * Exit code 0 = everything agreed; 2 = no HIP device (the library has no CPU fallback). */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
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

enum { SIDE = 64, NCAM = SIDE * SIDE, NPROBE = 5 };

int main(void)
{
    /* a 8 x 8 floor of quads at z = 0 under a camera at z = 10 looking down */
    enum { K = 8, NTRI = K * K * 2 };
    float verts[NTRI * 9];
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) {
            const float x0 = (float)i - 4, y0 = (float)j - 4, x1 = x0 + 1, y1 = y0 + 1;
            const float q[18] = {x0, y0, 0, x1, y0, 0, x0, y1, 0, /**/ x1, y0, 0, x1, y1, 0, x0, y1, 0};
            memcpy(verts + (size_t)(i * K + j) * 18, q, sizeof q);
        }
    static vt_tri64 recs[NTRI];
    vt_bvh* bvh = NULL;
    vt_host_scene* hs = NULL;
    vt_engine* eng = NULL;
    vt_scene* scene = NULL;
    int ndev = 0;
    CHECK(vt_device_count(&ndev));
    if (ndev <= 0) { fprintf(stderr, "no HIP device (the library has no CPU fallback)\n"); return 2; }
    CHECK(vt_tris_setup(verts, NULL, NTRI, recs));
    CHECK(vt_bvh_build(recs, NTRI, 0, &bvh));
    CHECK(vt_scene_linearise(bvh, recs, &hs));
    CHECK(vt_engine_open(0, &eng));
    CHECK(vt_scene_upload(eng, hs, &scene));
    vt_bvh_free(bvh);

    static vt_ray cam[NCAM];
    for (int y = 0; y < SIDE; ++y)
        for (int x = 0; x < SIDE; ++x) {
            const vt_ray r = {{0.f, 0.f, 10.f}, {((float)x + 0.5f) / SIDE - 0.5f, ((float)y + 0.5f) / SIDE - 0.5f, -1.f}, 0.f, FLT_MAX};
            cam[y * SIDE + x] = r;
        }
    const vt_ray probes[NPROBE] = {
        {{0.25f, 0.25f, 3.f}, {0, 0, -1}, 0.f, FLT_MAX}, {{-3.5f, 2.5f, 1.f}, {0, 0, -1}, 0.f, FLT_MAX}, {{0.f, 0.f, 1.f}, {0, 0, 1}, 0.f, FLT_MAX},
        {{9.f, 9.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}, {{1.5f, -1.5f, 2.f}, {0, 0, -1}, 0.f, 1.f}};
    const vt_ray* sets[3] = {cam, probes, NULL};
    const uint64_t counts[3] = {NCAM, NPROBE, 0};
    const uint32_t widths[3] = {SIDE, 0, 0};                 /* only the camera image is in image order */

    vt_batch_set* set = NULL;
    vt_batch* out[3] = {NULL, NULL, NULL};
    CHECK(vt_batch_set_begin(scene, VT_BATCH_CHECK_RANGES | VT_BATCH_FETCH_HITS, &set));
    for (int k = 0; k < 3; ++k) {
        uint64_t bad = 0;
        CHECK(vt_batch_set_add(set, sets[k], counts[k], widths[k], &bad));
    }
    CHECK(vt_batch_set_trace(set, out));                     /* ONE launch for all three; consumes the set */

    int ok = 1;
    uint64_t hit_count = 0;
    for (int k = 0; k < 3; ++k) {
        const vt_hit* merged = NULL;
        CHECK(vt_batch_hits(out[k], &merged));
        ok = ok && vt_batch_count(out[k]) == counts[k];
        if (counts[k] == 0) continue;
        vt_hit* single = (vt_hit*)malloc(counts[k] * sizeof(vt_hit));
        CHECK(vt_trace_closest(scene, sets[k], counts[k], single));
        ok = ok && memcmp(single, merged, counts[k] * sizeof(vt_hit)) == 0;
        for (uint64_t i = 0; i < counts[k]; ++i) hit_count += merged[i].prim != VT_MISS;
        free(single);
    }
    const vt_hit* ph = NULL;
    CHECK(vt_batch_hits(out[1], &ph));
    ok = ok && ph[0].prim != VT_MISS && ph[0].t == 3.f && ph[2].prim == VT_MISS && ph[3].prim == VT_MISS && ph[4].prim == VT_MISS;
    /* a ray with tMax <= tMin is refused by the add, with its index */
    vt_ray bad_rays[2] = {probes[0], probes[1]};
    bad_rays[1].tmax = bad_rays[1].tmin;
    uint64_t bad = 99;
    CHECK(vt_batch_set_begin(scene, VT_BATCH_CHECK_RANGES, &set));
    ok = ok && vt_batch_set_add(set, bad_rays, 2, 0, &bad) == VT_ERR_INVALID_ARG && bad == 1 && vt_batch_set_count(set) == 0;
    vt_batch_set_abort(set);

    printf("%llu of %d rays hit; merged launch %s the separate calls\n", (unsigned long long)hit_count, NCAM + NPROBE, ok ? "equals" : "DIFFERS FROM");
    for (int k = 0; k < 3; ++k) vt_batch_free(out[k]);
    vt_host_scene_free(hs);
    vt_scene_free(scene);
    vt_engine_close(eng);
    printf(ok ? "ok\n" : "MISMATCH\n");
    return ok ? 0 : 1;
}
