/* trace_batch.c -- the C ABI of include/vistrace_hip.h used from plain C (no C++, no Python):
 * set up a two-triangle scene, build + linearise on the CPU, open ALL devices as one group (vt_engine_open_multi: the
 * scene is replicated, big host batches are sharded), upload, trace a small batch, read the hits; then the single-ray
 * latency path (host walk) on the same records.
 *   gcc -std=c11 -Iinclude examples/trace_batch.c -Lvistrace_amd/lib -lvistrace_hip -Wl,-rpath,$PWD/vistrace_amd/lib
 * Exit code 0 = the expected hits came back; 2 = no HIP device (the library has no CPU fallback). */
#include <float.h>
#include <stdio.h>
#include <stdlib.h>

#include "vistrace_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc__ = (call);                                                       \
        if (rc__ != VT_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc__, vt_last_error()); \
            return rc__ == VT_ERR_HIP ? 2 : 1;                                   \
        }                                                                        \
    } while (0)

int main(void)
{
    /* two triangles: a floor piece at z = 0 and a smaller one above it at z = 1 */
    const float verts[2 * 9] = {0, 0, 0, 4, 0, 0, 0, 4, 0, /**/ 0, 0, 1, 1, 0, 1, 0, 1, 1};
    vt_tri64 recs[2];
    vt_bvh* bvh = NULL;
    vt_host_scene* hs = NULL;
    vt_engine* eng = NULL;
    vt_scene* scene = NULL;
    CHECK(vt_tris_setup(verts, NULL, 2, recs));
    CHECK(vt_bvh_build(recs, 2, 0, &bvh));
    int ndev = 0, devs[64];
    CHECK(vt_device_count(&ndev));
    if (ndev > 64) ndev = 64;
    for (int i = 0; i < ndev; ++i) devs[i] = i;
    if (ndev <= 0) { fprintf(stderr, "no HIP device (the library has no CPU fallback)\n"); return 2; }
    CHECK(vt_engine_open_multi(devs, ndev, &eng));
    /* Rebuild's upload step: the tree and the records go up as they are, the device re-packs them;
     * the host copy that single rays are walked on comes back from the device */
    CHECK(vt_scene_upload_tree(eng, bvh, recs, 2, &scene));
    vt_bvh_free(bvh);
    CHECK(vt_host_scene_download(scene, &hs));

    const vt_ray rays[3] = {
        {{0.25f, 0.25f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX}, /* hits the upper triangle at t = 4 */
        {{2.f, 1.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX},     /* passes it, hits the floor at t = 5  */
        {{9.f, 9.f, 5.f}, {0, 0, -1}, 0.f, FLT_MAX},     /* misses both                          */
    };
    vt_hit hits[3];
    uint8_t occluded[3];
    CHECK(vt_trace_closest(scene, rays, 3, hits));
    CHECK(vt_trace_any(scene, rays, 3, occluded));
    for (int i = 0; i < 3; ++i)
        printf("ray %d: prim %u t %g u %g v %g occluded %u\n", i, hits[i].prim, hits[i].t, hits[i].u, hits[i].v, occluded[i]);
    const int ok = hits[0].prim == 1 && hits[0].t == 4.f && hits[0].u == 0.25f && hits[0].v == 0.25f &&
                   hits[1].prim == 0 && hits[1].t == 5.f && hits[2].prim == VT_MISS && occluded[0] == 1 &&
                   occluded[1] == 1 && occluded[2] == 0;
    /* what one accel:Traverse call does: the same walk on the host copy, bit-identical */
    vt_hit hw[3];
    CHECK(vt_host_scene_trace_closest(hs, rays, 3, hw));
    int same = 1;
    for (int i = 0; i < 3; ++i) same = same && hw[i].prim == hits[i].prim && hw[i].t == hits[i].t && hw[i].u == hits[i].u && hw[i].v == hits[i].v;
    printf("devices in the group: %d; host walk %s the device\n", vt_engine_device_count(eng), same ? "equals" : "DIFFERS FROM");
    vt_host_scene_free(hs);
    vt_scene_free(scene);
    vt_engine_close(eng);
    printf(ok && same ? "ok\n" : "MISMATCH\n");
    return ok && same ? 0 : 1;
}
