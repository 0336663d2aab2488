// shading.hip -- the texture-free part of a TraceResult's shading frame, in bulk on the device: per-vertex normals / tangents
// as a scene side table (vt_scene_set_tri_frames), their skinning beside the positions (SkinTriangle,
// source/objects/AccelStruct.cpp:82-92) and TraceResult::CalcTBN without a normal map + CalcFootprint per hit
// (source/objects/TraceResult.cpp:58-62, 89-103, 132-137, 175-186).  Plain streaming kernels: one thread per hit / per
// triangle, HBM-bound gathers of 72-B frames; nothing here is on the traversal path.
//
// Arithmetic order follows the reference's glm expressions in their scalar forms (the test oracle restates the same):
// normalize(v) = v * (1 / sqrt(dot(v, v))), dot = (x + y) + z, lerp(x, y, a) = x * (1 - a) + y * a.  The library is built
// with -ffp-contract=off and correctly rounded sqrt / divide, so everything but log2 (the triangle's lod) rounds as on the host.
#include "engine_internal.h"

namespace vt {

struct HitTbnArgs {
    const vt_tri64* tris; const uint32_t* prim_to_slot; const vt_tri_frame* frames; const vt_tri_attribs* attribs;
    const vt_ray* rays; const vt_hit* hits; vt_hit_tbn* out; uint64_t n; float cone_width, cone_angle;
};
struct SkinFramesArgs {
    const vt_tri_frame* bind; const vt_skin_vertex* skin; const uint32_t* matrix_base; const float* mats; vt_tri_frame* out;
    uint32_t n, nmat;
};

__device__ __forceinline__ float dot_lr(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void cross_of(const float a[3], const float b[3], float c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ void normalise(float v[3])
{
    const float inv = 1.0f / sqrtf(dot_lr(v, v));
    v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv;
}

__global__ __launch_bounds__(kBlockThreads) void hit_tbn_kernel(HitTbnArgs a)
{
    const uint64_t i = uint64_t(blockIdx.x) * kBlockThreads + threadIdx.x;
    if (i >= a.n) return;
    const vt_hit h = a.hits[i];
    vt_hit_tbn o{};
    if (h.prim != VT_MISS) {
        const vt_tri64 T = a.tris[a.prim_to_slot[h.prim]];
        const vt_tri_frame F = a.frames[h.prim];
        const vt_ray r = a.rays[i];
        // wo, geometricNormal as the TraceResult ctor forms them (AccelStruct.cpp:826, TraceResult.cpp:56,71)
        float wo[3], ngeo[3];
        const float inv = 1.0f / sqrtf(dot_lr(r.dir, r.dir));
        const float len = sqrtf(dot_lr(T.n, T.n));
        for (int k = 0; k < 3; ++k) { wo[k] = -(r.dir[k] * inv); ngeo[k] = T.n[k] / len; }
        const float u = h.u, v = h.v, w = 1.0f - h.u - h.v;                       // :70
        float vB[3][3];
        for (int k = 0; k < 3; ++k) cross_of(F.tangent[k], F.normal[k], vB[k]);     // :60
        float n[3], t[3], b[3];
        for (int k = 0; k < 3; ++k) {                                               // :134-136
            n[k] = (w * F.normal[0][k] + u * F.normal[1][k]) + v * F.normal[2][k];
            t[k] = (w * F.tangent[0][k] + u * F.tangent[1][k]) + v * F.tangent[2][k];
            b[k] = (w * vB[0][k] + u * vB[1][k]) + v * vB[2][k];
        }
        normalise(n); normalise(t); normalise(b);
        const float kCosThetaThreshold = 0.1f;                                      // :175
        const float cos_theta = fabsf(dot_lr(wo, n));
        if (VT_MUT(43, cos_theta < kCosThetaThreshold, cos_theta <= kCosThetaThreshold)) {     // (VT_MUT: mutation sites, vt_internal.h)
            float s = cos_theta * (1.f / kCosThetaThreshold);                       // :178
            s = s < 0.f ? 0.f : s; s = s > 1.f ? 1.f : s;
            for (int k = 0; k < 3; ++k) n[k] = ngeo[k] * (1.f - s) + n[k] * s;      // :179
            normalise(n);
            const float tn = dot_lr(t, n);                                          // :181
            for (int k = 0; k < 3; ++k) t[k] = t[k] - n[k] * tn;
            normalise(t);
            VT_MUT(45, cross_of(n, t, b), cross_of(t, n, b));                        // :182
        }
        for (int k = 0; k < 3; ++k) { o.normal[k] = n[k]; o.tangent[k] = t[k]; o.binormal[k] = b[k]; }
        if (!(a.cone_width < 0.f || VT_MUT(44, a.cone_angle < 0.f, a.cone_angle <= 0.f))) {   // :54 mipOverride, :91
            const vt_tri_attribs A = a.attribs[h.prim];
            const float cw = a.cone_angle * h.t + a.cone_width;                     // :95
            const float normal_term = dot_lr(wo, ngeo);                             // :97
            const float uv10x = A.uv[1][0] - A.uv[0][0], uv10y = A.uv[1][1] - A.uv[0][1];   // Primitives.h:97-103
            const float uv20x = A.uv[2][0] - A.uv[0][0], uv20y = A.uv[2][1] - A.uv[0][1];
            const float area = fabsf(uv10x * uv20y - uv20x * uv10y);
            o.lod_info[0] = 0.5f * log2f(area / len);
            o.lod_info[1] = (cw * cw) / (normal_term * normal_term);                // :99-102
            o.lod_set = 1;
        }
    }
    a.out[i] = o;
}

// SkinTriangle's normals and tangents (AccelStruct.cpp:82-92): TransformToBone with angleOnly = true -- the vertex is (vec, 0);
// the matrices are the per-frame products skin_matrices_kernel has just formed.  One thread per VERTEX (its normal and its
// tangent share bones and weights), matrix columns fetched as four 16-B loads per bone: the kernel is bound by the vector L1's
// gather rate on the 128-KB matrix table, not by the 144 MB of frames it streams (one thread per vector with scalar matrix
// loads: 97 us for 1 M triangles).
__global__ __launch_bounds__(kBlockThreads) void skin_frames_kernel(SkinFramesArgs a)
{
    const uint64_t j = uint64_t(blockIdx.x) * kBlockThreads + threadIdx.x;
    if (j >= uint64_t(a.n) * 3) return;
    const uint32_t tri = uint32_t(j / 3), vi = uint32_t(j % 3);
    const float* nsrc = reinterpret_cast<const float*>(a.bind) + size_t(tri) * 18 + vi * 3;
    const float n0 = nsrc[0], n1 = nsrc[1], n2 = nsrc[2];
    const float t0 = nsrc[9], t1 = nsrc[10], t2 = nsrc[11];
    const vt_skin_vertex sv = a.skin[j];
    const uint32_t base = a.matrix_base[tri];
    float fn[3] = {0.f, 0.f, 0.f}, ft[3] = {0.f, 0.f, 0.f};
    for (uint32_t q = 0; q < sv.num_bones && q < 3u; ++q) {
        uint32_t mi = base + uint32_t(int(sv.bone[q]));
        mi = mi < a.nmat ? mi : 0u;                            // as skin_tris_kernel: stay inside the table
        const float4* M = reinterpret_cast<const float4*>(a.mats + size_t(mi) * 16);
        const float4 c0 = M[0], c1 = M[1], c2 = M[2], c3 = M[3];
        const float m0[3] = {c0.x, c0.y, c0.z}, m1[3] = {c1.x, c1.y, c1.z}, m2[3] = {c2.x, c2.y, c2.z}, m3[3] = {c3.x, c3.y, c3.z};
        for (int r = 0; r < 3; ++r) {
            const float a0 = m0[r] * n0 + m1[r] * n1;
            const float a1 = m2[r] * n2 + m3[r] * 0.f;
            fn[r] = fn[r] + (a0 + a1) * sv.weight[q];
            const float b0 = m0[r] * t0 + m1[r] * t1;
            const float b1 = m2[r] * t2 + m3[r] * 0.f;
            ft[r] = ft[r] + (b0 + b1) * sv.weight[q];
        }
    }
    float* dst = reinterpret_cast<float*>(a.out) + size_t(tri) * 18 + vi * 3;
    dst[0] = fn[0]; dst[1] = fn[1]; dst[2] = fn[2];
    dst[9] = ft[0]; dst[10] = ft[1]; dst[11] = ft[2];
}

hipError_t launch_hit_tbn(vt_scene* s, const void* d_rays, const void* d_hits, uint64_t n, float cone_width, float cone_angle,
                          void* d_out, hipStream_t stream)
{
    HitTbnArgs a{s->d_tris, s->d_prim_to_slot, s->d_frames, s->d_attribs, static_cast<const vt_ray*>(d_rays),
                 static_cast<const vt_hit*>(d_hits), static_cast<vt_hit_tbn*>(d_out), n, cone_width, cone_angle};
    const uint64_t blocks = (n + kBlockThreads - 1) / kBlockThreads;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hit_tbn_kernel, dim3(uint32_t(blocks)), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

// called by vt_scene_skin_refit behind skin_matrices_kernel (same stream): the frames follow the bones
hipError_t skin_frames(vt_scene* s, const float* d_prod, uint32_t nmat, hipStream_t stream)
{
    if (!s->d_frames_bind) return hipSuccess;
    SkinFramesArgs a{s->d_frames_bind, s->d_skin, s->d_matrix_base, d_prod, s->d_frames, s->ntris, nmat};
    const uint64_t vertices = uint64_t(s->ntris) * 3;
    hipLaunchKernelGGL(skin_frames_kernel, dim3(uint32_t((vertices + kBlockThreads - 1) / kBlockThreads)), dim3(kBlockThreads), 0, stream, a);
    return hipGetLastError();
}

} // namespace vt

using namespace vt;

extern "C" {

int vt_scene_set_tri_frames(vt_scene* s, const vt_tri_frame* frames, uint32_t n)
{
    if (s) for (vt_scene* rep : s->replicas) { const int rc = vt_scene_set_tri_frames(rep, frames, n); if (rc != VT_OK) return rc; }
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_frames: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_frames: the scene\'s engine has been closed");
    if (n != s->ntris) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_frames: n differs from the scene's triangle count");
    if (n != 0 && !frames) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_frames: frames is NULL");
    DeviceGuard guard(s->engine->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_set_tri_frames: hipSetDevice failed");
    if (n == 0) return VT_OK;
    const size_t bytes = size_t(n) * sizeof(vt_tri_frame);
    std::lock_guard<std::mutex> host_lock(s->engine->host_mu);   // as the other calls that rewrite a scene's tables
    if (!s->d_frames_bind) {
        // bind-pose and current frames in one block: the current ones are rewritten by every vt_scene_skin_refit
        VT_HIP(dev_malloc(reinterpret_cast<void**>(&s->d_frames_bind), 2 * bytes));
        s->d_frames = s->d_frames_bind + n;
        s->bytes += 2 * bytes;
    }
    VT_HIP(hipDeviceSynchronize());                  // vt_hit_tbn_dev launches in flight read the old table
    VT_HIP(hipMemcpy(s->d_frames_bind, frames, bytes, hipMemcpyHostToDevice));
    VT_HIP(hipMemcpy(s->d_frames, s->d_frames_bind, bytes, hipMemcpyDeviceToDevice));
    return VT_OK;
}

int vt_scene_read_tri_frames(vt_scene* s, vt_tri_frame* frames_out)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_scene_read_tri_frames: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_read_tri_frames: the scene\'s engine has been closed");
    if (s->ntris == 0) return VT_OK;
    if (!frames_out) return fail(VT_ERR_INVALID_ARG, "vt_scene_read_tri_frames: frames_out is NULL");
    if (!s->d_frames) return fail(VT_ERR_INVALID_ARG, "vt_scene_read_tri_frames: call vt_scene_set_tri_frames first");
    DeviceGuard guard(s->engine->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_read_tri_frames: hipSetDevice failed");
    VT_HIP(hipStreamSynchronize(s->engine->stream));
    VT_HIP(hipMemcpy(frames_out, s->d_frames, size_t(s->ntris) * sizeof(vt_tri_frame), hipMemcpyDeviceToHost));
    return VT_OK;
}

int vt_hit_tbn_dev(vt_scene* s, const void* d_rays, const void* d_hits, uint64_t n, float cone_width, float cone_angle,
                   void* d_out, void* stream)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_hit_tbn_dev: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_hit_tbn_dev: the scene\'s engine has been closed");
    if (n == 0) return VT_OK;
    if (!d_rays || !d_hits || !d_out) return fail(VT_ERR_INVALID_ARG, "vt_hit_tbn_dev: NULL device buffer");
    if (!s->d_frames) return fail(VT_ERR_INVALID_ARG, "vt_hit_tbn_dev: call vt_scene_set_tri_frames first");
    const bool cone_on = !(cone_width < 0.f || cone_angle <= 0.f);
    if (cone_on && !s->d_attribs) return fail(VT_ERR_INVALID_ARG, "vt_hit_tbn_dev: call vt_scene_set_tri_attribs first (with a cone the triangle's lod is asked for, which is derived from its uvs)");
    DeviceGuard guard(s->engine->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_hit_tbn_dev: hipSetDevice failed");
    VT_HIP(launch_hit_tbn(s, d_rays, d_hits, n, cone_width, cone_angle, d_out, static_cast<hipStream_t>(stream)));
    return VT_OK;
}

} // extern "C"
