/*
 * vt_oracle.h -- CPU ORACLE for the AccelStruct::Traverse hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and bench.py's cpu_baseline leg may load it.
 * The shipped library (vistrace_amd/lib/libvistrace_hip.so) never links,
 * loads or calls anything in oracle/.
 *
 * It restates, in plain C (fp32, no FMA contraction, no fast-math):
 *   - the in-tree triangle set-up and ray/triangle test of the reference
 *       source/objects/Primitives.h:75-102   (ctor, ComputeNormalAndLoD)
 *       source/objects/Primitives.h:168-215  (intersect)
 *   - the single-ray BVH walk of madmann91/bvh (v1 API), which the reference
 *     calls at source/objects/AccelStruct.cpp:818 through the types declared
 *     at source/objects/AccelStruct.h:27-31.  libs/bvh is an EMPTY, UNPINNED
 *     submodule under /root/reference (.gitmodules:4-6), so that part follows
 *     the library's published algorithm as recorded in SURVEY.md section 3.2
 *     ("UPSTREAM-RECALL").
 *   - the hit-record derivations of source/objects/TraceResult.cpp:45-86,255-262 and the
 *     texture-free part of its shading frame (:89-103, :132-137, :175-186)
 *   - the two helpers that define the bounce-ray workload:
 *       vistrace.CalcRayOrigin  source/VisTrace.cpp:1495-1517
 *       hemisphere_cos          source/libraries/BSDF.cpp:69-77
 *
 * PARITY STATUS: "parity unpinned" -- the reference holds no tests, golden
 * vectors or fixtures for this path and cannot be
 * compiled here (all of libs/ is empty).  The oracle is pinned instead by
 * analytic known-answer tests and by an independent brute-force intersector
 * (vto_trace_brute) that uses only the in-tree arithmetic of Primitives.h.
 */
#ifndef VT_ORACLE_H
#define VT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTO_MISS 0xFFFFFFFFu

/* flags of a triangle as the intersector sees them */
#define VTO_TRI_CULL_BACKFACE 1u /* oneSided && !(mat.flags & nocull)   Primitives.h:174 */
#define VTO_TRI_ALPHATEST     2u /* mat.flags & alphatest (Primitives.h:196): needs vto_set_alpha */

/* bvh::Ray<float> minus the pAccel back-pointer (Primitives.h:11-33) */
typedef struct { float org[3]; float dir[3]; float tmin; float tmax; } vto_ray;

/* The fields of TriangleBackfaceCull<float> that intersect() reads
 * (Primitives.h:56-57): p0, e1 = p0-p1, e2 = p2-p0, n = cross(e1,e2). */
typedef struct { float p0[3]; float e1[3]; float e2[3]; float n[3]; uint32_t flags; } vto_tri;

/* bvh::Bvh<float>::Node (v1): bounds interleaved {minx,maxx,miny,maxy,minz,maxz},
 * primitive_count (!=0 <=> leaf), first_child_or_primitive.  32 bytes. */
typedef struct { float bounds[6]; uint32_t prim_count; uint32_t first; } vto_node;

typedef struct { uint32_t prim; float t; float u; float v; } vto_hit;

/* bvh v1 Statistics: traversal_steps (loop iterations), intersections (triangles tested) */
typedef struct { uint64_t steps; uint64_t tests; } vto_stats;

/* Derived hit attributes: TraceResult.cpp:56-85, 255-262 */
typedef struct {
    float pos[3];       /* GetPos():  w*v0 + u*v1 + v*v2                          */
    float uvw[3];       /* (u, v, 1-u-v)                      TraceResult.cpp:70   */
    float ngeo[3];      /* nNorm = n/|n|                      Primitives.h:100     */
    float wo[3];        /* -normalize(dir)       AccelStruct.cpp:826, TraceResult.cpp:56 */
    uint32_t front;     /* dot(wo, ngeo) >= 0                 TraceResult.cpp:85   */
} vto_attrs;

/* Primitives.h:75-102 */
void vto_tri_setup(const float p0[3], const float p1[3], const float p2[3],
                   uint32_t flags, vto_tri* out);

/* Primitives.h:168-215.  Returns 1 and writes t,u,v on an accepted hit. */
int vto_tri_intersect(const vto_tri* tri, const float org[3], const float dir[3],
                      float tmin, float tmax, float* t, float* u, float* v);

/* Independent ground truth: test EVERY triangle in ascending index order with
 * the same accept rule (t <= current tmax, so a later equal-t triangle
 * replaces an earlier one).  any_hit!=0 stops at the first accepted hit. */
void vto_trace_brute(const vto_tri* tris, uint32_t ntris,
                     const vto_ray* rays, uint64_t nrays, int any_hit,
                     vto_hit* hits, int nthreads);

/* For a ray, the minimum accepted t over all triangles and how many distinct
 * triangles reach exactly that t (the admissible set for a tie-broken index).
 * ids receives up to max_ids of them. Returns the count. */
uint32_t vto_min_t_set(const vto_tri* tris, uint32_t ntris, const vto_ray* ray,
                       float* tmin_hit, uint32_t* ids, uint32_t max_ids);

/* bvh v1 SingleRayTraverser<Bvh,64,FastNodeIntersector>::traverse with
 * ClosestPrimitiveIntersector (any_hit=0) or AnyPrimitiveIntersector
 * (any_hit=1); SURVEY.md section 3.2.  nodes[0] is the root; prim_indices maps
 * leaf slot -> triangle index.  Returns 1 on hit. stats may be NULL. */
int vto_traverse(const vto_node* nodes, const uint32_t* prim_indices,
                 const vto_tri* tris, const vto_ray* ray, int any_hit,
                 vto_hit* hit, vto_stats* stats);

/* Bit mask of the -DVTO_ALT_<X> recall-sensitivity switches this library was built with (vt_oracle.c's header lists
 * them); 0 for the oracle proper -- the only build tests, smoke() and bench.py use as the checker. */
uint32_t vto_alt_mask(void);

/* Stack use of the calling thread's last vto_traverse (diagnostic for stack sizing). */
void vto_last_stack_use(uint32_t* max_sp, uint32_t* pushes);

/* OpenMP loop over rays around vto_traverse: schedule(dynamic,4096).
 * per_ray_stats (2*nrays uint32: steps,tests) and total may be NULL.
 * nthreads <= 0 -> omp default. Returns the thread count used. */
int vto_traverse_batch(const vto_node* nodes, const uint32_t* prim_indices,
                       const vto_tri* tris, const vto_ray* rays, uint64_t nrays,
                       int any_hit, vto_hit* hits, uint32_t* per_ray_stats,
                       vto_stats* total, int nthreads);

/* NUMA-aware variant for the timed CPU baseline (bench.py): one replica of the tree per NUMA node, first-touched
 * there; each worker walks the replica of the node it runs on.  Results identical to vto_traverse_batch. */
typedef struct vto_batch_ctx vto_batch_ctx;
vto_batch_ctx* vto_batch_ctx_create(const vto_node* nodes, uint64_t nnodes, const uint32_t* prim_indices, uint64_t nprims,
                                    const vto_tri* tris, uint64_t ntris, int nthreads);
int  vto_batch_ctx_replicas(const vto_batch_ctx* ctx);
void vto_batch_ctx_destroy(vto_batch_ctx* ctx);
int  vto_traverse_batch_ctx(const vto_batch_ctx* ctx, const vto_ray* rays, uint64_t nrays, int any_hit, vto_hit* hits,
                            uint32_t* per_ray_stats, vto_stats* total, int nthreads);

/* TraceResult.cpp:45-86 + GetPos :255-262 for hit {prim,u,v} of ray dir */
void vto_hit_attrs(const vto_tri* tri, const float dir[3], float u, float v, vto_attrs* out);

/* texUV and blendFactor of a hit: TraceResult.cpp:70,73-74 (uvs[3][2], alphas[3] of the triangle) */
void vto_hit_shade(float u, float v, const float uvs[6], const float alphas[3], float tex_uv[2], float* blend);

/* vistrace.CalcRayOrigin, VisTrace.cpp:1495-1517 */
void vto_calc_ray_origin(const float pos[3], const float normal[3], float out[3]);

/* hemisphere_cos, BSDF.cpp:69-77 (r1, r2 are the two sampler floats, in call order) */
void vto_hemisphere_cos(float r1, float r2, float out[3]);

/* ---- alpha test inside intersect(), source/objects/Primitives.h:196-208 ---------------------
 * In tree: texUV = (1-u-v)*uvs[0] + u*uvs[1] + v*uvs[2] (:198), TransformTexcoord (source/Utils.h:65-72),
 * the comparison alpha < mat.alphatestreference (:205, default 0.5: Material.h:122).
 * NOT in tree: mat.baseTexture->Sample(u, v, 0.f).a -- IVTFTexture::Sample forwards to the VTFParser
 * submodule (source/objects/VTFTexture.cpp:68-72), absent and unpinned.  The sampler is therefore DEFINED
 * here (and identically on the device): alpha plane of mip 0, 8 bits, repeat addressing, alpha = a / 255;
 * filter 0 = nearest texel floor(s*W), filter 1 = bilinear with texel centres at (i + 0.5) / W. */
typedef struct {
    float tex_mat[2][4];     /* Material::baseTexMat: tex_mat[r] = transform[r] of TransformTexcoord    */
    float tex_scale;         /* Material::texScale                                                       */
    float alpha_ref;         /* Material::alphatestreference                                             */
    uint32_t width, height;  /* alpha plane size (0 x 0: no texture -> alpha 1, the hit is kept)          */
    uint32_t filter;         /* 0 nearest, 1 bilinear                                                    */
    uint32_t pad;
    uint64_t offset;         /* first texel of this material in the texel array, row-major               */
} vto_alpha_material;

typedef struct {
    const vto_tri* tris_base;          /* the triangle array the walk indexes (original order)           */
    const float* tri_uv;               /* 6 floats per triangle: uvs[0..2]  (Primitives.h:65)            */
    const uint32_t* tri_material;      /* material index per triangle                                     */
    const vto_alpha_material* mats; uint32_t nmats;
    const uint8_t* texels;
} vto_alpha_ctx;

/* Side data for triangles flagged VTO_TRI_ALPHATEST (NULL = none: such a flag is then ignored).  Process-wide;
 * set it before a batch, not during one. */
void vto_set_alpha(const vto_alpha_ctx* ctx);
/* The alpha plane lookup defined above. */
float vto_alpha_sample(const vto_alpha_material* m, const uint8_t* texels, float s, float t);
/* 1 = the hit (u, v) on triangle `prim` survives the alpha test of Primitives.h:196-208. */
int vto_alpha_pass(const vto_alpha_ctx* ctx, uint32_t prim, float u, float v);

/* ---- skinning on Rebuild, source/objects/AccelStruct.cpp:34-102 ----------------
 * One vertex's Triangle::weights / boneIds / numBones (Primitives.h:68-70). 16 B. */
typedef struct vto_skin_vertex { float weight[3]; int8_t bone[3]; uint8_t num_bones; } vto_skin_vertex;

/* out[i] = bones[i] * binds[i] (the product TransformToBone forms per vertex and bone, :44);
 * glm::mat4 layout: 16 floats, column-major. */
void vto_skin_matrices(const float* bones, const float* binds, uint32_t nmat, float* out);

/* SkinTriangle :66-102 for n triangles: bind_verts n x 9 (the mesh's p0,p1,p2); the three
 * positions are re-derived through the stored p0/e1/e2 (:68-72), each is moved by
 * TransformToBone (:35-47) with the matrices mats[matrix_base[t] + bone], out_verts n x 9. */
void vto_skin_verts(const float* bind_verts, const vto_skin_vertex* skin, const uint32_t* matrix_base,
                    uint32_t n, const float* mats, float* out_verts);

/* ---- shading frame of a hit: TraceResult::CalcTBN without a normal map + CalcFootprint -------------
 * source/objects/TraceResult.cpp:58-62 (vN, vT, vB = cross(vT, vN)), :89-103 (CalcFootprint), :132-137 and
 * :175-186 (CalcTBN: interpolate, normalise, grazing-angle correction); the triangle's lod is
 * Primitives.h:97-104 (0.5 * log2(triUVArea / length(n))).  glm's normalize / dot / cross / lerp / saturate are
 * restated in their generic scalar forms (glm is un-vendored and unpinned):
 *   normalize(v) = v * (1 / sqrt(dot(v, v)));  dot = (x + y) + z;  lerp(x, y, a) = x * (1 - a) + y * a;
 *   saturate(x) = min(max(x, 0), 1).
 * normals / tangents: 3 x 3 floats (vertex-major), uvs: 3 x 2.  distance = the hit's t.  lod_set = 0 (and
 * lod_info = 0) when mipOverride holds (coneWidth < 0 || coneAngle <= 0, TraceResult.cpp:54). */
typedef struct vto_tbn { float normal[3]; float tangent[3]; float binormal[3]; float lod_info[2]; uint32_t lod_set; } vto_tbn;
void vto_hit_tbn(const vto_tri* tri, const float dir[3], float distance, float u, float v,
                 const float normals[9], const float tangents[9], const float uvs[6],
                 float cone_width, float cone_angle, vto_tbn* out);

/* SkinTriangle's normals / tangents (AccelStruct.cpp:82-92): TransformToBone with angleOnly = true, i.e. the
 * vertex is (vec, 0).  frames: n x 18 floats (normals[3][3], tangents[3][3]); not re-normalised (the reference
 * does not either). */
void vto_skin_frames(const float* bind_frames, const vto_skin_vertex* skin, const uint32_t* matrix_base,
                     uint32_t n, const float* mats, float* out_frames);

#ifdef __cplusplus
}
#endif
#endif
