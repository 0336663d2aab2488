/*
 * vistrace_hip.h -- C ABI of the MI355X (gfx950) ray-tracing core for VisTrace.
 *
 * This is the drop-in boundary for ONE path of Derpius/VisTrace:
 * AccelStruct::Traverse and the BVH hand-over that feeds it.  The reference has
 * no FFI for this path (it calls madmann91/bvh in-process), so every entry point
 * below names the reference code it replaces (paths relative to the reference
 * tree).  The host C++ class vistrace::AccelStruct (vistrace_amd/csrc/host/) sits
 * above this header and keeps the reference's GLua surface; INTEGRATION.md shows
 * the three-line patch a VisTrace maintainer would apply.
 *
 * Conventions: plain pointers and sizes, POD structs, `int` status (0 = ok),
 * no exceptions or longjmp across the boundary, caller owns every host buffer.
 * All calls are synchronous unless the name ends in `_dev` (device pointers,
 * enqueued on the given HIP stream, no host sync).  There is NO CPU fallback:
 * if no HIP device is usable the call fails with VT_ERR_HIP.
 *
 * Threading and streams: the reference serves one caller (GMod Lua is single-threaded);
 * this library is safe beyond that.  Every trace launch takes its own scratch (ray cursor, reserved-CU
 * counters, stack overflow area) from a ring of 16 launch slots, so `_dev` launches of one engine may be
 * in flight on any number of streams at once and host threads may share an engine (enqueueing is
 * serialised internally; the host-pointer entry points and vt_bounce_loop_dev, which use engine-wide
 * staging, run one at a time per engine; scenes and batch objects may be created and freed from any host
 * thread).  Calls that rewrite a scene in place (vt_scene_refit,
 * vt_scene_skin_refit, vt_scene_set_alpha, vt_scene_free) first wait for every launch in flight on the
 * device.  vt_engine_set_timing / vt_engine_last_kernel_ms describe the last launch only.
 *
 * Errors: a launch that cannot be enqueued fails its call and leaves the engine usable (the launch slot's ray cursors are cleared
 * before its next use).  After a device-side fault (an illegal address, a hung queue: VT_ERR_HIP from a synchronising call) the
 * HIP context is unusable: close the engine and open a new one -- persistent launches count on the last wave of the previous
 * launch on a slot having put the slot's cursors back, which a launch that died did not do.
 */
#ifndef VISTRACE_HIP_H
#define VISTRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_ABI_VERSION 5   /* 2: launch-slot ring, host walk, multi-GPU entries; 3: vt_batch, any-hit counters, gather timing; 4: merged launches, chunked gather, batch sets, vertex frames + vt_hit_tbn; 5: vt_engine_member, vt_scene_upload_stats (all additive) */

enum vt_status {
    VT_OK              = 0,
    VT_ERR_INVALID_ARG = 1,
    VT_ERR_HIP         = 2, /* HIP runtime/driver error or no device   */
    VT_ERR_UNSUPPORTED = 3, /* e.g. tracing alpha-tested triangles before vt_scene_set_alpha */
    VT_ERR_NOMEM       = 4,
    VT_ERR_STACK       = 5  /* tree deeper than the traversal stack    */
};

#define VT_MISS 0xFFFFFFFFu

/* triangle flags, resolved from Material::flags at upload time */
#define VT_TRI_CULL_BACKFACE 1u /* oneSided && !(mat.flags & nocull): source/objects/Primitives.h:174 */
#define VT_TRI_ALPHATEST     2u /* mat.flags & alphatest: source/objects/Primitives.h:196 (needs vt_scene_set_alpha) */

/* bvh::Ray<float> without the pAccel pointer: source/objects/Primitives.h:11-33.  32 B. */
typedef struct vt_ray { float org[3]; float dir[3]; float tmin; float tmax; } vt_ray;

/* What mpTraverser->traverse() returns (source/objects/AccelStruct.cpp:818-820):
 * primitive_index + Intersection{t,u,v} (source/objects/Primitives.h:47-51).
 * prim == VT_MISS <=> std::nullopt.  16 B. */
typedef struct vt_hit { uint32_t prim; float t; float u; float v; } vt_hit;

/* bvh::Bvh<float>::Node of madmann91/bvh v1 (type alias source/objects/AccelStruct.h:27):
 * bounds = {minx,maxx,miny,maxy,minz,maxz}; prim_count != 0 <=> leaf; `first` is
 * the first child (inner) or the first slot in prim_indices (leaf).  32 B.
 * nodes[0] is the root, the two children of an inner node are adjacent. */
typedef struct vt_bvh_node { float bounds[6]; uint32_t prim_count; uint32_t first; } vt_bvh_node;

/* Device node record: one sibling pair, 64 B, 64-B aligned.  For an inner child
 * `first` is the index of ITS pair; for a leaf child it is the first record in
 * the leaf-ordered vt_tri64 array. */
typedef struct vt_node_pair { vt_bvh_node child[2]; } vt_node_pair;

/* Device triangle record, 64 B: the fields TriangleBackfaceCull::intersect reads
 * (source/objects/Primitives.h:56,168-215) + the original triangle index the
 * reference reports as hit->primitive_index. */
typedef struct vt_tri64 {
    float p0[3]; float e1[3]; float e2[3]; float n[3];
    uint32_t prim; uint32_t flags; uint32_t pad[2];
} vt_tri64;

/* Per-ray traversal counters = bvh v1 Statistics{traversal_steps, intersections}. */
typedef struct vt_ray_stats { uint32_t steps; uint32_t tests; } vt_ray_stats;

/* Batched TraceResult core (source/objects/TraceResult.cpp:45-86, 255-262). 64 B. */
typedef struct vt_hit_attrs {
    float pos[3];   float t;        /* GetPos() :255-262 ; distance :52            */
    float ngeo[3];  uint32_t prim;  /* geometricNormal :71                          */
    float uvw[3];   uint32_t front; /* uvw :70 ; frontFacing :85                    */
    float wo[3];    uint32_t hit;   /* wo :56 ; hit = 1, or 0 for a miss (all zero) */
} vt_hit_attrs;

/* Per-triangle side table for the rest of the TraceResult constructor (original triangle
 * order): uvs/alphas of the three vertices (source/objects/Primitives.h:65-66), the owning
 * entity's id (Entity::id, source/objects/AccelStruct.h:36) and the material index. 48 B. */
typedef struct vt_tri_attribs { float uv[3][2]; float alpha[3]; uint32_t ent_id; uint32_t material; uint32_t pad; } vt_tri_attribs;

/* texUV, blendFactor, entIdx, submatIdx of a hit (source/objects/TraceResult.cpp:73-78). 32 B.
 * A miss yields zeros with ent_id = material = VT_MISS. */
typedef struct vt_hit_shade { float tex_uv[2]; float blend; uint32_t ent_id; uint32_t material; uint32_t pad[3]; } vt_hit_shade;

/* Per-vertex normals and tangents of a triangle (Triangle::normals / tangents, source/objects/Primitives.h:62-63), original
 * triangle order, as the mesh or the world loader produced them (not re-normalised here either).  72 B. */
typedef struct vt_tri_frame { float normal[3][3]; float tangent[3][3]; } vt_tri_frame;
/* What TraceResult::CalcTBN (source/objects/TraceResult.cpp:132-186) leaves in normal / tangent / binormal for a material
 * WITHOUT a normal map -- the interpolated, normalised vertex frame with vB = cross(vT, vN) per vertex (:59-61) and the
 * grazing-angle correction towards the geometric normal (:175-184) -- and TraceResult::CalcFootprint's textureLodInfo
 * (:89-103: x = the triangle's lod, Primitives.h:93-105, y = coneWidth^2 / dot(wo, geometricNormal)^2 with the cone
 * propagated to the hit: coneAngle * distance + coneWidth, distance = the hit's t in units of the direction AS GIVEN, which is what
 * the reference passes, AccelStruct.cpp:826).  lod_set = 0 and lod_info = 0 when the cone is switched off (coneWidth < 0 or coneAngle <= 0,
 * the mipOverride of TraceResult.cpp:54 -- the defaults of accel:Traverse).  The normal-map branch (:139-173) needs
 * IVTFTexture::Sample of the absent VTFParser submodule and is out of scope; a caller with a normal map perturbs the frame
 * returned here.  A miss yields zeros.  48 B. */
typedef struct vt_hit_tbn { float normal[3]; float tangent[3]; float binormal[3]; float lod_info[2]; uint32_t lod_set; } vt_hit_tbn;

/* Pinhole camera of the synthetic workloads (pixel-centre rays, row-major, top row first). */
typedef struct vt_camera { float pos[3]; float forward[3]; float up[3]; float vfov_deg; uint32_t width; uint32_t height; } vt_camera;

typedef struct vt_bvh        vt_bvh;        /* host: v1-layout tree                 */
typedef struct vt_host_scene vt_host_scene; /* host: linearised pairs + tri records */
typedef struct vt_engine     vt_engine;     /* one HIP device + stream              */
typedef struct vt_scene      vt_scene;      /* device-resident scene                */


/* ==== CORE: the twelve calls a VisTrace patch needs =============================================================================
 * INTEGRATION.md sections 1-3 use exactly these, in this order: module open, Rebuild (AccelStruct.cpp:762-775), Traverse
 * (:810-831), teardown (:525-542).  Everything behind this block is an EXTENSION: additive, and none of it is needed to replace
 * the reference's path (tests/test_abi_symbols.py checks this list against INTEGRATION.md). */

/*  1 */
/* Thread-local description of the last error on this thread ("" if none). */
const char* vt_last_error(void);

/*  2, 3: GMOD_MODULE_OPEN / GMOD_MODULE_CLOSE -- one engine = one HIP device + its stream and staging. */
int  vt_engine_open(int device, vt_engine** out);
void vt_engine_close(vt_engine* e);

/*  4 */
/* Triangle constructor + ComputeNormalAndLoD: source/objects/Primitives.h:75-102.
 * verts = n x {p0,p1,p2} (9 floats); flags may be NULL (all 0); out[i].prim = i. */
int vt_tris_setup(const float* verts, const uint8_t* flags, uint32_t n, vt_tri64* out);

/*  5, 6 */
/* BVH build tail of AccelStruct::PopulateAccel, source/objects/AccelStruct.cpp:763-770 (bounding boxes + centres,
 * Primitives.h:107-118; build; leaves of several triangles).  tris in ORIGINAL order.  n == 0 gives an empty tree (every
 * trace misses).  nthreads <= 0: min(OpenMP default, 16).  Two builders, same v1 node layout, same traversal, identical
 * t,u,v (only tie-broken indices can differ, as between any two trees):
 *   VT_BUILDER_BINNED_SAH  top-down binned SAH, task-parallel and deterministic for any thread count, every subtree task
 *                          refining its subtree by re-insertion (below) -- the DEFAULT of
 *                          vt_bvh_build: kernel time is proportional to the node steps per ray, and this tree needs 14 %
 *                          (incoherent rays) to 39 % (camera rays) fewer of them than the PLOC tree at the same Rebuild
 *                          time (1 M triangles, 8 threads: 0.27 s against 0.37 s; 10 M: 5.0 s against 6.3 s; README.md has the table by thread count);
 *   VT_BUILDER_BINNED_SAH_REFINED  the same followed by two passes of insertion-based optimisation over the WHOLE tree (Bittner
 *                          et al. 2013: the worst 1 % of the inner nodes are taken out and their subtrees re-inserted where they
 *                          enlarge the tree least).  Opt-in (VT_BUILDER=sah_refined) for scenes that are built once and traced
 *                          a lot: 1 M triangles +0.1 s of build, another 3 % fewer node steps;
 *   VT_BUILDER_PLOC        the reference's algorithm: Morton-32 sort, PLOC (search radius 14), SAH leaf collapse
 *                          (bvh v1 LocallyOrderedClusteringBuilder + LeafCollapser).  VT_BUILDER=ploc in the environment
 *                          makes vt_bvh_build use it. */
int  vt_bvh_build(const vt_tri64* tris, uint32_t n, int nthreads, vt_bvh** out);
void vt_bvh_free(vt_bvh* bvh);

/*  7, 8: replaces `new Intersector(...)` / `new Traverser(...)`, source/objects/AccelStruct.cpp:772-773, and their deletes */
/* Upload once per Rebuild, the re-packing done ON THE DEVICE: the tree as vt_bvh_build left it and the triangle records in
 * original order go up as three plain copies; kernels number the pairs depth-first, shuffle the triangles into leaf order and
 * derive the index tables.  Result byte-equal to the extension pair vt_scene_linearise + vt_scene_upload (records, level lists, triangle -> slot),
 * without the host walk (85-100 ms per million triangles on one core) -- stands where the reference constructs its intersector
 * and traverser over the finished tree (source/objects/AccelStruct.cpp:772-773).  tris = the vt_tris_setup output the tree
 * was built from, ntris = its length.  On a group's root the scene is built on every member. */
int  vt_scene_upload_tree(vt_engine* e, const vt_bvh* bvh, const vt_tri64* tris, uint32_t ntris, vt_scene** out);
void vt_scene_free(vt_scene* s);

/*  9, 10: single rays are walked on the host copy of the records (call 11) */
/* The host-side copy of a device scene (pairs, leaf-ordered triangles, depths): what vt_host_scene_trace_* and the accessors
 * need when the scene came from vt_scene_upload_tree.  From then on a device-side refit marks the copy stale and
 * vt_host_scene_sync refreshes it, exactly as for a host scene the device scene was uploaded from. */
int  vt_host_scene_download(vt_scene* s, vt_host_scene** out);
void vt_host_scene_free(vt_host_scene* hs);

/* 11: what ONE accel:Traverse(origin, dir) does, source/objects/AccelStruct.cpp:818 (see "the single-ray latency path" below) */
int vt_host_scene_trace_closest(const vt_host_scene* hs, const vt_ray* rays, uint64_t n, vt_hit* hits);

/* 12: batches */
/* The call at source/objects/AccelStruct.cpp:818, batched: closest hit per ray. */
/* Rays with a NaN or infinite origin / direction component miss, as in the reference -- but without the walk the
 * reference performs for them (its slab test ignores NaN terms, so such a ray can visit the whole tree before
 * missing); vt_trace_stats_dev alone still walks, to report the reference's counters. */
int vt_trace_closest(vt_scene* s, const vt_ray* rays, uint64_t n, vt_hit* hits);

/* ==== EXTENSIONS ================================================================================================================ */
int vt_abi_version(void);

/* ---- host side: Rebuild (CPU, as in the reference) --------------------------------- */

/* vt_bvh_build with the builder named explicitly (the three builders: at vt_bvh_build above), and the tree's arrays */
enum vt_builder { VT_BUILDER_PLOC = 0, VT_BUILDER_BINNED_SAH = 1, VT_BUILDER_BINNED_SAH_REFINED = 2 };
int             vt_bvh_build_ex(const vt_tri64* tris, uint32_t n, int nthreads, int builder, vt_bvh** out);
uint32_t        vt_bvh_node_count(const vt_bvh* bvh);
uint32_t        vt_bvh_prim_count(const vt_bvh* bvh);
const vt_bvh_node* vt_bvh_nodes(const vt_bvh* bvh);
const uint32_t* vt_bvh_prim_indices(const vt_bvh* bvh);

/* Refit: keep the topology, recompute every node's bounds bottom-up from
 * moved triangles (same count, same order).  Cheaper than a Rebuild when entities only deform; the
 * tree gets looser as they move far, results stay exact for the tree as refitted. */
int             vt_bvh_refit(vt_bvh* bvh, const vt_tri64* tris);

/* Re-pack for the device (replaces `new Intersector(mAccel, mTriangles.data())` /
 * `new Traverser(mAccel)`, source/objects/AccelStruct.cpp:772-773): sibling pairs
 * in depth-first order, triangles pre-shuffled into leaf order. */
int                 vt_scene_linearise(const vt_bvh* bvh, const vt_tri64* tris, vt_host_scene** out);
uint32_t            vt_host_scene_pair_count(const vt_host_scene* hs);
uint32_t            vt_host_scene_tri_count(const vt_host_scene* hs);
uint32_t            vt_host_scene_max_depth(const vt_host_scene* hs);
uint32_t            vt_host_scene_root_leaf_count(const vt_host_scene* hs);
const vt_node_pair* vt_host_scene_pairs(const vt_host_scene* hs);
const vt_tri64*     vt_host_scene_tris(const vt_host_scene* hs);

/* ---- device side ---------------------------------------------------------------------- */

int  vt_device_count(int* count);

/* ---- multi-GPU: BVH replicated in every device's HBM, rays sharded, ONE RCCL gather of hits ----
 * Rays never interact and the scene is read-only, so the path shards by independent units; the only exchange is the
 * gather of the 16-byte hit records to the root device (ncclGather, rccl.h:745 -- N-1 direct xGMI sends into the root).
 * The reference has nothing like it (one ray per call on one thread); this is the seam its C++ module would use.
 *
 * vt_engine_open_multi: ONE process, ndev devices.  The returned engine is the group's root (devices[0]):
 *   - vt_scene_upload / vt_scene_upload_tree (and every later call that changes the scene: refit, skinning, attribs, alpha)
 *     apply to all devices -- refit and skin refit in three phases over the members (prepare, enqueue, finish: no member is
 *     waited for before the last one's work has been enqueued); vt_engine_set_option / vt_engine_synchronize / vt_engine_close act on the whole group;
 *   - vt_trace_closest / vt_trace_any split a host ray array of >= 1 Mi rays into contiguous shards
 *     (vt_shard_bounds) and run one staging pipeline per device side by side; results land in the caller's array;
 *   - vt_trace_closest_gather_dev traces device-resident shards and gathers the hit records to the root device;
 *   - the *_dev entry points keep working on the root device alone.
 * RCCL is loaded on first use (dlopen librccl.so; override with VT_RCCL_LIB). */
int vt_engine_open_multi(const int* devices, int ndev, vt_engine** out);
int vt_engine_device_count(const vt_engine* e);          /* 1 for vt_engine_open */
int vt_engine_device(const vt_engine* e, int g);         /* HIP device of group member g (0 = root), -1 if out of range */
/* Member g of a group (0 = the root itself), NULL if out of range: a BORROWED handle, owned by the root, for the per-engine
 * queries (vt_engine_last_kernel_ms, vt_engine_last_gather_ms, vt_engine_launch_info, vt_engine_get_option).  Never close it,
 * never set options on it (options go to the root and reach every member), never upload scenes through it. */
vt_engine* vt_engine_member(vt_engine* e, int g);
/* Contiguous sharding (keeps the coherence of primary rays inside a shard): capacity = ceil(n / ndev) rounded up to
 * 64 rays, shard g = [g * capacity, min(n, (g + 1) * capacity)) -- the last shards may be short or empty. */
uint64_t vt_shard_capacity(uint64_t n, int ndev);
void     vt_shard_bounds(uint64_t n, int ndev, int g, uint64_t* lo, uint64_t* hi);
/* d_rays[g] (on device g of the group) holds the rays of shard g of an n-ray batch.  Every device traces its shard
 * (closest hit) on its own stream, then ONE ncclGather of `capacity` records per device brings the shards to the root
 * device: d_hits_root (root device, ndev * capacity records) holds ray i's hit at record i.  Asynchronous: returns
 * with traces and gather enqueued (the gather runs on a communication stream, so the next batch's traces overlap
 * it; per-device send buffers are double-buffered) -- vt_engine_synchronize(root) waits for all of it.
 * Ownership of d_hits_root: from this call until the batch's gather has completed (vt_engine_synchronize(root)) the
 * buffer belongs to the engine -- the root device traces straight into its first `capacity` records (ncclGather in
 * place) and the other shards arrive later.  To overlap batches pass a DIFFERENT buffer to the next call (alternate two)
 * and consume a buffer only after a synchronisation that covers its batch; passing the same buffer again lets batch
 * b + 1 overwrite shard 0 while batch b is still arriving.  The engines' streams are not ordered against the caller's: what
 * produced d_rays and whatever last wrote d_hits_root (a memset on another stream, say) must have COMPLETED before the call.
 * STATUS: experimental for ndev > 1 -- no multi-GPU node was available.  Exercised: one device per group with real RCCL; groups
 * of 2 - 8 members on one GPU against a test double for RCCL (tests/test_gpu_fake_group.py: every step below runs, the transfers
 * are stream-ordered device copies); the order of traces, waits and gathers on a simulated group
 * (tests/cpp/test_gather_schedule.cpp, vistrace_amd/csrc/gather_schedule.h).  The engine must come from
 * vt_engine_open_multi: an engine whose communicator was made by vt_engine_comm_init_rank is refused. */
int vt_trace_closest_gather_dev(vt_scene* s, const void* const* d_rays, uint64_t n, void* d_hits_root);
/* One process per GPU (e.g. under torch.distributed.run): the same gather with one communicator per process.
 * Rank 0 calls vt_comm_unique_id (128 bytes) and distributes it (any transport); every rank then calls
 * vt_engine_comm_init_rank on its single-device engine.  vt_gather_hits_dev: `count` records from d_send of every rank
 * into d_recv_root on `root` (nranks * count records, rank r's at r * count); the records must have been produced on
 * `stream`; the gather itself runs on the engine's communication stream.  The root may trace straight into its slice
 * (d_send == d_recv_root + root * count records: in place, no local copy).  vt_gather_wait(e, 1, stream): make
 * `stream` wait until at most ONE gather is still in flight (call it before overwriting the older of two alternating
 * send buffers); vt_gather_wait(e, 0, NULL): host-wait for every gather.  Engine option "gather_overlap" = 0 makes
 * vt_gather_wait(e, 1, stream) wait for the latest gather too (diagnostic: step = trace + gather).  With
 * vt_engine_set_timing on, vt_engine_last_gather_ms reports the duration of the latest batch's gather on the communication
 * stream: HIP events in front of the batch's first piece and behind its last one (one ncclGather when the batch is one piece;
 * with K > 1 pieces the span includes the waits for the traces of pieces 1 .. K-1).  Single-process group: the root's figure
 * (it receives every shard). */
int vt_comm_unique_id(void* id128);
int vt_engine_comm_init_rank(vt_engine* e, int nranks, int rank, const void* id128);
int vt_gather_hits_dev(vt_engine* e, const void* d_send, uint64_t count, void* d_recv_root, int root, void* stream);
int vt_gather_wait(vt_engine* e, int batches_in_flight, void* stream);
/* ONE batch in pieces (round 4): a one-shot batch costs trace + gather, because its gather can only start when its trace
 * has ended.  Cut into K pieces -- piece c = records [lo, hi) of every shard, vt_gather_chunk_bounds(count, K, c) -- piece c
 * crosses the links while piece c + 1 is traced, and the batch costs about max(trace, gather) + one piece.
 *   single process: engine option "gather_chunks" = K (1 .. 16) makes vt_trace_closest_gather_dev trace and gather every
 *     device's shard in K pieces (same arguments, same result array);
 *   one process per GPU: after vt_gather_wait(e, 1, stream), for c = 0 .. K-1: trace the rays of piece c into d_send + lo
 *     records on `stream`, then vt_gather_hits_part_dev(e, d_send, count, c, K, d_recv_root, root, stream).  Every rank
 *     passes the same count and K; the pieces of a batch are handed over in order.  vt_gather_hits_dev = one piece.
 *     Piece 0 always starts a new batch: a batch that was abandoned between two pieces (the caller's own trace failed) is
 *     dropped by the next piece 0 (or vt_gather_hits_dev) -- on EVERY rank, or the ranks' send / receive counts no longer
 *     match.  With count == 0 every piece is a no-op.
 * A piece of every shard cannot land at its final place through ncclGather (it puts rank r's data at r * piece size), so
 * pieces move as the sends and receives ncclGather consists of, one group per piece; K = 1 is the single ncclGather.
 * Each extra piece costs one more launch (~0.3 ms of drain): worth it for a one-shot batch whose gather is as long as its
 * trace (configs[4]); in a stream of batches the gather of batch b already hides behind the trace of batch b + 1. */
void vt_gather_chunk_bounds(uint64_t count, int nchunks, int chunk, uint64_t* lo, uint64_t* hi);
int  vt_gather_hits_part_dev(vt_engine* e, const void* d_send, uint64_t count, int chunk, int nchunks, void* d_recv_root, int root, void* stream);
int vt_engine_last_gather_ms(vt_engine* e, float* ms);

/* Upload once per Rebuild (north star: "uploaded once per Rebuild"). */
int  vt_scene_upload(vt_engine* e, const vt_host_scene* hs, vt_scene** out);
/* Where the time of a scene's upload went (host clock, ms): alloc = staging block, copy = issuing the host -> device copies,
 * device = kernels + read-backs + waiting for them, total = the whole call (for a group's root: all members). */
typedef struct vt_upload_stats { float alloc_ms, copy_ms, device_ms, total_ms; uint64_t bytes_h2d; uint32_t linearised_on_device; uint32_t pad; } vt_upload_stats;
int  vt_scene_upload_stats(const vt_scene* s, vt_upload_stats* out);
uint64_t vt_scene_device_bytes(const vt_scene* s);

/* bvh v1 AnyPrimitiveIntersector semantics (any_hit early-out): occluded[i] = 0/1. */
int vt_trace_any(vt_scene* s, const vt_ray* rays, uint64_t n, uint8_t* occluded);

/* Host arrays that a caller keeps across calls (a frame's ray and hit arrays) can be page-locked once: vt_trace_closest /
 * vt_trace_any then skip their staging copies -- the copy engines read the rays and write the results in place, uploads and
 * downloads overlap (16 Mi rays: 9.6 ms instead of 14).  Locking costs ~70 us per MB, so it pays for arrays that are reused;
 * arrays from hipHostMalloc or a pinned torch tensor are recognised without it.  Unregister before freeing the memory. */
int vt_host_register(void* p, size_t bytes);
int vt_host_unregister(void* p);

/* Device-pointer variants: d_rays/d_hits live on the scene's device; enqueued on
 * `stream`, a hipStream_t with HIP's own meaning (NULL = the legacy default stream, so a
 * caller that works on the default stream stays ordered); no host sync.
 * vt_engine_stream() returns the engine's private non-blocking stream. */
int vt_trace_closest_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* stream);
int vt_trace_any_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_occluded, void* stream);
/* Several batches in ONE launch.  A launch costs ~0.3 ms beyond its rays (grid start, and the drain in which every wave walks
 * its last, longest rays with most lanes idle): 7 % of a 16 Mi-ray batch, 40 % of a 1 Mi-ray one, nearly all of a 64 Ki-ray
 * one -- what a per-frame caller with many small ray sets (one per light, per tile, per entity; the reference's call shape is
 * one ray per call, source/VisTrace.cpp:831-836) pays again for every set.  Here the ray blocks of all batches are numbered
 * through and handed out by one cursor, so the batches share one start and one drain; every batch keeps its own ray and result
 * arrays (any device addresses, 16-B aligned; result ranges must not overlap) and its own ray_image_width (0 = not in image
 * order; see the engine option of that name, which the single-batch calls use instead).  Results are exactly those of nbatches
 * separate calls.  Batches may be empty; in a launch of several, each holds fewer than 2^32 rays.  Asynchronous like the other
 * _dev calls; `batches` is read before the call returns.  d_out: n x vt_hit (closest) or n bytes (any). */
typedef struct vt_batch_desc {
    const void* d_rays;
    void*       d_out;
    uint64_t    n;
    uint32_t    ray_image_width;
    uint32_t    reserved;          /* 0 */
} vt_batch_desc;
int vt_trace_closest_multi_dev(vt_scene* s, const vt_batch_desc* batches, uint32_t nbatches, void* stream);
int vt_trace_any_multi_dev(vt_scene* s, const vt_batch_desc* batches, uint32_t nbatches, void* stream);
/* Closest hit + per-ray counters (diagnostic kernel; same visitation order). */
int vt_trace_stats_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits,
                       void* d_ray_stats, void* stream);
/* Any hit + per-ray counters: the node steps and triangle tests the reference's AnyPrimitiveIntersector walk performs
 * up to its early-out (what SURVEY.md 8(d)'s algorithmic bytes of a shadow-ray batch are counted from). */
int vt_trace_any_stats_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_occluded,
                           void* d_ray_stats, void* stream);
/* TraceResult batch materialisation from hits (d_attrs: n x vt_hit_attrs). */
int vt_hit_attrs_dev(vt_scene* s, const void* d_rays, const void* d_hits, uint64_t n,
                     void* d_attrs, void* stream);

/* ---- one traced batch, kept on the device (what a GLua `accel:TraverseBatch(buffer)` returns a handle to) -----------
 * The reference builds one TraceResult per ray on the host (source/objects/AccelStruct.cpp:825-831,
 * source/objects/TraceResult.cpp:45-86); for a batch that is N constructor calls and N allocations.  vt_batch keeps the
 * batch where it was traced: vt_batch_trace_closest uploads the rays, traces them and materialises vt_hit_attrs (and
 * vt_hit_shade when vt_scene_set_tri_attribs has been called) with the device kernels, all enqueued on the engine's
 * stream; nothing comes back until it is asked for.  vt_batch_hits / _attrs / _shade download their array ONCE (first
 * call; the pointer stays valid until vt_batch_free) -- a consumer that only reads distances never pays for the rest.
 * The batch owns its device memory; free it before closing the engine (vt_engine_close releases what is left and the
 * arrays not yet downloaded are then lost: the getters fail).  Single device (a group's root).  `rays` may be any byte
 * address (it is only copied from; a Lua string carries no alignment promise).  A freed batch's device block is kept by
 * the engine and handed to the next batch that fits (no hipMalloc per batch in a steady stream of equal batches). */
typedef struct vt_batch vt_batch;
int      vt_batch_trace_closest(vt_scene* s, const vt_ray* rays, uint64_t n, vt_batch** out);
/* The same with what a scripting front end needs on the way in and out (round 4: accel:TraverseBatch(buffer) went from 106 to
 * > 500 Mrays/s end to end at 1 Mi rays).  The batch flows in chunks of 256 Ki rays: chunk c is uploaded while chunk c - 1 is
 * traced -- straight from the caller's memory where the runtime copies pageable memory at the pinned rate (measured once per
 * process; VT_BATCH_UPLOAD=staged|direct in the environment decides instead), else through pinned staging buffers filled by a few
 * host threads.
 *   ray_image_width        as vt_batch_desc::ray_image_width (vt_batch_trace_closest takes the engine option instead);
 *   VT_BATCH_CHECK_RANGES  the upload looks at every ray's range (a device kernel behind each chunk, or the staging copy): tMin < 0 or tMax <= tMin (the checks of
 *                          AccelStruct::Traverse, source/objects/AccelStruct.cpp:805-806) fails the call with
 *                          VT_ERR_INVALID_ARG and *bad_ray = the first such ray (else *bad_ray = n); no batch is returned;
 *   VT_BATCH_FETCH_HITS    the hit records of chunk c - 2 come back into pinned host memory while chunk c - 1 is traced:
 *                          vt_batch_hits then only waits for the last chunk instead of downloading n records. */
#define VT_BATCH_CHECK_RANGES 1u
#define VT_BATCH_FETCH_HITS   2u
int      vt_batch_trace_closest_ex(vt_scene* s, const vt_ray* rays, uint64_t n, uint32_t ray_image_width, uint32_t flags, uint64_t* bad_ray,
                                   vt_batch** out);
/* A SET of batches from nbatches host buffers (a script's ray sets of one frame: per light, per tile, per entity): every buffer is
 * staged and uploaded as above, then ALL of them are traced by one merged launch (vt_trace_closest_multi_dev: one grid start and
 * one drain instead of nbatches), then every batch gets its result kernels and -- with VT_BATCH_FETCH_HITS -- its download.  out
 * receives nbatches handles, each a vt_batch like any other (freed one by one).  ray_image_widths may be NULL (no images).  With
 * VT_BATCH_CHECK_RANGES the first offender is named by *bad_batch / *bad_ray and no batch is returned.  16 x 64 Ki rays into 1 M
 * triangles: 0.50 ms of tracing instead of 2.7 (profiles/r4/merged_*.txt). */
int      vt_batch_trace_closest_set(vt_scene* s, const vt_ray* const* rays, const uint64_t* n, const uint32_t* ray_image_widths,
                                    uint32_t nbatches, uint32_t flags, uint32_t* bad_batch, uint64_t* bad_ray, vt_batch** out);
/* The same, buffer by buffer -- for a front end that sees its buffers one at a time (a Lua binding walking a table: each string is
 * only guaranteed to stay while it is on the stack).  vt_batch_set_add stages and uploads its buffer before it returns (the caller's
 * memory is free then; with VT_BATCH_CHECK_RANGES a bad ray fails the add and leaves the set as it was); vt_batch_set_trace enqueues
 * the merged launch, writes vt_batch_set_count handles to `out` (in the order of the adds) and consumes the set, also when it fails;
 * vt_batch_set_abort drops a set that will not be traced.  One set at a time per thread of control; single device. */
typedef struct vt_batch_set vt_batch_set;
int      vt_batch_set_begin(vt_scene* s, uint32_t flags, vt_batch_set** out);
int      vt_batch_set_add(vt_batch_set* set, const vt_ray* rays, uint64_t n, uint32_t ray_image_width, uint64_t* bad_ray);
uint32_t vt_batch_set_count(const vt_batch_set* set);
int      vt_batch_set_trace(vt_batch_set* set, vt_batch** out);
void     vt_batch_set_abort(vt_batch_set* set);
uint64_t vt_batch_count(const vt_batch* b);
int      vt_batch_rays(vt_batch* b, const vt_ray** rays);     /* the rays as uploaded (so a caller need not keep its copy) */
int      vt_batch_hits(vt_batch* b, const vt_hit** hits);
int      vt_batch_attrs(vt_batch* b, const vt_hit_attrs** attrs);
int      vt_batch_shade(vt_batch* b, const vt_hit_shade** shade);     /* VT_ERR_INVALID_ARG without vt_scene_set_tri_attribs */
int      vt_batch_tbn(vt_batch* b, const vt_hit_tbn** tbn);           /* VT_ERR_INVALID_ARG without vt_scene_set_tri_frames; cone off */
void     vt_batch_free(vt_batch* b);

/* Launch configuration (also readable from VT_* environment variables at vt_engine_open).  Keys:
 *   "persistent"         0 = one ray per lane, 1 = persistent waves, 2 = auto by batch size (default)
 *   "auto_static_factor" auto: one ray per lane when n <= f x (CUs x 8 x 256) rays; f = 2 x factor for scenes up to
 *                        200 k node pairs, factor / 2 above (default factor 2: 2 Mi rays / 512 Ki rays on MI355X)
 *   "fetch_dma"          persistent kernel: quad-cooperative global->LDS record fetch (default 1)
 *   "coherent_detect"    persistent kernel: per-wave octant probe -> direct fetch + whole-wave re-fill (1)
 *   "lds_entries"        stack entries per lane kept in LDS, the rest spills to global memory (10)
 *   "blocks_per_cu"      cap on resident 256-thread blocks per CU (8; the occupancy query decides below it)
 *   "block_rays"         consecutive rays handed to a wave at a time (128)
 *   "refill_threshold"   idle lanes that trigger a re-fill (8);  "tri_threshold": waiting lanes that
 *                        trigger the triangle branch (4);  "static_overflow_mb": overflow-area cap of the
 *                        one-ray-per-lane kernel (256)
 *   "max_claim"          persistent kernel: ray blocks one atomic on the cursor may claim while plenty are left (guided
 *                        self-scheduling; single blocks near the end).  The cursor is ONE word and serves ~90 M
 *                        claims per second, i.e. at most 5.8 Grays/s with 64-ray claims: rays into small scenes are
 *                        cheaper than that (16 Mi bounce rays into 100 k triangles: 3.15 -> 2.48 ms with 4, primary
 *                        3.03 -> 1.45 ms), rays into 1 M triangles are not and lose locality (5.23 -> 5.36 ms with 8).
 *                        0 (default) = 4 for scenes up to 200 k node pairs, else 1.
 *   "xcd_cursors"        persistent kernel: one ray-block cursor per XCD, each over its own eighth of the batch, with
 *                        stealing (default 0).  Keeps neighbouring rays in one L2: S1M primary rays 3.34 -> 3.20 ms,
 *                        but eight distant ray ranges in flight enlarge the Infinity-Cache working set: S10M bounce
 *                        rays 8.28 -> 8.62 ms; no effect on S1M bounce / S10M primary.
 *   "ray_image_width"    rays per image row of the batches to come (0 = unknown, the default).  For batches in image order
 *                        (camera rays, row-major) a wave then takes its 64 rays as a 4-wide, 16-high pixel tile instead of 64
 *                        neighbours of one row, so its lanes walk the same nodes for longer: 16 Mi camera rays into 1 M
 *                        triangles 2.00 -> 1.87 ms, into 10 M 4.24 -> 3.76 ms, 1 Mi into 100 k 0.113 -> 0.095 ms.  Scheduling
 *                        only: the ray and hit arrays keep their order and every result is unchanged.  Needs a multiple of 4;
 *                        whole bands of 16 rows are tiled, the rest of a batch is taken in order.  The hint is also a PROMISE
 *                        that the batch is coherent (camera rays): with "persistent" = 2 such batches run one ray per lane at
 *                        every size while the scene's records fit the caches (<= 256 MiB: 1 M triangles 4 Mi rays 0.84 ->
 *                        0.59 ms).  Do not set it for bounce or shadow rays, even when they are stored in image order: their
 *                        directions differ from pixel to pixel, tiles gain nothing and the kernel choice would be the
 *                        wrong one (16 Mi bounce rays: 7.9 instead of 4.2 ms).  Not used by the alpha-test kernels.
 *   "spin_wait"          host batches of <= 256 rays: watch the pinned result slots change instead of waiting on
 *                        the stream (default 1; saves ~5 us of the ~24 us single-ray call)
 *   "reserved_cus"       CUs on which the persistent grid leaves room (0 = off): set it when another stream runs
 *                        kernels that must make progress during a trace (e.g. the RCCL gather of the previous
 *                        batch).  A resident persistent grid holds every CU's LDS and registers until its last
 *                        ray, so such kernels would otherwise only start when the trace ends.  The CUs are
 *                        chosen one per shader engine of every XCD in turn (the dispatcher binds a workgroup to
 *                        an XCD/SE before it looks for a CU, so 32 = one per SE is the useful value on MI355X);
 *                        Keep traces on ONE stream while it is set: a second persistent grid launched beside a resident one
 *                        finds free room only on the reserved CUs, where its blocks leave at once (measured: two 8 Mi-ray
 *                        launches on two streams 40 ms instead of 5);
 *   "reserved_limit"     blocks of the grid a reserved CU still keeps (2: measured to leave room for one
 *                        256-thread workgroup with the footprint of RCCL's kernel, 280 VGPRs + 20 KB LDS --
 *                        3 does not; 0 = keep the CU empty).  Later arrivals on a full reserved CU exit at once.
 *   "gather_overlap"     multi-GPU: 0 = every trace waits for the previous batch's gather (diagnostic: step = trace + gather)
 *   "gather_chunks"      multi-GPU, single-process form: pieces a shard is traced and gathered in (1 .. 16, default 1; see
 *                        vt_gather_hits_part_dev)
 *   "alpha_threshold"    scenes with alpha-tested triangles: lanes with a parked candidate hit that trigger the block which
 *                        turns their AlphaRecs into texel addresses (4; 1 = as soon as one lane has a candidate)
 * Read-only: "cu_count", "device", "device_count", "last_persistent", "last_fetch_dma" (what the last launch used);
 * "last_update_members", "last_update_early_waits", "last_update_enqueue_us", "last_update_wait_us" (the latest vt_scene_refit /
 * vt_scene_skin_refit of a scene of this engine: members it reached, members waited for before the last member's work had been
 * enqueued -- 0 by construction --, host time of the prepare + enqueue phases and of the waits).
 * Results never depend on these, only speed does. */
int vt_engine_set_option(vt_engine* e, const char* key, int64_t value);
int vt_engine_get_option(vt_engine* e, const char* key, int64_t* value);
void* vt_engine_stream(vt_engine* e);
/* Wait for everything enqueued on the engine's own stream. */
int vt_engine_synchronize(vt_engine* e);

/* Device-side refit of an uploaded scene: new vertices (n x 9 floats, host memory, original triangle
 * order; n = the scene's triangle count; flags may be NULL = unchanged) -> triangle records and all
 * pair bounds are recomputed in place on the device, level by level from the leaves up.  Produces
 * exactly the records vt_tris_setup + vt_bvh_refit + vt_scene_linearise would.  Vertices must be finite: NaN boxes
 * pass every slab test, so every ray would walk the poisoned subtree -- the call then fails with VT_ERR_INVALID_ARG
 * (the message counts the triangles) and the scene refuses to trace until a refit with finite data (the same holds
 * for vt_scene_skin_refit, e.g. with a NaN bone matrix).  In a multi-GPU group both calls reach EVERY member before a failure is
 * reported, so the members never hold different geometry: a refused refit leaves the whole group refusing to trace. */
int vt_scene_refit(vt_scene* s, const float* verts, const uint8_t* flags, uint32_t n);

/* Device-side skinning + refit (the per-frame half of AccelStruct::Rebuild for animated entities,
 * source/objects/AccelStruct.cpp:34-102 SkinTriangle/TransformToBone, call site :749).
 * vt_skin_vertex = one vertex's Triangle::weights / boneIds / numBones (source/objects/Primitives.h:68-70). */
typedef struct vt_skin_vertex { float weight[3]; int8_t bone[3]; uint8_t num_bones; } vt_skin_vertex;
/* Once per build: the mesh triangles in bind pose (n x 9 floats, original order; n = the scene's triangle
 * count), their 3 x n skin vertices, and for every triangle the index of its entity's first matrix in the
 * arrays later given to vt_scene_skin_refit (boneIds are relative to it).  Static triangles use one bone
 * of weight 1 and an identity matrix pair.  Host pointers; the data stays on the device. */
int vt_scene_set_skin(vt_scene* s, const float* bind_verts, const vt_skin_vertex* skin, const uint32_t* matrix_base,
                      uint32_t n);
/* Per frame: bones[i] (Entity:GetBoneMatrix, AccelStruct.cpp:655-667) and binds[i] (the model's bind
 * matrices, :668) as nmat glm::mat4 (16 floats, column-major) each; only these 128 x nmat bytes cross the
 * bus.  Every vertex is moved by TransformToBone, the triangle records are rebuilt (Primitives.h:82,93) and
 * all pair bounds refitted, exactly as vt_scene_refit would from the skinned vertices. */
int vt_scene_skin_refit(vt_scene* s, const float* bones, const float* binds, uint32_t nmat);

/* Copy the device-resident records back (pairs: vt_host_scene_pair_count entries, tris: leaf order);
 * either pointer may be NULL.  For inspection and tests. */
int vt_scene_read_records(vt_scene* s, vt_node_pair* pairs_out, vt_tri64* tris_out);

/* After vt_scene_refit / vt_scene_skin_refit: bring the host copy that vt_host_scene_trace_* walk (the single-ray path of
 * accel:Traverse, source/objects/AccelStruct.cpp:818) up to date with the device -- the records are read back into `hs`
 * (S1M: 101 MB), which must be the host scene `s` was uploaded from.  Not thread safe against host walks of `hs`. */
int vt_host_scene_sync(vt_host_scene* hs, vt_scene* s);

/* Optional per-triangle side table (n must equal the scene's triangle count); copied to the device. */
int vt_scene_set_tri_attribs(vt_scene* s, const vt_tri_attribs* attribs, uint32_t n);
/* Alpha test inside the triangle test (source/objects/Primitives.h:196-208).  In tree and followed exactly: texUV
 * = (1-u-v)*uvs[0] + u*uvs[1] + v*uvs[2], TransformTexcoord (source/Utils.h:65-72), `alpha < alphatestreference`
 * discards the hit (the walk continues, tmax unchanged).  NOT in tree: `baseTexture->Sample(u, v, 0.f).a` forwards
 * to the VTFParser submodule (source/objects/VTFTexture.cpp:68-72), which is absent and unpinned -- the lookup is
 * therefore DEFINED here: the host hands over the decoded mip-0 alpha plane of each material's base texture
 * (8 bits, row-major), addressing is repeat, alpha = a / 255, filter 0 = nearest texel floor(s*W), filter 1 =
 * bilinear with texel centres at (i + 0.5) / W.  A maintainer with VTFParser at hand picks the filter that matches
 * (or pre-filters the plane).  width = height = 0: no texture, alpha 1.  64 B. */
typedef struct vt_alpha_material {
    float tex_mat[2][4];     /* Material::baseTexMat (source/objects/Material.h): tex_mat[r] = transform[r]   */
    float tex_scale;         /* Material::texScale                                                            */
    float alpha_ref;         /* Material::alphatestreference (default 0.5, Material.h:122)                    */
    uint32_t width, height;
    uint32_t filter;
    uint32_t pad;
    uint64_t offset;         /* first texel of this material in `texels`                                      */
} vt_alpha_material;
/* Materials are indexed by vt_tri_attribs::material, uvs come from vt_tri_attribs::uv: call
 * vt_scene_set_tri_attribs first.  A scene that holds VT_TRI_ALPHATEST triangles cannot be traced until this is
 * set (VT_ERR_UNSUPPORTED); scenes without such triangles run the kernels compiled without the test. */
int vt_scene_set_alpha(vt_scene* s, const vt_alpha_material* mats, uint32_t nmats, const uint8_t* texels, uint64_t ntexels);

/* ---- host side: the single-ray latency path (BASELINE config 1) -------------------------------
 * What one `accel:Traverse(origin, dir)` call from GLua does (source/objects/AccelStruct.cpp:810-820): ONE ray.  A
 * lone ray on the GPU is launch bound (~20 us) while the same walk takes 1-2 us on a host core, so the host class
 * keeps the vt_host_scene it uploaded and answers single rays and batches below the crossover (~128 rays) here: the
 * reference's scalar walk (bvh v1 SingleRayTraverser + FastNodeIntersector + the in-tree triangle test) on the
 * linearised records, bit-identical to the device kernels' results.  This is a separate, explicitly named path, NOT
 * a fallback: vt_trace_* and the *_dev entries never run on the CPU, and without a HIP device no vt_scene exists.
 * Serial; callable from any thread (the scene is read-only).  Non-finite rays miss without the reference's walk,
 * as on the device.  Scenes with VT_TRI_ALPHATEST triangles need vt_host_scene_set_alpha first (VT_ERR_UNSUPPORTED). */
int vt_host_scene_trace_any(const vt_host_scene* hs, const vt_ray* rays, uint64_t n, uint8_t* occluded);
/* The alpha-test side data for the host walk: the same tables vt_scene_set_tri_attribs + vt_scene_set_alpha take. */
int vt_host_scene_set_alpha(vt_host_scene* hs, const vt_tri_attribs* attribs, uint32_t ntris, const vt_alpha_material* mats,
                            uint32_t nmats, const uint8_t* texels, uint64_t ntexels);

/* entIdx / texUV / blendFactor / submatIdx per hit (needs vt_scene_set_tri_attribs). d_out: n x vt_hit_shade. */
int vt_hit_shade_dev(vt_scene* s, const void* d_hits, uint64_t n, void* d_out, void* stream);

/* ---- shading frame of a hit: TraceResult::GetNormal / GetTangent / GetBinormal in bulk (vt_tri_frame, vt_hit_tbn above) ---- */
/* Optional side table (n must equal the scene's triangle count); copied to the device.  With skin data present
 * (vt_scene_set_skin) these are the BIND-pose frames: every vt_scene_skin_refit also moves them by TransformToBone with
 * angleOnly = true (source/objects/AccelStruct.cpp:82-92), as SkinTriangle does. */
int vt_scene_set_tri_frames(vt_scene* s, const vt_tri_frame* frames, uint32_t n);
/* d_out: n x vt_hit_tbn.  Needs vt_scene_set_tri_frames and, with the cone on (lod_info.x is derived from the uvs),
 * vt_scene_set_tri_attribs.  cone_width / cone_angle as accel:Traverse's coneWidth / coneAngle (one pair per batch). */
int vt_hit_tbn_dev(vt_scene* s, const void* d_rays, const void* d_hits, uint64_t n, float cone_width, float cone_angle,
                   void* d_out, void* stream);
/* Copy the current (skinned) frames back, original triangle order.  For inspection and tests. */
int vt_scene_read_tri_frames(vt_scene* s, vt_tri_frame* frames_out);

/* Device-side ray generation for wavefront-style callers; nothing is traced.
 * vt_gen_primary_dev: width*height normalised pinhole rays, range [0, FLT_MAX].
 * vt_gen_bounce_dev : one ray per hit record: origin = vistrace.CalcRayOrigin(Pos, Ng facing wo)
 *   (source/VisTrace.cpp:1495-1517), direction = cosine hemisphere about that normal with the mapping of
 *   hemisphere_cos (source/libraries/BSDF.cpp:69-77) and counter-based splitmix64 samples (seed, 2i / 2i+1);
 *   a missed record yields a null ray (tmax = 1e-30) so that the batch keeps its size and order. */
int vt_gen_primary_dev(vt_engine* e, const vt_camera* cam, void* d_rays, void* stream);
int vt_gen_bounce_dev(vt_engine* e, const void* d_attrs, uint64_t n, uint64_t seed, void* d_rays, void* stream);

/* Device-resident bounce loop (wavefront queue of live paths).  Starting from n rays
 * (d_rays, left untouched), `depth` times: trace the live paths' rays (closest hit), write the hits to row d of
 * d_hits (depth x n vt_hit, row-major, indexed by path = index of the starting ray; a path that has already
 * ended reads VT_MISS/0/0/0), and give every path that hit something its next ray -- exactly
 * vt_gen_bounce_dev(vt_hit_attrs_dev(...), seed + d) with the path index as the sample counter.  Paths that miss
 * leave the queue, which is compacted in path order, so the result does not depend on scheduling and equals the
 * uncompacted composition of the three calls.  The whole loop is ENQUEUED: no host wait per depth -- the live-path
 * count stays on the device, the launches behind depth 0 are sized for n and read it from there.  live_out (host,
 * depth entries, may be NULL): paths traced at each depth; entry 0 is written at once, the others by a host function
 * in stream order -- they are valid once `stream` has passed the call (synchronise it, or an event recorded behind
 * the call, before reading them), and the array must stay alive until then.  (A scene with alpha-tested triangles
 * waits for the count once per depth, as round 5 did: live_out is then complete on return.) */
int vt_bounce_loop_dev(vt_scene* s, const void* d_rays, uint64_t n, uint32_t depth, uint64_t seed, void* d_hits,
                       uint64_t* live_out, void* stream);

/* When enabled, every trace launch is bracketed by HIP events on its stream;
 * vt_engine_last_kernel_ms synchronises on the last pair and returns its time. */
int vt_engine_set_timing(vt_engine* e, int enabled);
int vt_engine_last_kernel_ms(vt_engine* e, float* ms);
/* Launch geometry actually used (for DESIGN.md / bench reporting). */
int vt_engine_launch_info(vt_engine* e, uint32_t* blocks, uint32_t* threads, uint32_t* lds_bytes);

/* ---- Test hooks ------------------------------------------------------------------------------------------------------
 * Environment switches compiled into the library so that a box with ONE GPU can execute code that a product run reaches
 * only on other hardware.  They are DEAD unless VT_ENABLE_TEST_HOOKS=1 is set beside them, and an active hook announces
 * itself on stderr once ("[vistrace_hip] TEST HOOK active: ...").  Never set them in production.
 *   VT_TEST_ALLOW_DEVICE_ALIASES=1   vt_engine_open_multi accepts a device listed several times (device 0 standing for every
 *                                    member of a group).  Real RCCL refuses such a group: use with VT_RCCL_LIB pointing at
 *                                    tests/cpp/_build/libfake_rccl.so (tests/test_gpu_fake_group.py, bench.py --form group).
 *   VT_TEST_RECORD_GAP=<records>     vt_scene_upload leaves that many unused 64-B records between the pairs and the
 *                                    triangles (a multi-GiB allocation): the 64-bit record addressing of scenes with more than
 *                                    67 M records on a 10 k-triangle scene (tests/test_gpu_parity.py::test_records_beyond_4_gib).
 *   VT_TEST_FAIL_ALLOC=<k>           fault injection: the k-th device / pinned-host allocation the library makes in this process
 *                                    fails as if memory had run out (every hipMalloc / hipHostMalloc of the library goes through
 *                                    one counted wrapper).  vt_test_fail_alloc(k) re-arms it at run time.  What the reference
 *                                    does on such paths: delete-before-throw (source/VisTrace.cpp:782-785,
 *                                    source/objects/AccelStruct.cpp:186-203, :780); here: a non-zero status, vt_last_error set,
 *                                    nothing leaked, the engine still usable (tests/test_gpu_fault_injection.py).
 *   VT_TEST_FAIL_HIP=<k>             the same for every OTHER HIP call the library checks (copies, event and stream calls, launch
 *                                    checks, synchronisations): the k-th checked call reports a failure instead of being made (the
 *                                    devices are drained first, as after a real sticky error).  Contract under test: a non-zero
 *                                    status, vt_last_error set, nothing leaked, the engine still usable; an object that was being
 *                                    UPDATED in place (refit, skin refit, new tables) is unspecified until the same call succeeds
 *                                    on it or it is freed.  vt_test_fail_hip(k) re-arms it at run time.
 * Not hooks but configuration, always honoured: VT_RCCL_LIB (path of the RCCL library to dlopen instead of librccl.so),
 * VT_BUILDER (default builder of vt_bvh_build), VT_BATCH_UPLOAD (staged | direct), VT_COPY_THREADS (staging-copy threads). */

/* Re-arms the fault injection above: the k-th allocation FROM NOW ON fails (once); k = 0 disarms.  The count restarts either
 * way.  VT_ERR_UNSUPPORTED unless VT_ENABLE_TEST_HOOKS=1. */
int vt_test_fail_alloc(uint64_t k);
/* Allocations the library has attempted since the last vt_test_fail_alloc (or since it was loaded); 0 with the hooks off. */
uint64_t vt_test_alloc_count(void);
/* The same pair for VT_TEST_FAIL_HIP: the k-th checked HIP call from now on fails (once); checked calls passed since. */
int vt_test_fail_hip(uint64_t k);
uint64_t vt_test_hip_count(void);

#ifdef __cplusplus
}
#endif
#endif
