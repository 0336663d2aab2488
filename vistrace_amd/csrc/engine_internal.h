// engine_internal.h -- what engine.hip, batch.hip and multi_gpu.hip share: the objects behind the opaque vt_engine / vt_scene
// handles of include/vistrace_hip.h and the two internal entry points every trace goes through.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "gather_schedule.h"
#include "trace_kernels.h"
#include "vt_internal.h"

// VT_TRY(call): the call's own status, for the places that collect a status and clean up themselves instead of returning at once
// (counted by the same fault-injection hook as VT_HIP)
#define VT_TRY(call) (vt::test_hip_fails() ? hipErrorUnknown : (call))

// (test_hip_fails: the second fault-injection hook, vt_internal.h -- the k-th VT_HIP site the library passes reports a failure
// INSTEAD of making its call; false, at the cost of one relaxed load, unless VT_ENABLE_TEST_HOOKS=1)
#define VT_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t err__ = vt::test_hip_fails() ? hipErrorUnknown : (call);                       \
        if (err__ != hipSuccess)                                                                  \
            return fail(VT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err__));        \
    } while (0)

struct vt_engine {
    int         device = 0;
    hipStream_t stream = nullptr;
    int         cu_count = 0;
    size_t      lds_per_block_max = 0;
    size_t      lds_per_cu = 160 * 1024;

    // launch configuration (vt_engine_set_option)
    int      persistent       = 2;    // 0 static (one ray per lane), 1 persistent waves, 2 auto by batch size
    uint32_t coherent_detect  = 1;    // persistent DMA kernel: per-wave coherence probe (see trace_kernels.hip)
    uint32_t static_overflow_mb = 256; // static kernel: largest per-lane stack overflow area (else full LDS stack)
    uint32_t auto_static_factor = 2;  // auto: static when n <= factor * (CUs * 8 blocks * 256 lanes)
    uint32_t lds_entries      = 10;   // stack entries per lane in LDS (rest spills to global)
    uint32_t blocks_per_cu    = 8;    // persistent grid = cu_count * blocks_per_cu
    uint32_t block_rays       = 128;  // consecutive rays handed to a wave at a time (128: primary rays -4 %, bounce rays unchanged)
    uint32_t refill_threshold = 8;    // idle lanes that trigger a re-fill
    uint32_t tri_threshold    = 4;    // lanes with pending triangles that trigger the TRI branch
    uint32_t alpha_threshold  = 4;    // ALPHA kernels: lanes with a parked candidate that trigger the AlphaRec -> texel-address block
    int      fetch_dma        = 1;    // quad-cooperative global->LDS record fetch (persistent mode)
    uint32_t ray_image_width  = 0;    // rays per image row of the batches to come (0 = unknown): lanes take 4 x 16 pixel tiles
    uint32_t max_claim        = 0;    // persistent mode: ray blocks one cursor atomic may claim while plenty are left (0 = auto)
    int      xcd_cursors      = 0;    // persistent mode: one ray-block cursor per XCD over its own eighth of the batch (opt-in)
    int      spin_wait        = 1;    // tiny host batches: watch the pinned result slots instead of a stream sync
    uint32_t reserved_cus     = 0;    // CUs on which the persistent grid leaves room (for a concurrent collective's kernels)
    uint32_t reserved_limit   = 2;    // blocks of the grid a reserved CU still keeps
    uint32_t* d_reserved      = nullptr; // 1024-bit set of the reserved CUs' __smid() values, then 1024 per-CU counters

    // Per-launch scratch: a ring of launch slots, rotated per launch, so that traces in flight on different
    // streams (or enqueued back to back on one) never share a ray cursor, reserved-CU counters or a stack
    // overflow area.  A slot is reused kLaunchSlots launches later; the new launch then waits (on the device,
    // hipStreamWaitEvent) for the event recorded behind the slot's previous launch.
    static constexpr uint32_t kLaunchSlots = 16;
    static constexpr size_t   kSlotCtlBytes = 8192;   // 512 B of cursors (8 x 64 B), then 4 KB of reserved-CU counters
    struct LaunchSlot {
        uint32_t*  d_ctl = nullptr;          // into d_slot_ctl
        uint32_t*  d_overflow = nullptr;     // grown on demand: a piece of ovf_block
        std::shared_ptr<void> ovf_block;     // the allocation d_overflow lies in (shared by the slots that grew together)
        size_t     overflow_words = 0;
        hipEvent_t done = nullptr;
        bool       used = false;
        bool       dirty = false;            // a launch on this slot failed: the cursors are cleared before the next one
        // merged launches (vt_trace_*_multi_dev): the batch table, pinned on the host and on the device
        vt::TraceSeg* h_segs = nullptr;
        vt::TraceSeg* d_segs = nullptr;
        size_t        segs_cap = 0;
        hipEvent_t    segs_copied = nullptr;  // behind the latest copy h_segs -> d_segs: h_segs may be rewritten once it has passed
        bool          segs_copied_valid = false;
        bool          segs_shared = false;    // h_segs / d_segs lie in the engine's one block for all slots (not freed per slot)
    };
    static constexpr uint32_t kSlotSegs = 64;   // batches per merged launch served from the shared table block (more: the slot grows its own)
    char*      d_segs_all = nullptr;            // kLaunchSlots x kSlotSegs batch descriptors, device / pinned host: allocated at the
    char*      h_segs_all = nullptr;            //   engine's first merged launch
    LaunchSlot slots[kLaunchSlots];
    char*      d_slot_ctl = nullptr;
    uint32_t   next_slot = 0;
    std::mutex launch_mu;                    // slot rotation + enqueue (host threads may share an engine)
    std::mutex host_mu;                      // the host-pointer entry points share the staging buffers
    hipEvent_t ev_loop = nullptr;            // behind the last vt_bounce_loop_dev (its queues are engine-wide)
    bool       loop_used = false;
    hipEvent_t ev_staged = nullptr;          // root of a group: behind the upload of a refit's vertices into d_rays (the replicas copy from there)

    // staging for the host-pointer entry points
    void*  d_rays = nullptr;  size_t d_rays_bytes = 0;
    void*  d_out  = nullptr;  size_t d_out_bytes = 0;
    // large host batches: pinned double buffers + copy streams, so that H2D, trace and D2H of successive
    // chunks overlap (pageable hipMemcpyAsync serialises on the host)
    static constexpr uint64_t kHostChunk = uint64_t(1) << 20;   // rays per pipelined chunk
    // kStageBufs buffers of each kind: chunk c uses buffer c % kStageBufs; the host takes chunk c - kStageLag's results out right
    // after it has enqueued chunk c, so it never waits for a chunk that has only just been started (round 4: two buffers and a
    // lag of one left the upload engine idle a third of the time -- 14.0 -> ~10 ms for 16 Mi rays)
    static constexpr int kStageBufs = 4, kStageLag = 2;
    char* h_stage_in[kStageBufs]  = {};
    char* h_stage_out[kStageBufs] = {};
    hipStream_t s_in = nullptr, s_out = nullptr;
    bool pipeline_ready = false;             // streams, staging buffers and events above all exist (batch.hip: ensure_host_pipeline)
    hipEvent_t ev_in[kStageBufs] = {}, ev_k[kStageBufs] = {}, ev_out[kStageBufs] = {};
    // Rebuild (scene_build.hip): staging block of the device lineariser / index tables, kept across Rebuilds; pinned read-back
    void*  d_build = nullptr;  size_t d_build_bytes = 0;
    char*  h_build = nullptr;  size_t h_build_bytes = 0;
    // bounce loop: two ray queues, two path-id queues, the queue's hit records, block offsets, live counter
    void*  d_loop = nullptr;  size_t d_loop_bytes = 0;
    uint32_t* h_live = nullptr;           // pinned read-back of the live-path counts
    size_t    h_live_bytes = 0;
    char*     h_bad = nullptr;            // pinned read-back of a batch upload's first-bad-ray word (batch.hip)
    // single-ray / tiny-batch path: pinned, device-mapped host memory the kernel reads and writes in
    // place (no copy calls: one launch + one stream sync per Traverse)
    static constexpr uint32_t kTinyRays = 256;
    vt_ray* h_tiny_rays = nullptr;  void* d_tiny_rays = nullptr;
    char*   h_tiny_out  = nullptr;  void* d_tiny_out  = nullptr;

    // timing
    int        timing = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool       ev_valid = false;

    // scenes uploaded through this engine: closing the engine releases their device memory and detaches them.  Like `batches`
    // below the list changes under launch_mu: uploads and frees may come from any host thread (track_scene / untrack_scene)
    std::vector<vt_scene*> scenes;
    std::vector<vt_batch*> batches;          // live vt_batch objects (their device arrays go with the engine)
    std::vector<std::pair<char*, size_t>> device_spare;   // device blocks of freed batches, for the next ones that fit (at most 32)
    std::vector<std::pair<void*, size_t>> pinned_spare;   // pinned host blocks of freed batches (their downloaded arrays)

    uint32_t last_update_members = 0, last_update_early_waits = 0, last_update_enqueue_us = 0, last_update_wait_us = 0;   // the latest group-wide refit / skin refit (diagnostic)

    bool alpha_regs_checked = false;         // ALPHA kernels: hipFuncGetAttributes agreed with the build-time ISA check

    // last launch geometry
    uint32_t last_blocks = 0, last_threads = 0, last_lds = 0;
    int      last_persistent = 0, last_dma = 0;

    // ---- multi-GPU (multi_gpu.hip) ------------------------------------------------------------------------------
    // vt_engine_open_multi: this engine is the ROOT of a single-process group (its device = devices[0]); `peers`
    // serve devices[1..] and are owned by it.  vt_engine_comm_init_rank: this engine is one rank of a
    // one-process-per-GPU job.  Either way `comm` is this device's RCCL communicator.
    std::vector<vt_engine*> peers;
    vt_engine*  root = nullptr;               // on a peer: its root
    void*       comm = nullptr;               // ncclComm_t
    bool        comm_from_init_all = false;   // made by ncclCommInitAll for the single-process group (not by vt_engine_comm_init_rank)
    int         comm_rank = 0, comm_size = 1;
    hipStream_t s_comm = nullptr;             // the gather of batch b runs here, beside the trace of batch b+1
    hipEvent_t  ev_traced[vt::kMaxGatherChunks] = {};   // trace stream -> comm stream, one per piece of a batch
    uint32_t    gather_chunks = 1;            // option "gather_chunks": pieces a shard is traced and gathered in (vt_trace_closest_gather_dev)
    std::vector<vt::GatherStep> part_steps;   // per-rank form: the steps of the batch whose pieces are being handed over
    int         part_next = 0, part_chunks = 0;
    hipEvent_t  ev_sent[2] = {nullptr, nullptr};   // comm stream: the gather that read send buffer b has completed
    hipEvent_t  ev_g0 = nullptr, ev_g1 = nullptr;  // comm stream, around the latest gather when `timing` is on
    bool        gather_timed = false;
    void*       d_send[2] = {nullptr, nullptr};    // peers: this device's shard of hit records, double-buffered across batches
    size_t      d_send_bytes[2] = {0, 0};
    vt::GatherSchedule sched;                      // batches issued + which sent events exist (group state lives on the root)
};

struct vt_batch_set;

struct vt_scene {
    vt_engine*    engine = nullptr;
    std::vector<vt_batch_set*> open_sets;  // vt_batch_set_begin .. _trace / _abort: detached when the scene is freed
    char*         d_records = nullptr; // pairs, then (128-B aligned) the leaf-ordered triangles
    vt_tri64*     d_tris = nullptr;    // = d_records + tri_base * 64
    uint32_t      tri_base = 0;
    uint32_t*     d_prim_to_slot = nullptr;
    vt_tri_attribs* d_attribs = nullptr;   // optional side table, original triangle order
    vt_tri_frame*   d_frames_bind = nullptr; // optional: per-vertex normals / tangents as handed over (vt_scene_set_tri_frames), and ...
    vt_tri_frame*   d_frames = nullptr;      // ... as the last vt_scene_skin_refit moved them (second half of the same block)
    // refit: pair indices sorted by depth (deepest level first) and where each level starts
    uint32_t*     d_level_pairs = nullptr;
    std::vector<uint32_t> level_begin;     // level_begin[k] .. level_begin[k+1]) = k-th deepest level
    // skinning inputs (vt_scene_set_skin) and the per-frame matrix table
    float*          d_bind_verts = nullptr;
    vt_skin_vertex* d_skin = nullptr;
    uint32_t*       d_matrix_base = nullptr;
    float*          d_skin_mats = nullptr;   // 3 x mats_cap matrices: bones | binds | products
    uint32_t        mats_cap = 0;
    // alpha test: set when a triangle carries VT_TRI_ALPHATEST; materials + alpha planes from vt_scene_set_alpha
    bool               has_alpha = false;
    vt_alpha_material* d_alpha_mats = nullptr;
    uint8_t*           d_alpha_texels = nullptr;
    uint32_t           n_alpha_mats = 0;
    uint64_t           alpha_table_bytes = 0; // what the two tables add to `bytes`
    // ... and what the kernels read: one 64-B AlphaRec per triangle slot behind the triangles in d_records (built from
    // the two side tables by alpha_records_kernel whenever either changes; trace_kernels.h)
    uint32_t           alpha_base = 0;        // record index of slot 0's AlphaRec; 0 = no room reserved yet
    bool               alpha_ready = false;
    size_t             record_capacity = 0;   // 64-B records allocated at d_records
    // refit / skinning with non-finite vertices: NaN boxes pass every slab test, so a poisoned subtree is walked by
    // every ray -- the scene is refused until it has been refitted with finite data
    float           coherent_radius2 = 0.f;   // (2 % of the scene's diagonal)^2: how far apart the origins of a ray packet may lie
    uint32_t*       d_bad = nullptr;
    char*           h_verdict = nullptr;     // pinned: [0] the finite check's count, [64] the root pair behind a refit
    bool            poisoned = false;
    hipGraphExec_t  refit_graph = nullptr;   // the level-by-level refit launches, captured once (launch-bound: ~30 tiny kernels)
    uint32_t      npairs = 0, ntris = 0, max_depth = 0, root_leaf_count = 0;
    uint64_t      bytes = 0;
    vt_upload_stats upload_stats{};         // where the time of this scene's upload went (vt_scene_upload_stats)
    // the host copies of this scene's records (the vt_host_scene it was uploaded from, and / or copies fetched with
    // vt_host_scene_download): a device-side refit marks every one of them stale
    std::vector<std::weak_ptr<std::atomic<int>>> host_copies;
    void add_host_copy(const std::shared_ptr<std::atomic<int>>& flag) { host_copies.emplace_back(flag); }
    void mark_host_copies_stale()
    {
        size_t live = 0;
        for (size_t k = 0; k < host_copies.size(); ++k)
            if (auto f = host_copies[k].lock()) { f->store(1, std::memory_order_release); host_copies[live++] = host_copies[k]; }
        host_copies.resize(live);                // copies that have been freed drop out
    }
    // multi-GPU group: the same scene on every peer device (replicas[g-1] lives on engine->peers[g-1]); owned by this scene
    std::vector<vt_scene*> replicas;
};

// one traced batch, resident on the device until its arrays are asked for (include/vistrace_hip.h)
struct vt_batch {
    vt_engine* engine = nullptr;             // NULL once the engine was closed
    uint64_t   n = 0;
    char*      d_mem = nullptr;              // rays | hits | attrs | shade | tbn
    size_t     d_mem_bytes = 0;
    // host copies, each fetched on first use into pinned memory (taken from / returned to the engine's spare list)
    struct HostArray { void* p = nullptr; size_t bytes = 0; bool have = false; };
    HostArray h_rays, h_hits, h_attrs, h_shade, h_tbn;
    void *d_hits = nullptr, *d_attrs = nullptr, *d_shade = nullptr, *d_tbn = nullptr;
    hipEvent_t done = nullptr;               // behind the last kernel of the batch
    hipEvent_t hits_down = nullptr;          // VT_BATCH_FETCH_HITS: behind the last download of hit records into h_hits
    bool       hits_in_flight = false;       // ... which vt_batch_hits still has to wait for
};

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

namespace vt {

inline void track_scene(vt_engine* e, vt_scene* s) { std::lock_guard<std::mutex> lock(e->launch_mu); e->scenes.push_back(s); }
inline void untrack_scene(vt_engine* e, vt_scene* s)
{
    std::lock_guard<std::mutex> lock(e->launch_mu);
    e->scenes.erase(std::remove(e->scenes.begin(), e->scenes.end(), s), e->scenes.end());
}

// EVERY device / pinned-host allocation of the library goes through these two (tests/test_abi_symbols.py greps for strays): one
// place for the fault injection of include/vistrace_hip.h's "Test hooks" (VT_TEST_FAIL_ALLOC); a plain call otherwise.
inline hipError_t dev_malloc(void** p, size_t bytes)
{
    if (test_alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
inline hipError_t pinned_malloc(void** p, size_t bytes, unsigned flags = hipHostMallocDefault)
{
    if (test_alloc_fails()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(p, bytes, flags);
}

int ensure_bytes(void** ptr, size_t* have, size_t need);
// one batch of a launch: d_out = its vt_hit array (closest hit) or its byte array (any hit); image_width as vt_batch_desc
// d_count (optional, one-batch closest-hit launches on scenes without alpha test): a device word holding how many of the n rays
// really exist -- the launch is sized for n, the kernel reads the word (vt_bounce_loop_dev: no host round trip per depth)
struct BatchReq { const void* d_rays; void* d_out; uint64_t n; uint32_t image_width; const uint32_t* d_count = nullptr; };
// ONE launch over nreq batches on `stream` (per-launch scratch from the engine's slot ring); d_stats: counters kernels, one batch
int launch_batches(vt_scene* s, const BatchReq* reqs, uint32_t nreq, void* d_stats, bool any_hit, bool stats, hipStream_t stream);
// enqueue one trace of n device-resident rays on `stream` (per-launch scratch from the engine's slot ring)
int engine_launch(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* d_occ, void* d_stats, bool any_hit, bool stats,
                  hipStream_t stream);
// engine.hip: (2 % of the scene's diagonal)^2 from the root pair: how far apart the origins of a ray packet may lie
float scene_packet_radius2(const vt_node_pair& root);
// scene_build.hip: prim_to_slot and the level lists of a scene whose records are on the device (h_pair_depth: depth of every pair)
int scene_index_tables(vt_scene* s, const uint32_t* h_pair_depth);
// engine.hip: the scene of a group's root, complete on its device, is copied device to device to every other member
// (s->replicas); `rebuild` uploads it to one member from the host instead -- only used if a peer copy is refused
int scene_replicate(vt_scene* s, const std::function<int(vt_engine*, vt_scene**)>& rebuild);
// batch.hip: the open batch sets of a scene that is being freed lose their scene (their later calls fail, abort still frees them)
void batch_sets_detach(vt_scene* s);
// the host-pointer path of ONE device: staging copies + launch(es) + copy-out, synchronous
int engine_trace_host(vt_scene* s, const vt_ray* rays, uint64_t n, void* out, size_t out_elem, bool any_hit);

// shading.hip
// TraceResult::CalcTBN (no normal map) + CalcFootprint for n hits; the scene holds frames and attribs
hipError_t launch_hit_tbn(vt_scene* s, const void* d_rays, const void* d_hits, uint64_t n, float cone_width, float cone_angle,
                          void* d_out, hipStream_t stream);
// the per-vertex frames follow the bones (no-op without vt_scene_set_tri_frames); d_prod = this frame's bones x binds products
hipError_t skin_frames(vt_scene* s, const float* d_prod, uint32_t nmat, hipStream_t stream);

// multi_gpu.hip
// a host ray array split into contiguous shards, one per device of the scene's group, traced side by side
int multi_trace_host(vt_scene* s, const vt_ray* rays, uint64_t n, void* out, size_t out_elem, bool any_hit);
// communicator, comm stream, events and send buffers of one engine
void multi_release(vt_engine* e);
// host batches below this many rays stay on the root device (a second device's staging pipeline does not pay)
constexpr uint64_t kMultiHostMin = uint64_t(1) << 20;

} // namespace vt
