// engine.hip -- device half of the C ABI in include/vistrace_hip.h.
//
// vt_engine  = one HIP device + one stream + launch configuration + staging buffers.
// vt_scene   = the linearised tree and triangle records resident in that device's HBM
//              (uploaded once per Rebuild, source/objects/AccelStruct.cpp:762-775).
// vt_trace_*_dev = the call at source/objects/AccelStruct.cpp:818, batched, on device-resident rays: launch_batches plans one
//              launch over one or several batches (kernel choice, launch slot, batch table) -- everything else that traces
//              goes through it (batch.hip: host arrays and batch objects; multi_gpu.hip: shards and their gather).
// There is no CPU fallback anywhere in this file: without a usable device every entry
// point returns VT_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "engine_internal.h"

using namespace vt;

namespace vt {

int ensure_bytes(void** ptr, size_t* have, size_t need)
{
    if (*have >= need) return VT_OK;
    if (*ptr) { VT_HIP(hipFree(*ptr)); *ptr = nullptr; *have = 0; }
    size_t cap = std::max(need, size_t(1) << 20);
    VT_HIP(dev_malloc(ptr, cap));
    *have = cap;
    return VT_OK;
}

} // namespace vt

namespace {

struct LaunchPlan {
    bool     persistent;
    bool     fetch_dma;
    uint32_t grid_blocks;
    uint32_t lds_entries;
    size_t   lds_bytes;
    uint32_t ovf_entries;
};

// n = rays of all batches of the launch; static_blocks = 256-ray blocks the one-ray-per-lane kernel would need for them
// camera_rays = every batch came with a usable ray_image_width (camera rays in image order)
int plan_launch(vt_engine* e, const vt_scene* s, uint64_t n, uint64_t static_blocks, bool camera_rays, bool any_hit, bool stats, LaunchPlan& p)
{
    // stack entries a ray can need = inner levels below the root pair
    const uint32_t need = s->max_depth;
    // persistent = 2 (default): small batches (a few rays per resident lane) finish sooner with one
    // ray per lane -- no block cursor, no re-fill, and no end-of-queue tail across the whole grid
    // (the persistent launch carries ~0.22 ms of fixed cost -- grid start, first fill, end-of-queue tail; rays into
    // small trees are cheap enough for one ray per lane to win up to twice the batch size, rays into large trees only
    // up to half of it: scripts/sweep.py tables in profiles/r1/notes.md)
    const uint64_t factor = s->npairs <= 200000u ? uint64_t(e->auto_static_factor) * 2 : (uint64_t(e->auto_static_factor) + 1) / 2;
    p.persistent = e->persistent == 1 || (e->persistent == 2 && n > factor * e->cu_count * 8 * kBlockThreads);
    const uint64_t rec_bytes = uint64_t(s->record_capacity) * 64;
    // Camera rays in image order (the caller said so with ray_image_width) walk in lock step as 4 x 16 pixel tiles and finish
    // together: nothing for the re-fill machinery to win, and one ray per lane is ahead at EVERY batch size while the scene's
    // records stay in the caches (round 4, profiles/r4/auto_policy_sweep.txt: 100 k triangles 0.88 against 1.18 ms for 16 Mi
    // rays, 1 M triangles 0.59 against 0.84 ms for 4 Mi and 6.4 against 6.8 ms for 64 Mi; 10 M triangles -- 1 GB of records,
    // beyond the 256 MiB Infinity Cache -- the other way round: 4.5 against 3.8 ms for 16 Mi)
    if (e->persistent == 2 && camera_rays && rec_bytes <= (uint64_t(256) << 20) && static_blocks <= 0x7FFFFFFFull) p.persistent = false;
    if (any_hit && stats) p.persistent = false;        // the any-hit counters kernel exists as the one-ray-per-lane variant only
    // the DMA-fetch kernel addresses records as base + 32-bit byte offset: scenes below 4 GiB
    p.fetch_dma = p.persistent && e->fetch_dma != 0 && rec_bytes < (uint64_t(1) << 32);
    const uint64_t blocks_for_rays = static_blocks;
    if (p.persistent) {
        p.lds_entries = std::min(std::max(e->lds_entries, 1u), std::max(need, 1u));
        p.ovf_entries = need > p.lds_entries ? need - p.lds_entries : 0;
        // persistent grid = what is resident at once (registers/LDS decide), capped by the option
        int occ = 0;
        VT_HIP(trace_blocks_per_cu(any_hit, stats, true, p.fetch_dma, s->has_alpha, trace_lds_bytes(p.lds_entries, p.fetch_dma), &occ));
        // LDS is granted in 1 280-B granules of the CU's 160 KB, which the occupancy query does not model
        const size_t lds_need = trace_lds_bytes(p.lds_entries, p.fetch_dma);
        const size_t lds_granted = (lds_need + 1279) / 1280 * 1280;
        const uint32_t by_lds = lds_granted ? uint32_t(e->lds_per_cu / lds_granted) : 64u;
        const uint32_t per_cu = std::max(1u, std::min({e->blocks_per_cu, uint32_t(std::max(occ, 1)), std::max(by_lds, 1u)}));
        // with reserved CUs the blocks that land there leave at once; surplus blocks make up for the ones the
        // dispatcher keeps sending to the emptied CUs while the grid is still being placed
        const uint64_t surplus = e->reserved_cus && e->d_reserved ? uint64_t(4) * e->reserved_cus * per_cu : 0;
        p.grid_blocks = uint32_t(std::min<uint64_t>(uint64_t(e->cu_count) * per_cu + surplus, blocks_for_rays + surplus));
    } else {
        if (blocks_for_rays > 0x7FFFFFFFull) return fail(VT_ERR_INVALID_ARG, "too many rays for one launch");
        p.grid_blocks = uint32_t(blocks_for_rays);
        // one ray per lane: short LDS stack + per-lane overflow while the overflow area stays small
        // (<= 256 MiB), else the whole stack in LDS
        p.lds_entries = std::min(std::max(e->lds_entries, 1u), std::max(need, 1u));
        p.ovf_entries = need > p.lds_entries ? need - p.lds_entries : 0;
        if (uint64_t(p.ovf_entries) * blocks_for_rays * kBlockThreads * sizeof(uint32_t) > (uint64_t(e->static_overflow_mb) << 20)) {
            p.lds_entries = std::max(need, 1u);
            p.ovf_entries = 0;
        }
    }
    p.lds_bytes = trace_lds_bytes(p.lds_entries, p.fetch_dma);
    if (p.lds_bytes > e->lds_per_block_max)
        return fail(VT_ERR_STACK, "tree depth " + std::to_string(need) + " needs more LDS stack than one block can hold");
    return VT_OK;
}

} // namespace

namespace vt {

// One launch over one or several batches (vt_trace_*_multi_dev): the ray blocks of all batches are numbered through, so the
// batches share one grid start and one drain.  d_out of a batch = its vt_hit array (closest hit) or its byte array (any hit).
// (BatchReq: engine_internal.h)
int launch_batches(vt_scene* s, const BatchReq* reqs, uint32_t nreq, void* d_stats, bool any_hit, bool stats, hipStream_t stream)
{
    vt_engine* e = s->engine;
    uint64_t n = 0, static_blocks = 0;
    const BatchReq* first = nullptr;                       // the first batch that holds rays: results are addressed from its array
    const size_t out_elem = any_hit ? 1 : sizeof(vt_hit);
    for (uint32_t k = 0; k < nreq; ++k) {
        const BatchReq& r = reqs[k];
        if (r.n == 0) continue;
        if (!r.d_rays || !r.d_out) return fail(VT_ERR_INVALID_ARG, "vt_trace_dev: NULL device buffer");
        if (reinterpret_cast<uintptr_t>(r.d_rays) % 16 != 0 || (!any_hit && reinterpret_cast<uintptr_t>(r.d_out) % 16 != 0))
            return fail(VT_ERR_INVALID_ARG, "vt_trace_dev: ray and hit arrays must be 16-byte aligned");
        if (nreq > 1 && r.n >= (uint64_t(1) << 32)) return fail(VT_ERR_INVALID_ARG, "vt_trace_multi_dev: a batch of a merged launch holds at most 2^32 - 1 rays");
        if (!first) first = &r;
        n += r.n;
        static_blocks += (r.n + kBlockThreads - 1) / kBlockThreads;
    }
    if (n == 0) return VT_OK;
    if (stats && nreq != 1) return fail(VT_ERR_INVALID_ARG, "the counters kernels take one batch per launch");
    if (first->d_count && (nreq != 1 || any_hit || stats || s->has_alpha))
        return fail(VT_ERR_UNSUPPORTED, "a device-side ray count needs a one-batch closest-hit launch on a scene without alpha test");
    if (s->poisoned)
        return fail(VT_ERR_INVALID_ARG, "the scene was last refitted with non-finite vertex positions; refit it with finite data");
    if (s->has_alpha && (!s->d_attribs || !s->d_alpha_mats || !s->alpha_ready))
        return fail(VT_ERR_UNSUPPORTED, "the scene holds alpha-tested triangles (Primitives.h:196-208): call "
                                        "vt_scene_set_tri_attribs and vt_scene_set_alpha before tracing");
    if (s->has_alpha && !e->alpha_regs_checked) {
        bool ok = true;
        VT_HIP(alpha_kernels_within_budget(&ok));
        if (!ok) return fail(VT_ERR_UNSUPPORTED, "the alpha-test kernels of this build exceed their register reservation (see check_isa.py); rebuild the library");
        e->alpha_regs_checked = true;
    }
    // image-order batches: whole bands of 16 rows are taken tile-wise; a block of the persistent kernel is one tile (64 rays) or two
    // side by side (128), so the row length must be a multiple of 4 resp. 8
    auto tile_width = [](const BatchReq& r) -> uint32_t {
        if (r.image_width < 4 || r.image_width % 4 != 0 || r.n >= (uint64_t(1) << 32)) return 0;
        return r.n / (uint64_t(r.image_width) * 16) != 0 ? r.image_width : 0;
    };
    bool any_tiled = false, all_tiled = true, all_mult8 = true;
    for (uint32_t k = 0; k < nreq; ++k) {
        if (reqs[k].n == 0) continue;
        if (tile_width(reqs[k]) != 0) { any_tiled = true; all_mult8 = all_mult8 && reqs[k].image_width % 8 == 0; }
        else all_tiled = false;
    }
    LaunchPlan p;
    int rc = plan_launch(e, s, n, static_blocks, any_tiled && all_tiled, any_hit, stats, p);
    if (rc != VT_OK) return rc;

    // this launch's private scratch: the next slot of the ring (see vt_engine::LaunchSlot)
    std::lock_guard<std::mutex> lock(e->launch_mu);
    vt_engine::LaunchSlot& slot = e->slots[e->next_slot % vt_engine::kLaunchSlots];
    ++e->next_slot;
    const size_t ovf_words = size_t(p.ovf_entries) * p.grid_blocks * kBlockThreads;
    if (ovf_words > slot.overflow_words) {
        // Grow the overflow area (rare: first launches on a deeper tree or with a larger one-ray-per-lane grid).  Only the launch that
        // last used THIS slot has to be over -- nothing else on the device is waited for.  Every other slot that is idle right now
        // and would need the same growth on its next turn gets its area from the same allocation: one hipMalloc instead of one per
        // slot over the next 16 launches (0.2 ms each: profiles/r4/notes.md section 1).  A block is freed when its last slot moves on.
        if (slot.used) VT_HIP(hipEventSynchronize(slot.done));
        std::vector<vt_engine::LaunchSlot*> takers{&slot};
        for (vt_engine::LaunchSlot& other : e->slots)
            if (&other != &slot && other.overflow_words < ovf_words && (!other.used || hipEventQuery(other.done) == hipSuccess)) takers.push_back(&other);
        (void)hipGetLastError();                             // hipEventQuery of a launch in flight reports hipErrorNotReady
        void* block = nullptr;
        if (dev_malloc(&block, takers.size() * ovf_words * sizeof(uint32_t)) != hipSuccess) {   // not enough room for all: this slot alone
            (void)hipGetLastError();
            takers.resize(1);
            VT_HIP(dev_malloc(&block, ovf_words * sizeof(uint32_t)));
        }
        std::shared_ptr<void> owner(block, [](void* p) { (void)hipFree(p); });
        for (size_t k = 0; k < takers.size(); ++k) {
            takers[k]->ovf_block = owner;
            takers[k]->d_overflow = static_cast<uint32_t*>(block) + k * ovf_words;
            takers[k]->overflow_words = ovf_words;
        }
    } else if (slot.used) {
        VT_HIP(hipStreamWaitEvent(stream, slot.done, 0));    // kLaunchSlots launches ago: normally long finished
    }
    uint32_t* const d_cursor = slot.d_ctl;                   // 8 cursors, 64 B apart
    uint32_t* const d_cu_slots = slot.d_ctl + 128;           // 1024 counters
    if (slot.dirty) {
        // the slot's previous launch did not end normally (enqueue error): its cursors may not have been put back to zero by
        // the last wave out, so they are cleared here -- once, in stream order behind the wait above
        VT_HIP(hipMemsetAsync(d_cursor, 0, 512, stream));
        slot.dirty = false;
    }

    TraceArgs a{};
    a.records = s->d_records;
    a.tri_base = s->tri_base;
    a.hits = any_hit ? nullptr : static_cast<vt_hit*>(first->d_out);
    a.occluded = any_hit ? static_cast<uint8_t*>(first->d_out) : nullptr;
    a.ray_stats = static_cast<vt_ray_stats*>(d_stats);
    a.overflow = slot.d_overflow;
    a.block_cursor = d_cursor;
    a.npairs = s->npairs;
    a.root_leaf_count = s->root_leaf_count;
    a.lds_entries = p.lds_entries;
    // rays handed to a wave at a time: the configured size, but never so large that the
    // grid's waves cannot all get several blocks (small batches would leave waves idle)
    {
        const uint64_t waves = uint64_t(p.grid_blocks) * (kBlockThreads / 64);
        uint64_t br = std::max<uint64_t>(64, (e->block_rays / 64u) * 64u);
        const uint64_t fair = (n / (waves * 8) / 64) * 64;
        a.block_rays = uint32_t(std::max<uint64_t>(64, std::min(br, std::max<uint64_t>(fair, 64))));
    }
    a.refill_threshold = std::min(std::max(e->refill_threshold, 1u), 64u);
    a.tri_threshold = std::min(std::max(e->tri_threshold, 1u), 64u);
    a.alpha_threshold = std::min(std::max(e->alpha_threshold, 1u), 64u);
    a.coherent_detect = e->coherent_detect;
    a.coherent_radius2 = s->coherent_radius2;
    a.alpha_base = s->alpha_base;
    a.alpha_texels = s->d_alpha_texels;
    a.reserved_cus = p.persistent && e->reserved_cus ? e->d_reserved : nullptr;
    a.cu_slots = d_cu_slots;
    a.reserved_limit = e->reserved_limit;
    a.live_n = first->d_count;

    // the first block of every wave is static (block w -> wave w); the cursor hands out the rest
    if (a.reserved_cus) VT_HIP(hipMemsetAsync(a.cu_slots, 0, 4096, stream));
    a.xcd_cursors = e->xcd_cursors != 0;
    if (any_tiled) {
        if (a.block_rays > 128 || (a.block_rays == 128 && !all_mult8)) a.block_rays = all_mult8 ? 128 : 64;
        if (a.block_rays != 64 && a.block_rays != 128) a.block_rays = 64;
    }
    const uint32_t unit = p.persistent ? a.block_rays : kBlockThreads;      // rays per ray block of this launch
    auto describe = [&](const BatchReq& r, uint32_t first_block, TraceSeg& sg) -> uint64_t {    // returns the batch's block count
        sg = TraceSeg{};
        sg.rays = static_cast<const vt_ray*>(r.d_rays);
        sg.n = r.n;
        sg.first_block = sg.end_block = first_block;
        if (r.n == 0) return 0;
        sg.out_off = (static_cast<const char*>(r.d_out) - static_cast<const char*>(first->d_out)) / ptrdiff_t(out_elem);
        sg.tile_w = tile_width(r);
        if (sg.tile_w) { const uint64_t band = uint64_t(sg.tile_w) * 16; sg.tiled_rays = r.n / band * band; }
        const uint64_t blocks = (r.n + unit - 1) / unit;
        sg.end_block = first_block + uint32_t(blocks);
        return blocks;
    };
    uint64_t nblocks = 0;
    if (nreq == 1) {
        nblocks = describe(reqs[0], 0, a.seg0);
        if (nblocks > 0xFFFFFFFEull) return fail(VT_ERR_INVALID_ARG, "too many rays for one launch");
    } else {
        // the batch table travels in the slot's pinned block -> its device block, in stream order ahead of the kernel; the
        // pinned block is only rewritten once the launch that last read it is over
        if (slot.segs_cap == 0 && !e->d_segs_all && nreq <= vt_engine::kSlotSegs) {
            // first merged launch of the engine: tables of kSlotSegs batches for ALL slots from one allocation each -- otherwise every
            // slot would allocate its own on its first merged launch (16 launches of ~0.2 ms extra: profiles/r4/notes.md section 1)
            const size_t per = vt_engine::kSlotSegs * sizeof(TraceSeg), all = vt_engine::kLaunchSlots * per;
            VT_HIP(dev_malloc(reinterpret_cast<void**>(&e->d_segs_all), all));
            if (pinned_malloc(reinterpret_cast<void**>(&e->h_segs_all), all) != hipSuccess) {
                (void)hipGetLastError(); (void)hipFree(e->d_segs_all); e->d_segs_all = nullptr;
                return fail(VT_ERR_HIP, "trace launch: no pinned memory for the batch tables");
            }
            for (uint32_t k = 0; k < vt_engine::kLaunchSlots; ++k) {
                vt_engine::LaunchSlot& sl = e->slots[k];
                if (sl.segs_cap != 0) continue;
                sl.d_segs = reinterpret_cast<TraceSeg*>(e->d_segs_all + k * per);
                sl.h_segs = reinterpret_cast<TraceSeg*>(e->h_segs_all + k * per);
                sl.segs_cap = vt_engine::kSlotSegs;
                sl.segs_shared = true;
            }
        }
        if (nreq > slot.segs_cap) {
            if (slot.used) VT_HIP(hipEventSynchronize(slot.done));   // its kernel may still read the device block
            if (slot.d_segs && !slot.segs_shared) VT_HIP(hipFree(slot.d_segs));
            if (slot.h_segs && !slot.segs_shared) VT_HIP(hipHostFree(slot.h_segs));
            slot.d_segs = nullptr; slot.h_segs = nullptr; slot.segs_shared = false;
            slot.segs_cap = 0;
            const size_t cap = std::max<size_t>(64, size_t(nreq) * 2);
            VT_HIP(dev_malloc(reinterpret_cast<void**>(&slot.d_segs), cap * sizeof(TraceSeg)));
            VT_HIP(pinned_malloc(reinterpret_cast<void**>(&slot.h_segs), cap * sizeof(TraceSeg)));
            slot.segs_cap = cap;
        } else if (slot.segs_copied_valid) {
            VT_HIP(hipEventSynchronize(slot.segs_copied));   // the table copy of the merged launch that last used this slot (long over)
        }
        {   // result ranges of one launch must not overlap (they are written in no particular order)
            std::vector<std::pair<const char*, const char*>> ranges;
            for (uint32_t k = 0; k < nreq; ++k)
                if (reqs[k].n != 0) ranges.push_back({static_cast<const char*>(reqs[k].d_out), static_cast<const char*>(reqs[k].d_out) + reqs[k].n * out_elem});
            std::sort(ranges.begin(), ranges.end());
            for (size_t k = 1; k < ranges.size(); ++k)
                if (ranges[k].first < ranges[k - 1].second)
                    return fail(VT_ERR_INVALID_ARG, "vt_trace_multi_dev: the result arrays of two batches overlap");
        }
        for (uint32_t k = 0; k < nreq; ++k) {
            if (reqs[k].n != 0 && (static_cast<const char*>(reqs[k].d_out) - static_cast<const char*>(first->d_out)) % ptrdiff_t(out_elem) != 0)
                return fail(VT_ERR_INVALID_ARG, "vt_trace_multi_dev: hit arrays must be 16-byte aligned");
            nblocks += describe(reqs[k], uint32_t(nblocks), slot.h_segs[k]);
            if (nblocks > 0xFFFFFFFEull) return fail(VT_ERR_INVALID_ARG, "too many rays for one launch");
        }
        a.seg0 = slot.h_segs[0];
        a.segs = slot.d_segs;
        a.nseg = nreq;
        VT_HIP(hipMemcpyAsync(slot.d_segs, slot.h_segs, size_t(nreq) * sizeof(TraceSeg), hipMemcpyHostToDevice, stream));
        if (!slot.segs_copied) VT_HIP(hipEventCreateWithFlags(&slot.segs_copied, hipEventDisableTiming));
        VT_HIP(hipEventRecord(slot.segs_copied, stream));    // from here on the pinned block may be rewritten
        slot.segs_copied_valid = true;
    }
    a.nblocks = uint32_t(nblocks);
    if (!p.persistent) p.grid_blocks = uint32_t(nblocks);
    // 0 = by scene size: cheap rays (small trees) finish fast enough for the single cursor word to become the limit
    a.max_claim = e->max_claim ? e->max_claim : (s->npairs <= 200000u ? 4u : 1u);
    // the cursors are zero: engine open cleared them and every persistent launch leaves them so (leave_grid)
    a.cursor_base = (a.reserved_cus || a.xcd_cursors) ? 0u : p.grid_blocks * (kBlockThreads / 64);
    if (e->timing) VT_HIP(hipEventRecord(e->ev_start, stream));
    const hipError_t lerr = launch_trace(a, any_hit, stats, p.persistent, p.fetch_dma, s->has_alpha, p.grid_blocks, p.lds_bytes, stream);
    if (lerr != hipSuccess) {
        slot.dirty = true;                                   // whatever did run may have left cursors behind
        return fail(VT_ERR_HIP, std::string("trace launch: ") + hipGetErrorString(lerr));
    }
    if (e->timing) { VT_HIP(hipEventRecord(e->ev_stop, stream)); e->ev_valid = true; }
    VT_HIP(hipEventRecord(slot.done, stream));
    slot.used = true;
    e->last_blocks = p.grid_blocks; e->last_threads = kBlockThreads; e->last_lds = uint32_t(p.lds_bytes);
    e->last_persistent = p.persistent; e->last_dma = p.fetch_dma;
    return VT_OK;
}

} // namespace vt

namespace {

int launch(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* d_occ, void* d_stats, bool any_hit,
           bool stats, hipStream_t stream)
{
    if (n == 0) return VT_OK;
    const BatchReq one{d_rays, any_hit ? d_occ : d_hits, n, s->engine->ray_image_width};
    return launch_batches(s, &one, 1, d_stats, any_hit, stats, stream);
}

// Pick `want` CUs, spread over the XCDs, that the persistent grid will leave empty.  The CUs are named by
// __smid() (xcc, se, cu): a probe launch records which values exist on this part (harvested CUs differ
// from device to device), the choice goes to the device as a 1024-bit set.  Returns the number reserved
// in e->reserved_cus (0 switches the feature off).
int reserve_cus(vt_engine* e, uint32_t want)
{
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "reserved_cus: hipSetDevice failed");
    VT_HIP(hipDeviceSynchronize());                       // no launch may be reading the set while it changes
    e->reserved_cus = 0;
    if (want == 0) return VT_OK;
    if (!e->d_reserved) VT_HIP(dev_malloc(reinterpret_cast<void**>(&e->d_reserved), 128 + 4096));
    uint32_t seen[32] = {};
    for (int round = 0; round < 4; ++round) {             // 2 probe blocks fit a CU: a few rounds reach all of them
        VT_HIP(hipMemsetAsync(e->d_reserved, 0, 128, e->stream));
        VT_HIP(launch_cu_probe(e->d_reserved, uint32_t(e->cu_count) * 4u, e->stream));
        uint32_t got[32];
        VT_HIP(hipMemcpyAsync(got, e->d_reserved, 128, hipMemcpyDeviceToHost, e->stream));
        VT_HIP(hipStreamSynchronize(e->stream));
        for (int w = 0; w < 32; ++w) seen[w] |= got[w];
    }
    // one list per shader engine of every XCD (id = xcc[9:6] se[5:4] cu[3:0]); take the last CU of each list
    // in turn: a first pass gives every (XCD, SE) one reserved CU, further passes a second one, ...
    std::vector<std::vector<uint32_t>> per_se(64);
    for (uint32_t id = 0; id < 1024; ++id)
        if ((seen[id >> 5] >> (id & 31u)) & 1u) per_se[id >> 4].push_back(id);
    uint32_t chosen[32] = {};
    uint32_t n = 0;
    for (bool any = true; any && n < want;) {
        any = false;
        for (auto& v : per_se) {
            if (v.size() > 1 && n < want) {               // never the last CU of a shader engine
                const uint32_t id = v.back();
                v.pop_back();
                chosen[id >> 5] |= 1u << (id & 31u);
                ++n;
                any = true;
            }
        }
    }
    VT_HIP(hipMemcpyAsync(e->d_reserved, chosen, 128, hipMemcpyHostToDevice, e->stream));
    VT_HIP(hipStreamSynchronize(e->stream));
    e->reserved_cus = n;
    return VT_OK;
}

// (2 % of the scene's diagonal)^2 from the root's two children: how far apart the origins of a ray packet may lie
float packet_radius2(const vt_node_pair& root)
{
    float d2 = 0.f;
    for (int k = 0; k < 3; ++k) {
        const vt_bvh_node& l = root.child[0];
        const vt_bvh_node& r = root.child[1];
        const float lo = std::min(l.bounds[2 * k], r.bounds[2 * k]), hi = std::max(l.bounds[2 * k + 1], r.bounds[2 * k + 1]);
        const float ext = hi - lo;
        if (ext == ext && ext > 0.f && ext < 1e18f) d2 += ext * ext;
    }
    return 0.02f * 0.02f * d2;
}

// The AlphaRecs of a scene with alpha-tested triangles (trace_kernels.h): room for them behind the triangles (the record
// array is re-allocated once if the scene was uploaded without any flagged triangle), then one kernel over the slots.
// The engine's device is current and idle (the callers synchronise first).
// ensure_alpha_room: one AlphaRec per triangle slot behind the triangles; when it fails nothing has changed.
int ensure_alpha_room(vt_scene* s)
{
    if (!s->has_alpha || s->ntris == 0 || s->alpha_base != 0) return VT_OK;
    vt_engine* e = s->engine;
    {
        const uint32_t base = (s->tri_base + s->ntris + 1u) & ~1u;
        const size_t cap = size_t(base) + s->ntris;
        char* grown = nullptr;
        VT_HIP(dev_malloc(reinterpret_cast<void**>(&grown), cap * 64));
        hipError_t err = VT_TRY(hipMemset(grown, 0, cap * 64));
        if (err == hipSuccess) err = VT_TRY(hipMemcpy(grown, s->d_records, s->record_capacity * 64, hipMemcpyDeviceToDevice));
        if (err != hipSuccess) { (void)hipFree(grown); return fail(VT_ERR_HIP, std::string("alpha records: ") + hipGetErrorString(err)); }
        {   // launches read these under launch_mu; the device is idle (the callers synchronised), so the old block can go
            std::lock_guard<std::mutex> swap_lock(e->launch_mu);
            (void)hipFree(s->d_records);
            s->bytes += (cap - s->record_capacity) * 64;
            s->d_records = grown;
            s->d_tris = reinterpret_cast<vt_tri64*>(grown + size_t(s->tri_base) * 64);
            s->record_capacity = cap;
            s->alpha_base = base;
        }
        if (s->refit_graph) { (void)hipGraphExecDestroy(s->refit_graph); s->refit_graph = nullptr; }   // it captured the old pointers
    }
    return VT_OK;
}

int build_alpha_records(vt_scene* s)
{
    if (!s->has_alpha || !s->d_attribs || !s->d_alpha_mats || s->ntris == 0) { s->alpha_ready = false; return VT_OK; }
    vt_engine* e = s->engine;
    const int room = ensure_alpha_room(s);       // (a scene without room has no AlphaRecs yet: alpha_ready is false already)
    if (room != VT_OK) return room;
    s->alpha_ready = false;
    AlphaRecArgs a{s->d_tris, s->d_attribs, s->d_alpha_mats, s->n_alpha_mats,
                   reinterpret_cast<AlphaRec*>(s->d_records + size_t(s->alpha_base) * 64), s->ntris};
    VT_HIP(launch_alpha_records(a, e->stream));
    VT_HIP(hipStreamSynchronize(e->stream));
    s->alpha_ready = true;
    return VT_OK;
}

// a pinned host block of at least `need` bytes from the engine's spare list, or a new one
long env_long(const char* name, long dflt)
{
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return std::strtol(v, nullptr, 10);
}

} // namespace

extern "C" {

int vt_device_count(int* count)
{
    if (!count) return fail(VT_ERR_INVALID_ARG, "vt_device_count: NULL");
    *count = 0;
    VT_HIP(hipGetDeviceCount(count));
    return VT_OK;
}

// what the fault-injection hook on checked HIP calls runs before it reports its failure (vt_internal.h: test_hip_fails)
static thread_local bool t_capturing = false;     // this thread is between hipStreamBeginCapture and hipStreamEndCapture
static void drain_every_device()
{
    if (t_capturing) return;                       // (a synchronisation would invalidate the capture: nothing it records runs yet)
    int count = 0, before = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || hipGetDevice(&before) != hipSuccess) return;
    for (int d = 0; d < count; ++d)
        if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
    (void)hipSetDevice(before);
}

int vt_engine_open(int device, vt_engine** out)
{
    if (!out) return fail(VT_ERR_INVALID_ARG, "vt_engine_open: out is NULL");
    *out = nullptr;
    set_test_drain(&drain_every_device);
    int count = 0;
    VT_HIP(hipGetDeviceCount(&count));
    if (count <= 0) return fail(VT_ERR_HIP, "vt_engine_open: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= count) return fail(VT_ERR_INVALID_ARG, "vt_engine_open: bad device index");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_engine_open: hipSetDevice failed");

    hipDeviceProp_t prop;
    VT_HIP(hipGetDeviceProperties(&prop, device));
    vt_engine* e = new vt_engine();
    e->device = device;
    e->cu_count = prop.multiProcessorCount;
    e->lds_per_block_max = prop.sharedMemPerBlock; // 64 KiB default window; plenty for the stack
    if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) e->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    e->persistent = int(env_long("VT_PERSISTENT", e->persistent));
    e->lds_entries = uint32_t(env_long("VT_LDS_ENTRIES", e->lds_entries));
    e->blocks_per_cu = uint32_t(env_long("VT_BLOCKS_PER_CU", e->blocks_per_cu));
    e->block_rays = uint32_t(env_long("VT_BLOCK_RAYS", e->block_rays));
    e->refill_threshold = uint32_t(env_long("VT_REFILL_THRESHOLD", e->refill_threshold));
    e->tri_threshold = uint32_t(env_long("VT_TRI_THRESHOLD", e->tri_threshold));
    e->fetch_dma = int(env_long("VT_FETCH_DMA", e->fetch_dma));
    hipError_t err = VT_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    if (err == hipSuccess) err = dev_malloc(reinterpret_cast<void**>(&e->d_slot_ctl), vt_engine::kLaunchSlots * vt_engine::kSlotCtlBytes);
    if (err == hipSuccess) err = VT_TRY(hipMemset(e->d_slot_ctl, 0, vt_engine::kLaunchSlots * vt_engine::kSlotCtlBytes));
    for (uint32_t k = 0; k < vt_engine::kLaunchSlots && err == hipSuccess; ++k) {
        e->slots[k].d_ctl = reinterpret_cast<uint32_t*>(e->d_slot_ctl + k * vt_engine::kSlotCtlBytes);
        err = VT_TRY(hipEventCreateWithFlags(&e->slots[k].done, hipEventDisableTiming));
    }
    if (err == hipSuccess) err = VT_TRY(hipEventCreateWithFlags(&e->ev_loop, hipEventDisableTiming));
    if (err == hipSuccess) err = pinned_malloc(reinterpret_cast<void**>(&e->h_tiny_rays), vt_engine::kTinyRays * sizeof(vt_ray), hipHostMallocMapped);
    if (err == hipSuccess) err = pinned_malloc(reinterpret_cast<void**>(&e->h_tiny_out), vt_engine::kTinyRays * sizeof(vt_hit), hipHostMallocMapped);
    if (err == hipSuccess) err = VT_TRY(hipHostGetDevicePointer(&e->d_tiny_rays, e->h_tiny_rays, 0));
    if (err == hipSuccess) err = VT_TRY(hipHostGetDevicePointer(&e->d_tiny_out, e->h_tiny_out, 0));
    if (err == hipSuccess) err = VT_TRY(hipEventCreate(&e->ev_start));
    if (err == hipSuccess) err = VT_TRY(hipEventCreate(&e->ev_stop));
    if (err != hipSuccess) {
        vt_engine_close(e);
        return fail(VT_ERR_HIP, std::string("vt_engine_open: ") + hipGetErrorString(err));
    }
    *out = e;
    return VT_OK;
}

static void release_scene_device(vt_scene* s);

void vt_engine_close(vt_engine* e)
{
    if (!e) return;
    for (vt_engine* p : e->peers) vt_engine_close(p);   // a group's root owns the engines of the other devices
    e->peers.clear();
    DeviceGuard guard(e->device);
    (void)hipDeviceSynchronize();               // launches on caller streams may still use the cursor / overflow areas
    multi_release(e);
    std::vector<vt_scene*> left;
    { std::lock_guard<std::mutex> lock(e->launch_mu); left.swap(e->scenes); }
    for (vt_scene* sc : left) {                 // scenes that outlive their engine become inert shells
        release_scene_device(sc);
        sc->engine = nullptr;
    }
    for (vt_batch* b : e->batches) {            // batches that outlive their engine keep what they have downloaded
        if (b->d_mem) (void)hipFree(b->d_mem);
        if (b->done) (void)hipEventDestroy(b->done);
        if (b->hits_down) (void)hipEventDestroy(b->hits_down);
        if (b->hits_in_flight) { b->hits_in_flight = false; b->h_hits.have = true; }   // the device was synchronised above: they have arrived
        b->d_mem = nullptr; b->done = nullptr; b->hits_down = nullptr; b->engine = nullptr;
    }
    e->batches.clear();
    for (auto& ds : e->device_spare) (void)hipFree(ds.first);
    e->device_spare.clear();
    for (auto& ps : e->pinned_spare) (void)hipHostFree(ps.first);
    e->pinned_spare.clear();
    for (vt_engine::LaunchSlot& sl : e->slots) {
        sl.ovf_block.reset(); sl.d_overflow = nullptr;
        if (sl.d_segs && !sl.segs_shared) (void)hipFree(sl.d_segs);
        if (sl.h_segs && !sl.segs_shared) (void)hipHostFree(sl.h_segs);
        if (sl.segs_copied) (void)hipEventDestroy(sl.segs_copied);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (e->d_segs_all) (void)hipFree(e->d_segs_all);
    if (e->h_segs_all) (void)hipHostFree(e->h_segs_all);
    if (e->d_slot_ctl) (void)hipFree(e->d_slot_ctl);
    if (e->ev_loop) (void)hipEventDestroy(e->ev_loop);
    if (e->ev_staged) (void)hipEventDestroy(e->ev_staged);
    if (e->d_rays) (void)hipFree(e->d_rays);
    if (e->d_out) (void)hipFree(e->d_out);
    if (e->d_loop) (void)hipFree(e->d_loop);
    if (e->d_build) (void)hipFree(e->d_build);
    if (e->h_build) (void)hipHostFree(e->h_build);
    if (e->d_reserved) (void)hipFree(e->d_reserved);
    if (e->h_live) (void)hipHostFree(e->h_live);
    if (e->h_bad) (void)hipHostFree(e->h_bad);
    for (int k = 0; k < vt_engine::kStageBufs; ++k) {
        if (e->h_stage_in[k]) (void)hipHostFree(e->h_stage_in[k]);
        if (e->h_stage_out[k]) (void)hipHostFree(e->h_stage_out[k]);
        if (e->ev_in[k]) (void)hipEventDestroy(e->ev_in[k]);
        if (e->ev_k[k]) (void)hipEventDestroy(e->ev_k[k]);
        if (e->ev_out[k]) (void)hipEventDestroy(e->ev_out[k]);
    }
    if (e->s_in) (void)hipStreamDestroy(e->s_in);
    if (e->s_out) (void)hipStreamDestroy(e->s_out);
    if (e->h_tiny_rays) (void)hipHostFree(e->h_tiny_rays);
    if (e->h_tiny_out) (void)hipHostFree(e->h_tiny_out);
    if (e->ev_start) (void)hipEventDestroy(e->ev_start);
    if (e->ev_stop) (void)hipEventDestroy(e->ev_stop);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int vt_engine_set_option(vt_engine* e, const char* key, int64_t value)
{
    if (!e || !key) return fail(VT_ERR_INVALID_ARG, "vt_engine_set_option: NULL");
    for (vt_engine* p : e->peers) { const int rc = vt_engine_set_option(p, key, value); if (rc != VT_OK) return rc; }
    const std::string k(key);
    if (k == "persistent" && value >= 0 && value <= 2) e->persistent = int(value);
    else if (k == "auto_static_factor" && value >= 0 && value <= 1024) e->auto_static_factor = uint32_t(value);
    else if (k == "coherent_detect") e->coherent_detect = value != 0;
    else if (k == "static_overflow_mb" && value >= 0 && value <= 65536) e->static_overflow_mb = uint32_t(value);
    else if (k == "lds_entries" && value >= 1 && value <= 4096) e->lds_entries = uint32_t(value);
    else if (k == "blocks_per_cu" && value >= 1 && value <= 64) e->blocks_per_cu = uint32_t(value);
    else if (k == "block_rays" && value >= 64 && value <= (1 << 24)) e->block_rays = uint32_t(value);
    else if (k == "refill_threshold" && value >= 1 && value <= 64) e->refill_threshold = uint32_t(value);
    else if (k == "tri_threshold" && value >= 1 && value <= 64) e->tri_threshold = uint32_t(value);
    else if (k == "alpha_threshold" && value >= 1 && value <= 64) e->alpha_threshold = uint32_t(value);
    else if (k == "fetch_dma") e->fetch_dma = value != 0;
    else if (k == "spin_wait") e->spin_wait = value != 0;
    else if (k == "xcd_cursors") e->xcd_cursors = value != 0;
    else if (k == "max_claim" && value >= 0 && value <= 1024) e->max_claim = uint32_t(value);
    else if (k == "ray_image_width" && value >= 0) e->ray_image_width = uint32_t(value);
    else if (k == "reserved_cus" && value >= 0 && value <= e->cu_count / 2) return reserve_cus(e, uint32_t(value));
    else if (k == "reserved_limit" && value >= 0 && value <= 64) e->reserved_limit = uint32_t(value);
    else if (k == "gather_overlap") e->sched.overlap = value != 0;
    else if (k == "gather_chunks" && value >= 1 && value <= vt::kMaxGatherChunks) e->gather_chunks = uint32_t(value);
    else return fail(VT_ERR_INVALID_ARG, "vt_engine_set_option: unknown key or value out of range: " + k);
    return VT_OK;
}

int vt_engine_get_option(vt_engine* e, const char* key, int64_t* value)
{
    if (!e || !key || !value) return fail(VT_ERR_INVALID_ARG, "vt_engine_get_option: NULL");
    const std::string k(key);
    if (k == "persistent") *value = e->persistent;
    else if (k == "lds_entries") *value = e->lds_entries;
    else if (k == "blocks_per_cu") *value = e->blocks_per_cu;
    else if (k == "block_rays") *value = e->block_rays;
    else if (k == "refill_threshold") *value = e->refill_threshold;
    else if (k == "auto_static_factor") *value = e->auto_static_factor;
    else if (k == "static_overflow_mb") *value = e->static_overflow_mb;
    else if (k == "coherent_detect") *value = e->coherent_detect;
    else if (k == "tri_threshold") *value = e->tri_threshold;
    else if (k == "alpha_threshold") *value = e->alpha_threshold;
    else if (k == "fetch_dma") *value = e->fetch_dma;
    else if (k == "spin_wait") *value = e->spin_wait;
    else if (k == "xcd_cursors") *value = e->xcd_cursors;
    else if (k == "max_claim") *value = e->max_claim;
    else if (k == "ray_image_width") *value = int(e->ray_image_width);
    else if (k == "reserved_cus") *value = e->reserved_cus;
    else if (k == "reserved_limit") *value = e->reserved_limit;
    else if (k == "gather_overlap") *value = e->sched.overlap;
    else if (k == "gather_chunks") *value = e->gather_chunks;
    else if (k == "cu_count") *value = e->cu_count;
    else if (k == "last_persistent") *value = e->last_persistent;
    else if (k == "last_fetch_dma") *value = e->last_dma;
    else if (k == "device") *value = e->device;
    else if (k == "device_count") *value = int64_t(e->peers.size()) + 1;
    // the latest vt_scene_refit / vt_scene_skin_refit of a scene of this (root) engine: members it went to, and how many of them
    // were waited for before the last member's work had been enqueued (0 by construction; tests/fake_group_check.py)
    else if (k == "last_update_members") *value = e->last_update_members;
    else if (k == "last_update_early_waits") *value = e->last_update_early_waits;
    // ... and its host time: prepare + enqueue of every member (the host is busy), then the waits (one member's device time when
    // every member has a device of its own)
    else if (k == "last_update_enqueue_us") *value = e->last_update_enqueue_us;
    else if (k == "last_update_wait_us") *value = e->last_update_wait_us;
    else return fail(VT_ERR_INVALID_ARG, "vt_engine_get_option: unknown key: " + k);
    return VT_OK;
}

int vt_scene_upload(vt_engine* e, const vt_host_scene* hsw, vt_scene** out)
{
    if (!e || !hsw || !out) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload: NULL argument");
    *out = nullptr;
    const HostScene& hs = hsw->hs;
    // Primitives.h:196-208: alpha-tested triangles need vt_scene_set_alpha before tracing (the flag scan is vt_scene_linearise's,
    // kept current by vt_host_scene_sync)
    const bool has_alpha = hs.has_alpha;
    if (hs.pair_depth.size() != hs.pairs.size()) return fail(VT_ERR_INVALID_ARG, "vt_scene_upload: the host scene lacks its pair depths");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_upload: hipSetDevice failed");

    vt_scene* s = new vt_scene();
    s->engine = e;
    s->add_host_copy(hsw->stale);
    s->has_alpha = has_alpha;
    if (!hs.pairs.empty()) s->coherent_radius2 = packet_radius2(hs.pairs[0]);
    s->npairs = uint32_t(hs.pairs.size());
    s->ntris = uint32_t(hs.tris.size());
    s->max_depth = hs.max_depth;
    s->root_leaf_count = hs.root_leaf_count;

    // Record layout: pairs first, triangles behind them on a 128-B boundary.
    const auto t_begin = std::chrono::steady_clock::now();
    hipError_t err = hipSuccess;
    // (test hook, dead without VT_ENABLE_TEST_HOOKS=1: VT_TEST_RECORD_GAP=<records> leaves that many unused records between the
    // pairs and the triangles, so that a small scene has triangle and AlphaRec records beyond 4 GiB -- the 64-bit addressing and
    // the kernel choice of scenes with more than 67 M records, without building one: tests/test_gpu_parity.py)
    uint64_t gap = 0;
    if (const char* env = test_hook("VT_TEST_RECORD_GAP")) gap = std::strtoull(env, nullptr, 10) & ~uint64_t(1);
    if (uint64_t(s->npairs) + gap + 2 * uint64_t(s->ntris) + 4 >= 0xFFFFFFFFull) { delete s; return fail(VT_ERR_INVALID_ARG, "vt_scene_upload: scene too large"); }
    {
        s->tri_base = ((s->npairs + 1u) & ~1u) + uint32_t(gap);
        const size_t pair_bytes = hs.pairs.size() * sizeof(vt_node_pair);
        const size_t tri_off = size_t(s->tri_base) * 64, tri_bytes = hs.tris.size() * sizeof(vt_tri64);
        // a scene with alpha-tested triangles keeps one AlphaRec per triangle slot behind the triangles (filled once
        // vt_scene_set_tri_attribs and vt_scene_set_alpha have both been called)
        if (has_alpha) s->alpha_base = (s->tri_base + s->ntris + 1u) & ~1u;
        s->record_capacity = std::max<size_t>(has_alpha ? size_t(s->alpha_base) + s->ntris : size_t(s->tri_base) + s->ntris, 2);
        // never empty: idle lanes of the DMA-fetch kernel read record 0, so it must exist (zeros for an empty scene)
        const size_t rec_bytes = s->record_capacity * 64;
        err = dev_malloc(reinterpret_cast<void**>(&s->d_records), rec_bytes);
        // zeros only where no copy lands: the padding between pairs and triangles (+ the test gap), the AlphaRec room
        if (err == hipSuccess && tri_off > pair_bytes) err = VT_TRY(hipMemsetAsync(s->d_records + pair_bytes, 0, tri_off - pair_bytes, e->stream));
        if (err == hipSuccess && rec_bytes > tri_off + tri_bytes) err = VT_TRY(hipMemsetAsync(s->d_records + tri_off + tri_bytes, 0, rec_bytes - tri_off - tri_bytes, e->stream));
        if (err == hipSuccess && pair_bytes) err = VT_TRY(hipMemcpyAsync(s->d_records, hs.pairs.data(), pair_bytes, hipMemcpyHostToDevice, e->stream));
        if (err == hipSuccess && tri_bytes) err = VT_TRY(hipMemcpyAsync(s->d_records + tri_off, hs.tris.data(), tri_bytes, hipMemcpyHostToDevice, e->stream));
        s->d_tris = reinterpret_cast<vt_tri64*>(s->d_records + tri_off);
        s->bytes += rec_bytes;
        s->upload_stats.bytes_h2d = pair_bytes + tri_bytes + hs.pair_depth.size() * 4;
    }
    s->upload_stats.copy_ms = float(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    if (err != hipSuccess) {
        track_scene(e, s);
        vt_scene_free(s);
        return fail(VT_ERR_HIP, std::string("vt_scene_upload: ") + hipGetErrorString(err));
    }
    track_scene(e, s);
    {
        // triangle -> slot (refits) and the pairs by depth, deepest first (level-wise refit): built on the device from the
        // records just uploaded and the pairs' depths (scene_build.hip) -- round 4 made both on the host, serially
        std::lock_guard<std::mutex> host_lock(e->host_mu);
        const int rc = scene_index_tables(s, hs.pair_depth.data());
        if (rc != VT_OK) { vt_scene_free(s); return rc; }
    }
    s->upload_stats.total_ms = float(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    s->upload_stats.device_ms = s->upload_stats.total_ms - s->upload_stats.copy_ms;
    // a group's scene lives on every device: the BVH is replicated, rays are what is sharded
    {
        const int rc = scene_replicate(s, [&](vt_engine* p, vt_scene** rep) { return vt_scene_upload(p, hsw, rep); });
        if (rc != VT_OK) { vt_scene_free(s); return rc; }
    }
    *out = s;
    return VT_OK;
}

// frees a scene's device memory (its engine's device is current); the shell stays
static void release_scene_device(vt_scene* s)
{
    void** bufs[] = {reinterpret_cast<void**>(&s->d_records), reinterpret_cast<void**>(&s->d_prim_to_slot),
                     reinterpret_cast<void**>(&s->d_attribs), reinterpret_cast<void**>(&s->d_level_pairs),
                     reinterpret_cast<void**>(&s->d_alpha_mats), reinterpret_cast<void**>(&s->d_alpha_texels),
                     reinterpret_cast<void**>(&s->d_bind_verts), reinterpret_cast<void**>(&s->d_skin),
                     reinterpret_cast<void**>(&s->d_matrix_base), reinterpret_cast<void**>(&s->d_skin_mats),
                     reinterpret_cast<void**>(&s->d_bad), reinterpret_cast<void**>(&s->d_frames_bind)};
    for (void** b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    if (s->refit_graph) (void)hipGraphExecDestroy(s->refit_graph);
    s->refit_graph = nullptr;
    if (s->h_verdict) (void)hipHostFree(s->h_verdict);
    s->h_verdict = nullptr;
    s->d_tris = nullptr;
    s->d_frames = nullptr;
}

void vt_scene_free(vt_scene* s)
{
    if (!s) return;
    batch_sets_detach(s);
    for (vt_scene* rep : s->replicas) vt_scene_free(rep);
    s->replicas.clear();
    if (vt_engine* e = s->engine) {             // NULL once the engine was closed: only the shell is left
        DeviceGuard guard(e->device);
        (void)hipDeviceSynchronize();           // traces of this scene may be in flight on caller streams
        release_scene_device(s);
        untrack_scene(e, s);
    }
    delete s;
}

uint64_t vt_scene_device_bytes(const vt_scene* s) { return s ? s->bytes : 0; }

static int trace_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* d_occ, void* d_stats, bool any_hit,
                     bool stats, void* stream)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_trace_dev: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_trace_dev: the scene\'s engine has been closed");
    if (n == 0) return VT_OK;
    if (!d_rays || (!d_hits && !d_occ)) return fail(VT_ERR_INVALID_ARG, "vt_trace_dev: NULL device buffer");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_trace_dev: hipSetDevice failed");
    return launch(s, d_rays, n, d_hits, d_occ, d_stats, any_hit, stats, static_cast<hipStream_t>(stream));
}

int vt_trace_closest_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* stream)
{
    return trace_dev(s, d_rays, n, d_hits, nullptr, nullptr, false, false, stream);
}

int vt_trace_any_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_occluded, void* stream)
{
    return trace_dev(s, d_rays, n, nullptr, d_occluded, nullptr, true, false, stream);
}

static int trace_multi_dev(vt_scene* s, const vt_batch_desc* batches, uint32_t nbatches, bool any_hit, void* stream, const char* who)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, std::string(who) + ": scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, std::string(who) + ": the scene's engine has been closed");
    if (nbatches == 0) return VT_OK;
    if (!batches) return fail(VT_ERR_INVALID_ARG, std::string(who) + ": batches is NULL");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, std::string(who) + ": hipSetDevice failed");
    std::vector<BatchReq> reqs(nbatches);
    for (uint32_t k = 0; k < nbatches; ++k) {
        if (batches[k].reserved != 0) return fail(VT_ERR_INVALID_ARG, std::string(who) + ": vt_batch_desc::reserved must be 0");
        reqs[k] = BatchReq{batches[k].d_rays, batches[k].d_out, batches[k].n, batches[k].ray_image_width};
    }
    return launch_batches(s, reqs.data(), nbatches, nullptr, any_hit, false, static_cast<hipStream_t>(stream));
}

int vt_trace_closest_multi_dev(vt_scene* s, const vt_batch_desc* batches, uint32_t nbatches, void* stream)
{
    return trace_multi_dev(s, batches, nbatches, false, stream, "vt_trace_closest_multi_dev");
}

int vt_trace_any_multi_dev(vt_scene* s, const vt_batch_desc* batches, uint32_t nbatches, void* stream)
{
    return trace_multi_dev(s, batches, nbatches, true, stream, "vt_trace_any_multi_dev");
}

int vt_trace_stats_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* d_ray_stats, void* stream)
{
    if (!d_ray_stats) return fail(VT_ERR_INVALID_ARG, "vt_trace_stats_dev: d_ray_stats is NULL");
    return trace_dev(s, d_rays, n, d_hits, nullptr, d_ray_stats, false, true, stream);
}

int vt_trace_any_stats_dev(vt_scene* s, const void* d_rays, uint64_t n, void* d_occluded, void* d_ray_stats, void* stream)
{
    if (!d_ray_stats) return fail(VT_ERR_INVALID_ARG, "vt_trace_any_stats_dev: d_ray_stats is NULL");
    return trace_dev(s, d_rays, n, nullptr, d_occluded, d_ray_stats, true, true, stream);
}

int vt_hit_attrs_dev(vt_scene* s, const void* d_rays, const void* d_hits, uint64_t n, void* d_attrs, void* stream)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_hit_attrs_dev: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_hit_attrs_dev: the scene\'s engine has been closed");
    if (n == 0) return VT_OK;
    if (!d_rays || !d_hits || !d_attrs) return fail(VT_ERR_INVALID_ARG, "vt_hit_attrs_dev: NULL device buffer");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_hit_attrs_dev: hipSetDevice failed");
    HitAttrsArgs a{};
    a.tris = s->d_tris;
    a.prim_to_slot = s->d_prim_to_slot;
    a.rays = static_cast<const vt_ray*>(d_rays);
    a.hits = static_cast<const vt_hit*>(d_hits);
    a.attrs = static_cast<vt_hit_attrs*>(d_attrs);
    a.n = n;
    VT_HIP(launch_hit_attrs(a, static_cast<hipStream_t>(stream)));
    return VT_OK;
}

// The reference's bounce loop is the script's: trace, shade, vistrace.CalcRayOrigin + a cosine-hemisphere direction, trace again
// (source/VisTrace.cpp:1478-1519, source/libraries/BSDF.cpp:69-77), one Lua call per ray.  Here every depth is a trace over the
// queue of live paths and a queue step (scatter the hits to their paths, emit the next queue in path order), all of it enqueued
// WITHOUT a host round trip: the number of live paths stays on the device -- the traces and queue steps behind depth 0 are sized for
// n and read the real count from the word the previous queue step wrote (trace_kernel_devn, QueueArgs::m_dev).  Round 5 read the
// count back and waited for it once per depth.  live_out is written by a host function in stream order: valid once the stream has
// passed this call.  (Scenes with alpha-tested triangles take the older form below: their kernels have no device-count variant.)
namespace {
struct LiveCopy { const uint32_t* src; uint64_t* dst; uint32_t count; };
void copy_live_counts(void* p)
{
    LiveCopy* c = static_cast<LiveCopy*>(p);
    for (uint32_t k = 0; k < c->count; ++k) c->dst[k] = c->src[k];
    delete c;
}
} // namespace

int vt_bounce_loop_dev(vt_scene* s, const void* d_rays, uint64_t n, uint32_t depth, uint64_t seed, void* d_hits,
                       uint64_t* live_out, void* stream_)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_bounce_loop_dev: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_bounce_loop_dev: the scene\'s engine has been closed");
    if (n == 0 || depth == 0) return VT_OK;
    if (!d_rays || !d_hits) return fail(VT_ERR_INVALID_ARG, "vt_bounce_loop_dev: NULL device buffer");
    if (n >= (uint64_t(1) << 32)) return fail(VT_ERR_INVALID_ARG, "vt_bounce_loop_dev: more than 2^32-1 paths");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_bounce_loop_dev: hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint64_t nblocks = (n + kBlockThreads - 1) / kBlockThreads;
    auto al = [](uint64_t b) { return (b + 255) & ~uint64_t(255); };
    const uint64_t ray_b = al(n * sizeof(vt_ray)), id_b = al(n * 4), hit_b = al(n * sizeof(vt_hit)), off_b = al(nblocks * 4);
    const uint64_t live_b = al((uint64_t(depth) + 1) * 4);
    // the queues are engine-wide scratch: a loop on another stream may still be using them
    std::lock_guard<std::mutex> loop_lock(e->host_mu);
    const size_t loop_need = 2 * ray_b + 2 * id_b + hit_b + off_b + live_b;
    if (e->loop_used) {
        if (loop_need > e->d_loop_bytes || live_b > e->h_live_bytes) VT_HIP(hipEventSynchronize(e->ev_loop));     // about to be freed and re-allocated
        else VT_HIP(hipStreamWaitEvent(stream, e->ev_loop, 0));
    }
    int rc = ensure_bytes(&e->d_loop, &e->d_loop_bytes, loop_need);
    if (rc != VT_OK) return rc;
    if (e->h_live_bytes < live_b) {
        if (e->h_live) { (void)hipHostFree(e->h_live); e->h_live = nullptr; e->h_live_bytes = 0; }
        VT_HIP(pinned_malloc(reinterpret_cast<void**>(&e->h_live), live_b));
        e->h_live_bytes = live_b;
    }
    // (the body is a function of its own: whatever it returns, the event behind what it DID enqueue is recorded -- a loop that
    // failed half-way has work in flight that still uses the engine's queues, and the next loop must wait for it)
    auto enqueue_loop = [&]() -> int {
        char* base = static_cast<char*>(e->d_loop);
        vt_ray* R[2] = {reinterpret_cast<vt_ray*>(base), reinterpret_cast<vt_ray*>(base + ray_b)};
        uint32_t* I[2] = {reinterpret_cast<uint32_t*>(base + 2 * ray_b), reinterpret_cast<uint32_t*>(base + 2 * ray_b + id_b)};
        vt_hit* hits_scratch = reinterpret_cast<vt_hit*>(base + 2 * ray_b + 2 * id_b);
        uint32_t* offsets = reinterpret_cast<uint32_t*>(base + 2 * ray_b + 2 * id_b + hit_b);
        uint32_t* d_live = reinterpret_cast<uint32_t*>(base + 2 * ray_b + 2 * id_b + hit_b + off_b);   // [d] = paths alive at depth d (d >= 1)
        vt_hit* H = static_cast<vt_hit*>(d_hits);

        const vt_ray* rays_q = static_cast<const vt_ray*>(d_rays);
        const uint32_t* ids_q = nullptr;
        if (s->has_alpha) {
            // ---- older form: the live count comes back to the host once per depth (alpha-test kernels take their ray count from the host)
            uint64_t m = n;
            if (live_out) live_out[0] = n;
            for (uint32_t d = 0; d < depth; ++d) {
                vt_hit* row = H + uint64_t(d) * n;
                if (m < n) VT_HIP(launch_fill_miss(row, n, nullptr, stream));   // paths that ended earlier read as misses
                const bool last = d + 1 == depth;
                if (m != 0) {
                    vt_hit* hits_q = d == 0 ? row : hits_scratch;               // depth 0: queue order = path order
                    rc = launch(s, rays_q, m, hits_q, nullptr, nullptr, false, false, stream);
                    if (rc != VT_OK) return rc;
                    QueueArgs qa{s->d_tris, s->d_prim_to_slot, rays_q, hits_q, ids_q, m, nullptr, d == 0 ? nullptr : row,
                                 last ? nullptr : R[d & 1], last ? nullptr : I[d & 1], offsets, seed + d};
                    VT_HIP(launch_queue_step(qa, d_live, stream));
                    if (!last) {
                        VT_HIP(hipMemcpyAsync(e->h_live, d_live, 4, hipMemcpyDeviceToHost, stream));
                        VT_HIP(hipStreamSynchronize(stream));
                        m = *e->h_live;
                        rays_q = R[d & 1];
                        ids_q = I[d & 1];
                    }
                }
                if (live_out && !last) live_out[d + 1] = m;
            }
            return VT_OK;
        }
        for (uint32_t d = 0; d < depth; ++d) {
            vt_hit* row = H + uint64_t(d) * n;
            const uint32_t* count = d == 0 ? nullptr : d_live + d;          // depth 0: all n paths, known here
            if (d != 0) VT_HIP(launch_fill_miss(row, n, count, stream));    // paths that ended earlier read as misses
            const bool last = d + 1 == depth;
            vt_hit* hits_q = d == 0 ? row : hits_scratch;                   // depth 0: queue order = path order
            const BatchReq req{rays_q, hits_q, n, 0, count};
            rc = launch_batches(s, &req, 1, nullptr, false, false, stream);
            if (rc != VT_OK) return rc;
            QueueArgs qa{s->d_tris, s->d_prim_to_slot, rays_q, hits_q, ids_q, n, count, d == 0 ? nullptr : row,
                         last ? nullptr : R[d & 1], last ? nullptr : I[d & 1], offsets, seed + d};
            VT_HIP(launch_queue_step(qa, d_live + d + 1, stream));
            rays_q = R[d & 1];
            ids_q = I[d & 1];
        }
        if (live_out) {
            live_out[0] = n;
            if (depth > 1) {
                VT_HIP(hipMemcpyAsync(e->h_live, d_live + 1, size_t(depth - 1) * 4, hipMemcpyDeviceToHost, stream));
                LiveCopy* job = new LiveCopy{e->h_live, live_out + 1, depth - 1};
                const hipError_t herr = VT_TRY(hipLaunchHostFunc(stream, copy_live_counts, job));
                if (herr != hipSuccess) { delete job; return fail(VT_ERR_HIP, std::string("vt_bounce_loop_dev: ") + hipGetErrorString(herr)); }
            }
        }
        return VT_OK;
    };
    const int loop_rc = enqueue_loop();
    const hipError_t rec = hipEventRecord(e->ev_loop, stream);
    if (rec == hipSuccess) e->loop_used = true;
    if (loop_rc != VT_OK) return loop_rc;
    if (rec != hipSuccess) return fail(VT_ERR_HIP, std::string("vt_bounce_loop_dev: hipEventRecord: ") + hipGetErrorString(rec));
    return VT_OK;
}

// ---- calls that move a scene's geometry: refit and skin refit, in three phases over ALL members of the scene's group ----------
// A group's scene lives on every member (replicas); a call that moves its geometry goes to every one of them, also when one
// of them fails: a refit that is refused for non-finite vertices has rewritten that member's records by then (and left it refusing
// to trace), and the members of a group must not end up with different geometry.  One host thread, three phases:
//   prepare  every member: argument checks, the device synchronisation (records are rewritten in place: traces in flight on
//            caller streams must finish first), allocations -- everything that may block or allocate;
//   enqueue  every member: copies, kernels, the level-by-level refit (a hipGraph, captured on first use), the read-back of the
//            verdict -- asynchronous calls only, so member k + 1's work is enqueued while member k's runs;
//   finish   every member: wait for its stream, read the verdict (non-finite triangles poison the scene), refresh the packet radius.
// No member is waited for before the last member's work has been enqueued: a frame costs one member's device time plus the
// enqueue time of the others instead of their sum (round 4: members one after the other, 0.3 - 0.4 ms each for a skinned
// 1 M-triangle scene).  Side by side on host THREADS was tried in round 4 and taken out: a member's allocations / device
// synchronisation invalidate the stream capture of another member's refit graph when members share a device; with the
// allocations and synchronisations in their own phase no capture runs beside them.
struct UpdateJob {
    vt_scene* s = nullptr;
    int rc = VT_OK;              // first failure of this member (a failed member skips its later phases)
    std::string msg;
    bool active = false;         // prepare passed and there is work to do
};

static int job_fail(UpdateJob& j, int rc)
{
    if (j.rc == VT_OK) { j.rc = rc; j.msg = vt_last_error(); }
    j.active = false;
    return rc;
}

// pinned read-back block of a scene: the finite check's counter and the root pair behind a refit
static int ensure_verdict_block(vt_scene* s)
{
    if (!s->d_bad) VT_HIP(dev_malloc(reinterpret_cast<void**>(&s->d_bad), 64));
    if (!s->h_verdict) VT_HIP(pinned_malloc(reinterpret_cast<void**>(&s->h_verdict), 128));
    return VT_OK;
}

// enqueue: all pair bounds from the (already rewritten) triangle records, deepest level first, then the read-back of the verdict
static int enqueue_levels_and_verdict(vt_scene* s)
{
    vt_engine* e = s->engine;
    const size_t levels = s->level_begin.empty() ? 0 : s->level_begin.size() - 1;
    auto enqueue = [&]() -> int {
        for (size_t k = 0; k < levels; ++k) {
            RefitLevelArgs la{reinterpret_cast<vt_node_pair*>(s->d_records), s->d_tris, s->d_level_pairs + s->level_begin[k],
                              s->level_begin[k + 1] - s->level_begin[k]};
            VT_HIP(launch_refit_level(la, e->stream));
        }
        return VT_OK;
    };
    // One launch per tree level, most of them a handful of threads: launch-bound.  The sequence only depends on
    // the scene's topology, so it is captured into a hipGraph on first use and replayed afterwards.
    if (!s->refit_graph && levels > 4) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            t_capturing = true;
            const int rc = enqueue();
            const hipError_t end = hipStreamEndCapture(e->stream, &graph);
            t_capturing = false;
            if (rc == VT_OK && end == hipSuccess && graph &&
                hipGraphInstantiate(&s->refit_graph, graph, nullptr, nullptr, 0) != hipSuccess)
                s->refit_graph = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
        }
        (void)hipGetLastError();                        // a failed capture falls back to plain launches below
    }
    if (s->refit_graph) {
        VT_HIP(hipGraphLaunch(s->refit_graph, e->stream));
    } else {
        const int rc = enqueue();
        if (rc != VT_OK) return rc;
    }
    // the verdict: how many triangles had a non-finite vertex, and the root pair (the scene's extent moved with the vertices:
    // the packet probe's radius follows) -- into pinned memory, read in the finish phase
    VT_HIP(hipMemcpyAsync(s->h_verdict, s->d_bad, 4, hipMemcpyDeviceToHost, e->stream));
    if (s->npairs != 0) VT_HIP(hipMemcpyAsync(s->h_verdict + 64, s->d_records, sizeof(vt_node_pair), hipMemcpyDeviceToHost, e->stream));
    return VT_OK;
}

// the order of enqueues ('E', one per member, logged when the member's last asynchronous call has returned) and host waits ('W',
// logged where the update path waits for a member's stream) of the latest group-wide update: a 'W' in front of the last 'E'
// means a member was waited for while another still had work to enqueue
static thread_local std::string t_update_log;

static int finish_update(vt_scene* s, const char* who)
{
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, std::string(who) + ": hipSetDevice failed");
    t_update_log.push_back('W');
    VT_HIP(hipStreamSynchronize(e->stream));
    uint32_t bad = 0;
    std::memcpy(&bad, s->h_verdict, 4);
    if (s->npairs != 0) {
        vt_node_pair root;
        std::memcpy(&root, s->h_verdict + 64, sizeof(root));
        s->coherent_radius2 = packet_radius2(root);
    }
    s->poisoned = bad != 0;
    if (bad != 0)
        return fail(VT_ERR_INVALID_ARG, std::string(who) + ": " + std::to_string(bad) + " triangles have a non-finite (NaN / inf) vertex; "
                                        "the scene cannot be traced until it is refitted with finite data");
    return VT_OK;
}

// The three phases over every member (replicas first, the root last, as before); the caller sees the root's failure, else the
// first replica's.  The order log becomes engine option "last_update_early_waits" (tests/fake_group_check.py asserts 0).
// `stage` (optional) runs once between the two first phases, on the root: input that every member needs goes to the root's device
// there, the other members fetch it device to device in their enqueue step.
static int update_every_member(vt_scene* root, const char* who, const std::function<int(vt_scene*)>& prepare,
                               const std::function<int(vt_scene*)>& enqueue, const std::function<int(vt_scene*)>& after,
                               const std::function<int(vt_scene*)>& stage = nullptr)
{
    std::vector<UpdateJob> jobs;
    if (root) for (vt_scene* rep : root->replicas) { UpdateJob j; j.s = rep; jobs.push_back(j); }
    { UpdateJob j; j.s = root; jobs.push_back(j); }
    t_update_log.clear();
    const auto t_begin = std::chrono::steady_clock::now();
    bool prepared = true;
    for (UpdateJob& j : jobs) {                          // prepare: may block, may allocate -- and changes nothing a trace can see
        const int rc = prepare(j.s);
        if (rc == VT_OK) j.active = true;
        else if (rc > 0) { job_fail(j, rc); prepared = false; }   // rc < 0: nothing to do for this member (empty scene), not a failure
    }
    // a member that cannot even prepare (bad arguments, out of memory) stops the call for ALL of them before anything is
    // rewritten: the members of a group never end up with different geometry because one of them could not allocate
    if (!prepared) for (UpdateJob& j : jobs) j.active = false;
    if (stage && jobs.back().active) {
        const int rc = stage(root);
        if (rc != VT_OK) { job_fail(jobs.back(), rc); for (UpdateJob& j : jobs) j.active = false; }   // nothing has been rewritten yet
    }
    for (UpdateJob& j : jobs) {                          // enqueue: asynchronous calls only
        if (!j.active) continue;
        DeviceGuard guard(j.s->engine->device);
        if (!guard.ok) { job_fail(j, fail(VT_ERR_HIP, std::string(who) + ": hipSetDevice failed")); continue; }
        const int rc = enqueue(j.s);
        t_update_log.push_back('E');
        if (rc != VT_OK) job_fail(j, rc);
    }
    const auto t_enqueued = std::chrono::steady_clock::now();
    for (UpdateJob& j : jobs) {                          // finish: the first host wait of the call
        if (!j.active) continue;
        int rc = finish_update(j.s, who);
        if (rc == VT_OK && after) rc = after(j.s);
        if (rc != VT_OK) job_fail(j, rc);
    }
    if (root && root->engine) {
        const size_t last_e = t_update_log.rfind('E');
        const uint32_t early_waits = last_e == std::string::npos ? 0 : uint32_t(std::count(t_update_log.begin(), t_update_log.begin() + long(last_e), 'W'));
        root->engine->last_update_members = uint32_t(jobs.size());
        root->engine->last_update_early_waits = early_waits;
        const auto t_end = std::chrono::steady_clock::now();
        root->engine->last_update_enqueue_us = uint32_t(std::chrono::duration<double, std::micro>(t_enqueued - t_begin).count());
        root->engine->last_update_wait_us = uint32_t(std::chrono::duration<double, std::micro>(t_end - t_enqueued).count());
    }
    for (UpdateJob& j : jobs)                            // a failed member that passed `prepare` left work on its stream: drain it
        if (j.rc != VT_OK && j.s && j.s->engine) { DeviceGuard guard(j.s->engine->device); (void)hipStreamSynchronize(j.s->engine->stream); }
    const UpdateJob& rj = jobs.back();
    if (rj.rc != VT_OK) return fail(rj.rc, rj.msg);
    for (const UpdateJob& j : jobs) if (j.rc != VT_OK) return fail(j.rc, j.msg);
    return VT_OK;
}

int vt_scene_refit(vt_scene* s, const float* verts, const uint8_t* flags, uint32_t n)
{
    std::vector<std::unique_lock<std::mutex>> locks;       // one call that rewrites a scene at a time, on every member
    bool alpha_from_flags = false;
    if (flags) for (uint32_t i = 0; i < n && !alpha_from_flags; ++i) alpha_from_flags = (flags[i] & VT_TRI_ALPHATEST) != 0;
    auto prepare = [&](vt_scene* m) -> int {
        if (!m) return fail(VT_ERR_INVALID_ARG, "vt_scene_refit: scene is NULL");
        if (!m->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_refit: the scene\'s engine has been closed");
        if (n != m->ntris) return fail(VT_ERR_INVALID_ARG, "vt_scene_refit: n differs from the scene's triangle count");
        if (n == 0) return -1;
        if (!verts) return fail(VT_ERR_INVALID_ARG, "vt_scene_refit: verts is NULL");
        vt_engine* e = m->engine;
        DeviceGuard guard(e->device);
        if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_refit: hipSetDevice failed");
        locks.emplace_back(e->host_mu);
        // records and bounds are rewritten in place: traces of this scene still in flight on caller streams must finish
        // first (vt_trace_*_dev is asynchronous), or rays would read half-updated boxes
        VT_HIP(hipDeviceSynchronize());
        int rc = ensure_bytes(&e->d_rays, &e->d_rays_bytes, size_t(n) * 9 * sizeof(float));
        if (rc == VT_OK && flags) rc = ensure_bytes(&e->d_out, &e->d_out_bytes, n);
        return rc != VT_OK ? rc : ensure_verdict_block(m);
    };
    // The vertices (36 B per triangle, pageable caller memory: the copy holds the host for its whole length) go up ONCE, to the
    // root's staging area; the other members of a group fetch them from there, device to device, behind an event -- round 5
    // uploaded them per member (0.7 ms of host time each for a million triangles, profiles/r5/notes.md section 3).
    const size_t vert_b = size_t(n) * 9 * sizeof(float);
    auto stage = [&](vt_scene* root) -> int {
        vt_engine* e = root->engine;
        DeviceGuard guard(e->device);
        if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_refit: hipSetDevice failed");
        VT_HIP(hipMemcpyAsync(e->d_rays, verts, vert_b, hipMemcpyHostToDevice, e->stream));
        if (flags) VT_HIP(hipMemcpyAsync(e->d_out, flags, n, hipMemcpyHostToDevice, e->stream));
        if (!root->replicas.empty()) {
            if (!e->ev_staged) VT_HIP(hipEventCreateWithFlags(&e->ev_staged, hipEventDisableTiming));
            VT_HIP(hipEventRecord(e->ev_staged, e->stream));
        }
        return VT_OK;
    };
    auto enqueue = [&](vt_scene* m) -> int {
        vt_engine* e = m->engine;
        if (flags) m->has_alpha = alpha_from_flags;        // new flags replace the old ones: so does the scene's alpha-test state
        if (vt_engine* re = m != s ? e->root : nullptr) {  // a replica of s: the staged input comes from the root's device
            VT_HIP(hipStreamWaitEvent(e->stream, re->ev_staged, 0));
            VT_HIP(hipMemcpyPeerAsync(e->d_rays, e->device, re->d_rays, re->device, vert_b, e->stream));
            if (flags) VT_HIP(hipMemcpyPeerAsync(e->d_out, e->device, re->d_out, re->device, n, e->stream));
        }
        VT_HIP(hipMemsetAsync(m->d_bad, 0, 4, e->stream));
        RefitTrisArgs ta{static_cast<const float*>(e->d_rays), flags ? static_cast<const uint8_t*>(e->d_out) : nullptr,
                         m->d_prim_to_slot, m->d_tris, n, m->d_bad};
        VT_HIP(launch_refit_tris(ta, e->stream));
        m->mark_host_copies_stale();             // the host copies (single-ray path) are now out of date
        return enqueue_levels_and_verdict(m);
    };
    auto after = [&](vt_scene* m) -> int {                 // the flags switched the alpha test on: its records are built now
        return flags && m->has_alpha && !m->alpha_ready ? build_alpha_records(m) : VT_OK;
    };
    return update_every_member(s, "vt_scene_refit", prepare, enqueue, after, stage);
}

int vt_scene_set_skin(vt_scene* s, const float* bind_verts, const vt_skin_vertex* skin, const uint32_t* matrix_base, uint32_t n)
{
    if (s) for (vt_scene* rep : s->replicas) { const int rc = vt_scene_set_skin(rep, bind_verts, skin, matrix_base, n); if (rc != VT_OK) return rc; }
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_skin: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_skin: the scene\'s engine has been closed");
    if (n != s->ntris) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_skin: n differs from the scene's triangle count");
    if (n == 0) return VT_OK;
    if (!bind_verts || !skin || !matrix_base) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_skin: NULL argument");
    for (size_t i = 0; i < size_t(n) * 3; ++i) {
        if (skin[i].num_bones > 3) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_skin: a vertex has more than 3 bones");
        for (uint32_t q = 0; q < skin[i].num_bones; ++q)
            if (skin[i].bone[q] < 0) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_skin: negative bone id");
    }
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_set_skin: hipSetDevice failed");
    const size_t vb = size_t(n) * 9 * sizeof(float), sb = size_t(n) * 3 * sizeof(vt_skin_vertex), mb = size_t(n) * sizeof(uint32_t);
    if (!s->d_bind_verts) {
        // all three or none: d_bind_verts != NULL is what "the scene has skin data" means everywhere else
        void *v = nullptr, *k = nullptr, *m = nullptr;
        hipError_t err = dev_malloc(&v, vb);
        if (err == hipSuccess) err = dev_malloc(&k, sb);
        if (err == hipSuccess) err = dev_malloc(&m, mb);
        if (err != hipSuccess) {
            (void)hipGetLastError();
            if (v) (void)hipFree(v);
            if (k) (void)hipFree(k);
            return fail(VT_ERR_HIP, std::string("vt_scene_set_skin: ") + hipGetErrorString(err));
        }
        s->d_bind_verts = static_cast<float*>(v); s->d_skin = static_cast<vt_skin_vertex*>(k); s->d_matrix_base = static_cast<uint32_t*>(m);
        s->bytes += vb + sb + mb;
    }
    VT_HIP(hipMemcpyAsync(s->d_bind_verts, bind_verts, vb, hipMemcpyHostToDevice, e->stream));
    VT_HIP(hipMemcpyAsync(s->d_skin, skin, sb, hipMemcpyHostToDevice, e->stream));
    VT_HIP(hipMemcpyAsync(s->d_matrix_base, matrix_base, mb, hipMemcpyHostToDevice, e->stream));
    VT_HIP(hipStreamSynchronize(e->stream));
    return VT_OK;
}

int vt_scene_skin_refit(vt_scene* s, const float* bones, const float* binds, uint32_t nmat)
{
    std::vector<std::unique_lock<std::mutex>> locks;
    auto prepare = [&](vt_scene* m) -> int {
        if (!m) return fail(VT_ERR_INVALID_ARG, "vt_scene_skin_refit: scene is NULL");
        if (!m->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_skin_refit: the scene\'s engine has been closed");
        if (m->ntris == 0) return -1;
        if (!m->d_bind_verts) return fail(VT_ERR_INVALID_ARG, "vt_scene_skin_refit: no skin data (call vt_scene_set_skin first)");
        if (nmat == 0 || !bones || !binds) return fail(VT_ERR_INVALID_ARG, "vt_scene_skin_refit: no matrices");
        vt_engine* e = m->engine;
        DeviceGuard guard(e->device);
        if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_skin_refit: hipSetDevice failed");
        locks.emplace_back(e->host_mu);
        VT_HIP(hipDeviceSynchronize());          // in-flight traces of this scene read the records that are about to change
        if (nmat > m->mats_cap) {
            if (m->d_skin_mats) { VT_HIP(hipFree(m->d_skin_mats)); m->d_skin_mats = nullptr; m->mats_cap = 0; }
            VT_HIP(dev_malloc(reinterpret_cast<void**>(&m->d_skin_mats), size_t(nmat) * 3 * 64));
            m->mats_cap = nmat;
        }
        return ensure_verdict_block(m);
    };
    auto enqueue = [&](vt_scene* m) -> int {
        vt_engine* e = m->engine;
        float* d_bones = m->d_skin_mats;
        float* d_binds = d_bones + size_t(m->mats_cap) * 16;
        float* d_prod = d_binds + size_t(m->mats_cap) * 16;
        VT_HIP(hipMemcpyAsync(d_bones, bones, size_t(nmat) * 64, hipMemcpyHostToDevice, e->stream));
        VT_HIP(hipMemcpyAsync(d_binds, binds, size_t(nmat) * 64, hipMemcpyHostToDevice, e->stream));
        VT_HIP(launch_skin_matrices(SkinMatricesArgs{d_bones, d_binds, d_prod, nmat}, e->stream));
        VT_HIP(hipMemsetAsync(m->d_bad, 0, 4, e->stream));
        SkinTrisArgs ta{m->d_bind_verts, m->d_skin, m->d_matrix_base, d_prod, m->d_prim_to_slot, m->d_tris, m->ntris, nmat, m->d_bad};
        VT_HIP(launch_skin_tris(ta, e->stream));
        VT_HIP(skin_frames(m, d_prod, nmat, e->stream));                     // normals / tangents, AccelStruct.cpp:82-92
        m->mark_host_copies_stale();
        return enqueue_levels_and_verdict(m);
    };
    return update_every_member(s, "vt_scene_skin_refit", prepare, enqueue, nullptr);
}

int vt_scene_read_records(vt_scene* s, vt_node_pair* pairs_out, vt_tri64* tris_out)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_scene_read_records: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_read_records: the scene\'s engine has been closed");
    DeviceGuard guard(s->engine->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_read_records: hipSetDevice failed");
    VT_HIP(hipStreamSynchronize(s->engine->stream));
    if (pairs_out && s->npairs) VT_HIP(hipMemcpy(pairs_out, s->d_records, size_t(s->npairs) * sizeof(vt_node_pair), hipMemcpyDeviceToHost));
    if (tris_out && s->ntris) VT_HIP(hipMemcpy(tris_out, s->d_tris, size_t(s->ntris) * sizeof(vt_tri64), hipMemcpyDeviceToHost));
    return VT_OK;
}

int vt_host_scene_sync(vt_host_scene* hsw, vt_scene* s)
{
    if (!hsw || !s) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_sync: NULL argument");
    HostScene& hs = hsw->hs;
    if (hs.pairs.size() != s->npairs || hs.tris.size() != s->ntris || hs.root_leaf_count != s->root_leaf_count)
        return fail(VT_ERR_INVALID_ARG, "vt_host_scene_sync: the scene was not uploaded from this host scene");
    if (s->poisoned) return fail(VT_ERR_INVALID_ARG, "vt_host_scene_sync: the last refit left non-finite triangles; refit with finite data first");
    const int rc = vt_scene_read_records(s, hs.pairs.data(), hs.tris.data());
    if (rc != VT_OK) return rc;
    bool alpha = false;                                   // a refit may change the flags
    for (const vt_tri64& t : hs.tris) alpha |= (t.flags & VT_TRI_ALPHATEST) != 0;
    hs.has_alpha = alpha;
    hsw->stale->store(0, std::memory_order_release);
    return VT_OK;
}

int vt_scene_set_tri_attribs(vt_scene* s, const vt_tri_attribs* attribs, uint32_t n)
{
    if (s) for (vt_scene* rep : s->replicas) { const int rc = vt_scene_set_tri_attribs(rep, attribs, n); if (rc != VT_OK) return rc; }
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_attribs: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_attribs: the scene\'s engine has been closed");
    if (n != s->ntris) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_attribs: n differs from the scene's triangle count");
    if (n != 0 && !attribs) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_tri_attribs: attribs is NULL");
    DeviceGuard guard(s->engine->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_set_tri_attribs: hipSetDevice failed");
    if (n == 0) return VT_OK;
    const size_t bytes = size_t(n) * sizeof(vt_tri_attribs);
    std::lock_guard<std::mutex> host_lock(s->engine->host_mu);   // as vt_scene_refit: one call that rewrites the scene at a time
    if (!s->d_attribs) {
        VT_HIP(dev_malloc(reinterpret_cast<void**>(&s->d_attribs), bytes));
        s->bytes += bytes;
    }
    VT_HIP(hipDeviceSynchronize());                  // traces in flight may read the AlphaRecs derived from the old table
    VT_HIP(hipMemcpy(s->d_attribs, attribs, bytes, hipMemcpyHostToDevice));
    return build_alpha_records(s);
}

int vt_scene_set_alpha(vt_scene* s, const vt_alpha_material* mats, uint32_t nmats, const uint8_t* texels, uint64_t ntexels)
{
    if (s) for (vt_scene* rep : s->replicas) { const int rc = vt_scene_set_alpha(rep, mats, nmats, texels, ntexels); if (rc != VT_OK) return rc; }
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_alpha: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_alpha: the scene\'s engine has been closed");
    if (nmats == 0 || !mats) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_alpha: no materials");
    for (uint32_t i = 0; i < nmats; ++i) {
        const vt_alpha_material& m = mats[i];
        if (m.filter > 1) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_alpha: filter must be 0 (nearest) or 1 (bilinear)");
        if ((m.width == 0) != (m.height == 0)) return fail(VT_ERR_INVALID_ARG, "vt_scene_set_alpha: width and height must both be 0 or both be set");
        if (m.width && (m.offset > ntexels || uint64_t(m.width) * m.height > ntexels - m.offset || !texels))
            return fail(VT_ERR_INVALID_ARG, "vt_scene_set_alpha: an alpha plane lies outside the texel array");
        // what one 64-B AlphaRec can hold (trace_kernels.h): 16-bit plane sides, and 31-bit texel indices -- the kernels form
        // plane offset + y * width + x in 32 bits, so the plane's LAST texel must lie below 2^31 too
        if (m.width > 65535u || m.height > 65535u || (m.width && m.offset + uint64_t(m.width) * m.height > (uint64_t(1) << 31)))
            return fail(VT_ERR_UNSUPPORTED, "vt_scene_set_alpha: alpha planes are limited to 65535 x 65535 texels and must end within the first 2 GiB of the texel array");
    }
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_scene_set_alpha: hipSetDevice failed");
    std::lock_guard<std::mutex> host_lock(e->host_mu);           // as vt_scene_refit: one call that rewrites the scene at a time
    VT_HIP(hipDeviceSynchronize());                              // no launch may still read the old tables
    // room for the AlphaRecs and the new tables first, then the swap: a call that runs out of memory leaves the scene as it was
    // (old tables, old AlphaRecs)
    if (const int room = ensure_alpha_room(s); room != VT_OK) return room;
    const size_t mats_b = size_t(nmats) * sizeof(vt_alpha_material), tex_b = std::max<uint64_t>(ntexels, 16);
    void *d_mats = nullptr, *d_tex = nullptr;
    hipError_t err = dev_malloc(&d_mats, mats_b);
    if (err == hipSuccess) err = dev_malloc(&d_tex, tex_b);
    if (err == hipSuccess) err = VT_TRY(hipMemcpy(d_mats, mats, mats_b, hipMemcpyHostToDevice));
    if (err == hipSuccess && ntexels) err = VT_TRY(hipMemcpy(d_tex, texels, ntexels, hipMemcpyHostToDevice));
    if (err != hipSuccess) {
        (void)hipGetLastError();
        if (d_mats) (void)hipFree(d_mats);
        if (d_tex) (void)hipFree(d_tex);
        return fail(VT_ERR_HIP, std::string("vt_scene_set_alpha: ") + hipGetErrorString(err));
    }
    if (s->d_alpha_mats) { (void)hipFree(s->d_alpha_mats); s->bytes -= s->alpha_table_bytes; }
    if (s->d_alpha_texels) (void)hipFree(s->d_alpha_texels);
    s->d_alpha_mats = static_cast<vt_alpha_material*>(d_mats);
    s->d_alpha_texels = static_cast<uint8_t*>(d_tex);
    s->n_alpha_mats = nmats;
    s->alpha_table_bytes = mats_b + ntexels;
    s->bytes += s->alpha_table_bytes;
    return build_alpha_records(s);
}

int vt_hit_shade_dev(vt_scene* s, const void* d_hits, uint64_t n, void* d_out, void* stream)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_hit_shade_dev: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_hit_shade_dev: the scene\'s engine has been closed");
    if (n == 0) return VT_OK;
    if (!d_hits || !d_out) return fail(VT_ERR_INVALID_ARG, "vt_hit_shade_dev: NULL device buffer");
    if (!s->d_attribs) return fail(VT_ERR_INVALID_ARG, "vt_hit_shade_dev: call vt_scene_set_tri_attribs first");
    DeviceGuard guard(s->engine->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_hit_shade_dev: hipSetDevice failed");
    HitShadeArgs a{};
    a.attribs = s->d_attribs;
    a.hits = static_cast<const vt_hit*>(d_hits);
    a.out = static_cast<vt_hit_shade*>(d_out);
    a.n = n;
    VT_HIP(launch_hit_shade(a, static_cast<hipStream_t>(stream)));
    return VT_OK;
}

int vt_gen_primary_dev(vt_engine* e, const vt_camera* cam, void* d_rays, void* stream)
{
    if (!e || !cam || !d_rays) return fail(VT_ERR_INVALID_ARG, "vt_gen_primary_dev: NULL argument");
    if (uint64_t(cam->width) * cam->height > 0x7FFFFFFFull * kBlockThreads) return fail(VT_ERR_INVALID_ARG, "vt_gen_primary_dev: image too large");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_gen_primary_dev: hipSetDevice failed");
    GenPrimaryArgs a{*cam, static_cast<vt_ray*>(d_rays)};
    VT_HIP(launch_gen_primary(a, static_cast<hipStream_t>(stream)));
    return VT_OK;
}

int vt_gen_bounce_dev(vt_engine* e, const void* d_attrs, uint64_t n, uint64_t seed, void* d_rays, void* stream)
{
    if (!e) return fail(VT_ERR_INVALID_ARG, "vt_gen_bounce_dev: engine is NULL");
    if (n == 0) return VT_OK;
    if (!d_attrs || !d_rays) return fail(VT_ERR_INVALID_ARG, "vt_gen_bounce_dev: NULL device buffer");
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_gen_bounce_dev: hipSetDevice failed");
    GenBounceArgs a{static_cast<const vt_hit_attrs*>(d_attrs), static_cast<vt_ray*>(d_rays), n, seed};
    VT_HIP(launch_gen_bounce(a, static_cast<hipStream_t>(stream)));
    return VT_OK;
}

int vt_engine_synchronize(vt_engine* e)
{
    if (!e) return fail(VT_ERR_INVALID_ARG, "vt_engine_synchronize: NULL");
    for (vt_engine* p : e->peers) { const int rc = vt_engine_synchronize(p); if (rc != VT_OK) return rc; }
    DeviceGuard guard(e->device);
    VT_HIP(hipStreamSynchronize(e->stream));
    if (e->s_comm) VT_HIP(hipStreamSynchronize(e->s_comm));      // gathers in flight (multi_gpu.hip)
    return VT_OK;
}

int vt_engine_set_timing(vt_engine* e, int enabled)
{
    if (!e) return fail(VT_ERR_INVALID_ARG, "vt_engine_set_timing: NULL");
    for (vt_engine* p : e->peers) (void)vt_engine_set_timing(p, enabled);      // a group is timed as a whole
    e->timing = enabled != 0;
    e->ev_valid = false;
    return VT_OK;
}

int vt_engine_last_kernel_ms(vt_engine* e, float* ms)
{
    if (!e || !ms) return fail(VT_ERR_INVALID_ARG, "vt_engine_last_kernel_ms: NULL");
    if (!e->ev_valid) return fail(VT_ERR_INVALID_ARG, "vt_engine_last_kernel_ms: no timed launch yet");
    DeviceGuard guard(e->device);
    VT_HIP(hipEventSynchronize(e->ev_stop));
    VT_HIP(hipEventElapsedTime(ms, e->ev_start, e->ev_stop));
    return VT_OK;
}

int vt_engine_launch_info(vt_engine* e, uint32_t* blocks, uint32_t* threads, uint32_t* lds_bytes)
{
    if (!e) return fail(VT_ERR_INVALID_ARG, "vt_engine_launch_info: NULL");
    if (blocks) *blocks = e->last_blocks;
    if (threads) *threads = e->last_threads;
    if (lds_bytes) *lds_bytes = e->last_lds;
    return VT_OK;
}

} // extern "C"

namespace vt {

float scene_packet_radius2(const vt_node_pair& root) { return packet_radius2(root); }

// Rebuild on a group (reference: one tree per AccelStruct, source/objects/AccelStruct.cpp:762-775; here one copy per GPU).  The
// root's scene is complete on its device; every other member gets the same bytes -- records, triangle -> slot table, level lists --
// by device-to-device copies on its own stream (xGMI between the GPUs of a node), in the three phases of the group-wide refits:
// prepare (allocations on every member), enqueue (asynchronous copies only), finish (the waits).  The host uploads the tree ONCE;
// round 5 uploaded and re-numbered it per member, one member after the other.
int scene_replicate(vt_scene* s, const std::function<int(vt_engine*, vt_scene**)>& rebuild)
{
    vt_engine* e = s->engine;
    if (e->peers.empty()) return VT_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t rec_b = s->record_capacity * 64, slot_b = size_t(s->ntris) * 4, lvl_b = size_t(s->npairs) * 4;
    std::vector<vt_scene*> reps;
    int rc = VT_OK;
    for (vt_engine* p : e->peers) {                       // prepare: the shells and their device arrays
        DeviceGuard guard(p->device);
        if (!guard.ok) { rc = fail(VT_ERR_HIP, "scene replica: hipSetDevice failed"); break; }
        vt_scene* r = new vt_scene();
        r->engine = p;
        r->has_alpha = s->has_alpha; r->npairs = s->npairs; r->ntris = s->ntris; r->max_depth = s->max_depth;
        r->root_leaf_count = s->root_leaf_count; r->tri_base = s->tri_base; r->alpha_base = s->alpha_base;
        r->record_capacity = s->record_capacity; r->coherent_radius2 = s->coherent_radius2; r->level_begin = s->level_begin;
        r->upload_stats = s->upload_stats;
        r->host_copies = s->host_copies;                  // the host copies the root was uploaded from go stale with this member's refits too
        track_scene(p, r);
        reps.push_back(r);
        hipError_t err = dev_malloc(reinterpret_cast<void**>(&r->d_records), rec_b);
        if (err == hipSuccess && slot_b) err = dev_malloc(reinterpret_cast<void**>(&r->d_prim_to_slot), slot_b);
        if (err == hipSuccess && lvl_b) err = dev_malloc(reinterpret_cast<void**>(&r->d_level_pairs), lvl_b);
        if (err != hipSuccess) { (void)hipGetLastError(); rc = fail(VT_ERR_HIP, std::string("scene replica: ") + hipGetErrorString(err)); break; }
        r->d_tris = reinterpret_cast<vt_tri64*>(r->d_records + size_t(r->tri_base) * 64);
        r->bytes = rec_b + slot_b + lvl_b;
    }
    std::string log;                                      // 'E' a member's copies are enqueued, 'W' a member is waited for
    bool copy_refused = false;
    if (rc == VT_OK) {
        for (vt_scene* r : reps) {                        // enqueue: asynchronous copies only (the root's stream is idle: its upload ended with a wait)
            vt_engine* p = r->engine;
            DeviceGuard guard(p->device);
            hipError_t err = guard.ok ? hipSuccess : hipErrorInvalidDevice;
            if (err == hipSuccess) err = VT_TRY(hipMemcpyPeerAsync(r->d_records, p->device, s->d_records, e->device, rec_b, p->stream));
            if (err == hipSuccess && slot_b) err = VT_TRY(hipMemcpyPeerAsync(r->d_prim_to_slot, p->device, s->d_prim_to_slot, e->device, slot_b, p->stream));
            if (err == hipSuccess && lvl_b) err = VT_TRY(hipMemcpyPeerAsync(r->d_level_pairs, p->device, s->d_level_pairs, e->device, lvl_b, p->stream));
            log.push_back('E');
            if (err != hipSuccess) { (void)hipGetLastError(); copy_refused = true; rc = fail(VT_ERR_HIP, std::string("scene replica: ") + hipGetErrorString(err)); break; }
        }
    }
    const auto t_enqueued = std::chrono::steady_clock::now();
    for (vt_scene* r : reps) {                            // finish: the first host waits of the call (also behind a failure: nothing may be in flight when the shells go)
        DeviceGuard guard(r->engine->device);
        if (log.find('E') != std::string::npos) log.push_back('W');
        const hipError_t err = VT_TRY(hipStreamSynchronize(r->engine->stream));
        if (err != hipSuccess && rc == VT_OK) { copy_refused = true; rc = fail(VT_ERR_HIP, std::string("scene replica: ") + hipGetErrorString(err)); }
    }
    const size_t last_e = log.rfind('E');
    e->last_update_members = uint32_t(reps.size()) + 1;
    e->last_update_early_waits = last_e == std::string::npos ? 0 : uint32_t(std::count(log.begin(), log.begin() + long(last_e), 'W'));
    e->last_update_enqueue_us = uint32_t(std::chrono::duration<double, std::micro>(t_enqueued - t_begin).count());
    e->last_update_wait_us = uint32_t(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enqueued).count());
    if (rc != VT_OK) {
        const std::string msg = vt_last_error();
        for (vt_scene* r : reps) vt_scene_free(r);
        reps.clear();
        if (!copy_refused || !rebuild) return fail(rc, msg);
        // a runtime that refuses copies between these two devices: every member is uploaded from the host instead (round 5's way)
        for (vt_engine* p : e->peers) {
            vt_scene* rep = nullptr;
            rc = rebuild(p, &rep);
            if (rc != VT_OK) { const std::string m2 = vt_last_error(); for (vt_scene* r : reps) vt_scene_free(r); return fail(rc, m2); }
            reps.push_back(rep);
        }
    }
    s->replicas = reps;
    return VT_OK;
}

int engine_launch(vt_scene* s, const void* d_rays, uint64_t n, void* d_hits, void* d_occ, void* d_stats, bool any_hit, bool stats,
                  hipStream_t stream)
{
    return launch(s, d_rays, n, d_hits, d_occ, d_stats, any_hit, stats, stream);
}

} // namespace vt
