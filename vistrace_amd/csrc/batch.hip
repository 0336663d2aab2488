// batch.hip -- the host-facing half of the device engine: everything that moves rays and results between caller memory and the
// device around the launches of engine.hip.
//
//   vt_trace_closest / vt_trace_any      host arrays in, host arrays out: tiny batches in mapped pinned memory, small ones with one
//                                        copy each way, large ones through a chunked pipeline (pinned staging for pageable arrays,
//                                        the copy engines alone for page-locked ones: vt_host_register);
//   vt_batch_*                           a traced batch that stays on the device (what accel:TraverseBatch(buffer) holds), streamed
//                                        in through the same staging buffers, range checks inside the staging copy;
//   vt_batch_set_* / vt_batch_trace_closest_set   several host buffers, ONE merged launch.
// The reference has none of this: it traces one ray per Lua call on the host (source/objects/AccelStruct.cpp:778-838).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "engine_internal.h"

using namespace vt;

namespace {

int take_pinned(vt_engine* e, size_t need, vt_batch::HostArray& h)
{
    {
        std::lock_guard<std::mutex> lock(e->launch_mu);
        for (size_t k = 0; k < e->pinned_spare.size(); ++k)
            if (e->pinned_spare[k].second >= need && e->pinned_spare[k].second <= 2 * need) {
                h.p = e->pinned_spare[k].first; h.bytes = e->pinned_spare[k].second;
                e->pinned_spare.erase(e->pinned_spare.begin() + long(k));
                return VT_OK;
            }
    }
    VT_HIP(pinned_malloc(&h.p, need));
    h.bytes = need;
    return VT_OK;
}

int ensure_host_pipeline(vt_engine* e)
{
    if (e->pipeline_ready) return VT_OK;
    // every piece is created once; a call that failed half-way (out of pinned memory) is resumed by the next one
    const uint64_t C = vt_engine::kHostChunk;
    if (!e->s_in) VT_HIP(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
    if (!e->s_out) VT_HIP(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
    for (int k = 0; k < vt_engine::kStageBufs; ++k) {
        if (!e->h_stage_in[k]) VT_HIP(pinned_malloc(reinterpret_cast<void**>(&e->h_stage_in[k]), C * sizeof(vt_ray)));
        if (!e->h_stage_out[k]) VT_HIP(pinned_malloc(reinterpret_cast<void**>(&e->h_stage_out[k]), C * sizeof(vt_hit)));
        if (!e->ev_in[k]) VT_HIP(hipEventCreateWithFlags(&e->ev_in[k], hipEventDisableTiming));
        if (!e->ev_k[k]) VT_HIP(hipEventCreateWithFlags(&e->ev_k[k], hipEventDisableTiming));
        if (!e->ev_out[k]) VT_HIP(hipEventCreateWithFlags(&e->ev_out[k], hipEventDisableTiming));
    }
    e->pipeline_ready = true;
    return VT_OK;
}

// the range checks of AccelStruct::Traverse (source/objects/AccelStruct.cpp:805-806) over rays that are already on the device:
// first_bad = min(first_bad, index of a ray with tMin < 0 or tMax <= tMin); NaN ranges pass, as in the reference
__global__ __launch_bounds__(256) void check_ranges_kernel(const vt_ray* rays, uint64_t n, uint64_t base, unsigned long long* first_bad)
{
    const uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float tmin = rays[i].tmin, tmax = rays[i].tmax;
    if (tmin < 0.f || VT_MUT(57, tmax < tmin, tmax <= tmin)) atomicMin(first_bad, static_cast<unsigned long long>(VT_MUT(58, i, base + i)));   // (VT_MUT: vt_internal.h)
}

constexpr size_t kBatchTail = 256;       // bytes behind a batch's arrays in its device block: the first-bad-ray word of its upload

// Does this host's runtime copy PAGEABLE memory to the device at the rate of pinned memory?  (On the round-4 boxes it does: 56 GB/s
// either way, profiles/r4/upload_probe.txt -- then staging a caller's buffer through pinned memory with host threads only adds
// work.)  Measured once per process, 8 MB each way, on the engine's upload stream.
bool pageable_copies_are_fast(vt_engine* e)
{
    static std::atomic<int> known{-1};
    int k = known.load(std::memory_order_acquire);
    if (k >= 0) return k != 0;
    if (const char* env = std::getenv("VT_BATCH_UPLOAD")) {                 // "staged" / "direct": skip the measurement
        k = std::strcmp(env, "direct") == 0 ? 1 : 0;
        known.store(k, std::memory_order_release);
        return k != 0;
    }
    const size_t bytes = size_t(8) << 20;
    k = 0;
    void* d = nullptr;
    char* pageable = static_cast<char*>(std::malloc(bytes));
    if (pageable && dev_malloc(&d, bytes) == hipSuccess) {
        std::memset(pageable, 1, bytes);
        auto timed = [&](const void* src) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                if (hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, e->s_in) != hipSuccess || hipStreamSynchronize(e->s_in) != hipSuccess) return 1e30;
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            return best;
        };
        const double t_pinned = timed(e->h_stage_in[0]), t_pageable = timed(pageable);
        k = t_pageable <= 1.4 * t_pinned ? 1 : 0;
    }
    if (d) (void)hipFree(d);
    std::free(pageable);
    (void)hipGetLastError();
    known.store(k, std::memory_order_release);
    return k != 0;
}

// The batch's rays -> device, traced, hit records -> the batch's pinned host block, chunk by chunk: while chunk c is staged (a few
// host threads copy it from the caller's bytes into a pinned buffer, looking at every ray's range on the way if asked to) and
// uploaded, chunk c - 1 is traced and the hit records of chunk c - 2 come back.  The caller's memory is free when this returns;
// the tail of the trace, the result kernels and the last download are not waited for (b->done, b->hits_down).
// e->host_mu is held.  *bad_ray < n: a ray failed the range checks, nothing of the batch is valid.
// trace_chunks = false: only staged and uploaded (a batch set is traced by ONE merged launch behind all its uploads).
// *stage_turn: which of the two pinned staging buffers is next (carried across the batches of a set).
int batch_pipeline(vt_scene* s, vt_batch* b, const void* rays, uint64_t n, uint32_t image_width, uint32_t flags, uint64_t* bad_ray,
                   bool trace_chunks = true, uint64_t* stage_turn = nullptr)
{
    vt_engine* e = s->engine;
    int rc = ensure_host_pipeline(e);
    if (rc != VT_OK) return rc;
    // chunk = 256 Ki rays (8 MB of rays: long enough to stream at the link's rate, short enough for four stages to overlap within
    // a 1 Mi-ray batch); whole bands of 16 image rows when the batch is an image, so that every chunk is tiled like the whole
    uint64_t C = uint64_t(1) << 18;
    if (image_width >= 4 && image_width % 4 == 0) {
        const uint64_t band = uint64_t(image_width) * 16;
        if (band <= vt_engine::kHostChunk) C = std::max<uint64_t>(C / band, 1) * band;
    }
    const bool check = (flags & VT_BATCH_CHECK_RANGES) != 0;
    const bool fetch = (flags & VT_BATCH_FETCH_HITS) != 0;
    if (fetch) {
        rc = take_pinned(e, n * sizeof(vt_hit), b->h_hits);
        if (rc != VT_OK) return rc;
    }
    uint64_t own_turn = 0;
    uint64_t& turn = stage_turn ? *stage_turn : own_turn;
    // Where the runtime moves pageable memory at the pinned rate the chunks go up straight from the caller's bytes (no staging
    // copy, no host thread touches a ray) and the range checks run on the device behind each upload: 1 Mi rays 1.1 -> 0.8 ms.
    const bool direct = pageable_copies_are_fast(e);
    unsigned long long* const d_first_bad = reinterpret_cast<unsigned long long*>(b->d_mem + b->d_mem_bytes - kBatchTail);
    if (direct && check) {
        if (!e->h_bad) VT_HIP(pinned_malloc(reinterpret_cast<void**>(&e->h_bad), 64));
        VT_HIP(hipMemsetAsync(d_first_bad, 0xFF, sizeof(unsigned long long), e->s_in));
    }
    // (the pinned staging buffers are free: every host-pointer call leaves them so, and e->host_mu is held; the kernels an earlier
    // batch may still have in flight on the engine's stream work on that batch's own device block)
    const uint64_t nchunks = (n + C - 1) / C;
    for (uint64_t c = 0; c < nchunks; ++c, ++turn) {
        const int k = int(turn % vt_engine::kStageBufs);
        const uint64_t lo = c * C, m = std::min(C, n - lo);
        char* d_in = b->d_mem + lo * sizeof(vt_ray);
        char* d_res = static_cast<char*>(b->d_hits) + lo * sizeof(vt_hit);
        const void* src = static_cast<const char*>(rays) + lo * sizeof(vt_ray);
        if (!direct) {
            if (turn >= uint64_t(vt_engine::kStageBufs)) VT_HIP(hipEventSynchronize(e->ev_in[k]));   // pinned input buffer k is free again
            vt_ray* stage = reinterpret_cast<vt_ray*>(e->h_stage_in[k]);
            if (check) {
                const uint64_t bad = parallel_copy_checked(stage, src, m);
                if (bad < m) { *bad_ray = lo + bad; return VT_OK; }
            } else {
                parallel_copy(stage, src, m * sizeof(vt_ray));
            }
            src = stage;
        }
        VT_HIP(hipMemcpyAsync(d_in, src, m * sizeof(vt_ray), hipMemcpyHostToDevice, e->s_in));
        if (direct && check) {
            hipLaunchKernelGGL(check_ranges_kernel, dim3(uint32_t((m + 255) / 256)), dim3(256), 0, e->s_in,
                               reinterpret_cast<const vt_ray*>(d_in), m, lo, d_first_bad);
            VT_HIP(hipGetLastError());
        }
        VT_HIP(hipEventRecord(e->ev_in[k], e->s_in));
        VT_HIP(hipStreamWaitEvent(e->stream, e->ev_in[k], 0));
        if (!trace_chunks) continue;
        const BatchReq one{d_in, d_res, m, image_width};
        rc = launch_batches(s, &one, 1, nullptr, false, false, e->stream);
        if (rc != VT_OK) return rc;
        if (fetch) {
            VT_HIP(hipEventRecord(e->ev_k[k], e->stream));
            VT_HIP(hipStreamWaitEvent(e->s_out, e->ev_k[k], 0));
            VT_HIP(hipMemcpyAsync(static_cast<char*>(b->h_hits.p) + lo * sizeof(vt_hit), d_res, m * sizeof(vt_hit), hipMemcpyDeviceToHost, e->s_out));
        }
    }
    if (direct && check) {                                   // the verdict of the device-side checks: behind the last upload, not behind a trace
        VT_HIP(hipMemcpyAsync(e->h_bad, d_first_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, e->s_in));
        VT_HIP(hipStreamSynchronize(e->s_in));
        unsigned long long first_bad;
        std::memcpy(&first_bad, e->h_bad, sizeof(first_bad));
        if (first_bad < n) *bad_ray = first_bad;
    }
    if (!trace_chunks) return VT_OK;                         // the set's caller traces, downloads and waits for the uploads
    if (fetch) { VT_HIP(hipEventRecord(b->hits_down, e->s_out)); b->hits_in_flight = true; }
    VT_HIP(hipStreamSynchronize(e->s_in));                   // every upload has left the staging buffers (and the caller's memory long before)
    return VT_OK;
}

// a batch object with its device block (rays | hits | attrs | shade | tbn) from the engine's spare blocks or a new allocation
int batch_new(vt_scene* s, uint64_t n, vt_batch** out)
{
    vt_engine* e = s->engine;
    vt_batch* b = new vt_batch();
    b->engine = e;
    b->n = n;
    *out = b;
    if (n == 0) return VT_OK;
    auto al = [](uint64_t x) { return (x + 255) & ~uint64_t(255); };
    const uint64_t ray_b = al(n * sizeof(vt_ray)), hit_b = al(n * sizeof(vt_hit)), att_b = al(n * sizeof(vt_hit_attrs));
    const uint64_t sha_b = s->d_attribs ? al(n * sizeof(vt_hit_shade)) : 0;
    const uint64_t tbn_b = s->d_frames ? al(n * sizeof(vt_hit_tbn)) : 0;
    const size_t need = ray_b + hit_b + att_b + sha_b + tbn_b + kBatchTail;
    hipError_t err = hipSuccess;
    {
        std::lock_guard<std::mutex> lock(e->launch_mu);
        for (size_t k = 0; k < e->device_spare.size(); ++k)
            if (e->device_spare[k].second >= need && e->device_spare[k].second <= 2 * need) {
                b->d_mem = e->device_spare[k].first; b->d_mem_bytes = e->device_spare[k].second;
                e->device_spare.erase(e->device_spare.begin() + long(k));
                break;
            }
    }
    if (!b->d_mem) { err = dev_malloc(reinterpret_cast<void**>(&b->d_mem), need); b->d_mem_bytes = need; }
    if (err == hipSuccess) err = VT_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    if (err == hipSuccess) err = VT_TRY(hipEventCreateWithFlags(&b->hits_down, hipEventDisableTiming));
    if (err != hipSuccess) {
        if (b->d_mem) (void)hipFree(b->d_mem);
        if (b->done) (void)hipEventDestroy(b->done);
        delete b;
        *out = nullptr;
        return fail(VT_ERR_HIP, std::string("vt_batch_trace_closest: ") + hipGetErrorString(err));
    }
    b->d_hits = b->d_mem + ray_b;
    b->d_attrs = b->d_mem + ray_b + hit_b;
    b->d_shade = sha_b ? b->d_mem + ray_b + hit_b + att_b : nullptr;
    b->d_tbn = tbn_b ? b->d_mem + ray_b + hit_b + att_b + sha_b : nullptr;
    return VT_OK;
}

// the result kernels behind a batch's trace (TraceResult core, shading part) and its `done` event, on the engine's stream
int batch_finish(vt_scene* s, vt_batch* b)
{
    if (b->n == 0) return VT_OK;
    vt_engine* e = s->engine;
    HitAttrsArgs a{s->d_tris, s->d_prim_to_slot, reinterpret_cast<const vt_ray*>(b->d_mem), static_cast<const vt_hit*>(b->d_hits),
                   static_cast<vt_hit_attrs*>(b->d_attrs), b->n};
    hipError_t err = launch_hit_attrs(a, e->stream);
    if (err == hipSuccess && b->d_shade) {
        HitShadeArgs sa{s->d_attribs, static_cast<const vt_hit*>(b->d_hits), static_cast<vt_hit_shade*>(b->d_shade), b->n};
        err = launch_hit_shade(sa, e->stream);
    }
    // the shading frame, cone switched off as by accel:Traverse's defaults (coneWidth = coneAngle = -1, AccelStruct.cpp:796-806)
    if (err == hipSuccess && b->d_tbn) err = launch_hit_tbn(s, b->d_mem, b->d_hits, b->n, -1.f, -1.f, b->d_tbn, e->stream);
    if (err == hipSuccess) err = VT_TRY(hipEventRecord(b->done, e->stream));
    if (err != hipSuccess) return fail(VT_ERR_HIP, std::string("vt_batch_trace_closest: ") + hipGetErrorString(err));
    return VT_OK;
}

// a batch that will not be handed out: its blocks go back to the engine (everything enqueued for it has been waited for)
void batch_discard(vt_engine* e, vt_batch* b)
{
    b->hits_in_flight = false;
    {
        std::lock_guard<std::mutex> lock(e->launch_mu);
        e->batches.push_back(b);                             // vt_batch_free takes it off again and recycles its blocks
    }
    vt_batch_free(b);
}

} // namespace

extern "C" {

static int trace_host(vt_scene* s, const vt_ray* rays, uint64_t n, void* out, size_t out_elem, bool any_hit)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_trace: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_trace: the scene\'s engine has been closed");
    if (n == 0) return VT_OK;
    if (!rays || !out) return fail(VT_ERR_INVALID_ARG, "vt_trace: NULL buffer");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_trace: hipSetDevice failed");
    std::lock_guard<std::mutex> host_lock(e->host_mu);       // one host-buffer call at a time per engine (shared staging)
    if (n <= vt_engine::kTinyRays) {
        // tiny batches: the kernel works directly on pinned host memory (no copy calls).  The result slots are pre-set
        // to values the kernel never writes -- prim 0xFFFFFFFE, and a NaN with a payload no arithmetic produces in t, u
        // and v (a hit's t, u, v are finite, a miss writes zeros) -- and the host watches ALL FOUR words of every
        // record change (any-hit: the one byte), so nothing depends on a 16-byte device->host store landing as one
        // piece.  That is a PCIe write away from the last ray's finish, while hipStreamSynchronize adds the queue's
        // completion signalling (~5-10 us per call).  If nothing arrives within a few milliseconds (a fault, a
        // debugger) the ordinary wait takes over and reports the error.
        std::memcpy(e->h_tiny_rays, rays, n * sizeof(vt_ray));
        const uint32_t kPendingPrim = 0xFFFFFFFEu;               // not VT_MISS and never a triangle index
        const uint32_t kPendingF32  = 0x7FA5C3E1u;               // signalling-NaN pattern
        volatile uint32_t* const slots32 = reinterpret_cast<volatile uint32_t*>(e->h_tiny_out);
        volatile uint8_t* const slots8 = reinterpret_cast<volatile uint8_t*>(e->h_tiny_out);
        for (uint64_t i = 0; i < n; ++i) {
            if (any_hit) slots8[i] = 0xFFu;
            else { slots32[i * 4] = kPendingPrim; slots32[i * 4 + 1] = slots32[i * 4 + 2] = slots32[i * 4 + 3] = kPendingF32; }
        }
        std::atomic_thread_fence(std::memory_order_seq_cst);
        int rc = engine_launch(s, e->d_tiny_rays, n, any_hit ? nullptr : e->d_tiny_out, any_hit ? e->d_tiny_out : nullptr, nullptr,
                        any_hit, false, e->stream);
        if (rc != VT_OK) return rc;
        bool arrived = false;
        if (e->spin_wait) {
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spins = 0; !arrived; ++spins) {
                arrived = true;
                for (uint64_t i = 0; i < n && arrived; ++i)
                    arrived = any_hit ? slots8[i] != 0xFFu
                                      : (slots32[i * 4] != kPendingPrim && slots32[i * 4 + 1] != kPendingF32 &&
                                         slots32[i * 4 + 2] != kPendingF32 && slots32[i * 4 + 3] != kPendingF32);
                if (!arrived && (spins & 1023u) == 1023u &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5))
                    break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!arrived) VT_HIP(hipStreamSynchronize(e->stream));
        std::memcpy(out, e->h_tiny_out, n * out_elem);
        return VT_OK;
    }
    const uint64_t C = vt_engine::kHostChunk;
    if (n <= 2 * C) {   // small batch: one copy each way around one launch (256 Ki-ray chunks were measured for 1 - 4 Mi rays: no gain)
        int rc = ensure_bytes(&e->d_rays, &e->d_rays_bytes, n * sizeof(vt_ray));
        if (rc == VT_OK) rc = ensure_bytes(&e->d_out, &e->d_out_bytes, n * out_elem);
        if (rc != VT_OK) return rc;
        VT_HIP(hipMemcpyAsync(e->d_rays, rays, n * sizeof(vt_ray), hipMemcpyHostToDevice, e->stream));
        rc = engine_launch(s, e->d_rays, n, any_hit ? nullptr : e->d_out, any_hit ? e->d_out : nullptr, nullptr, any_hit, false, e->stream);
        if (rc != VT_OK) return rc;
        VT_HIP(hipMemcpyAsync(out, e->d_out, n * out_elem, hipMemcpyDeviceToHost, e->stream));
        VT_HIP(hipStreamSynchronize(e->stream));
        return VT_OK;
    }

    // Large batch: chunks of C rays flow through pinned double buffers -- while chunk c is traced, chunk c+1 is
    // staged and uploaded and chunk c-1 comes back and is copied out to the caller's (pageable) memory.
    if (int prc = ensure_host_pipeline(e); prc != VT_OK) return prc;
    constexpr uint64_t NB = vt_engine::kStageBufs, LAG = vt_engine::kStageLag;
    int rc = ensure_bytes(&e->d_rays, &e->d_rays_bytes, NB * C * sizeof(vt_ray));
    if (rc == VT_OK) rc = ensure_bytes(&e->d_out, &e->d_out_bytes, NB * C * sizeof(vt_hit));
    if (rc != VT_OK) return rc;
    VT_HIP(hipStreamSynchronize(e->stream));                 // earlier work on the engine's stream owns the staging buffers
    const uint64_t nchunks = (n + C - 1) / C;
    // Caller arrays that are page-locked already (vt_host_register, hipHostMalloc, a pinned torch tensor) need no staging at all:
    // the copy engines read and write them directly, uploads and downloads overlap, and no host thread touches a byte
    // (16 Mi rays: 9.6 instead of 14 ms on the round-4 box; registering a buffer costs ~70 us per MB once, which is why
    // pageable arrays -- a fresh Lua string per call -- are staged instead)
    auto page_locked = [](const void* p) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return at.type == hipMemoryTypeHost;
    };
    if (page_locked(rays) && page_locked(reinterpret_cast<const char*>(rays) + n * sizeof(vt_ray) - 1) && page_locked(out) &&
        page_locked(static_cast<const char*>(out) + n * out_elem - 1)) {
        for (uint64_t c = 0; c < nchunks; ++c) {
            const int b = int(c % NB);
            const uint64_t m = std::min(C, n - c * C);
            char* d_in = static_cast<char*>(e->d_rays) + size_t(b) * C * sizeof(vt_ray);
            char* d_res = static_cast<char*>(e->d_out) + size_t(b) * C * sizeof(vt_hit);
            if (c >= NB) VT_HIP(hipStreamWaitEvent(e->s_in, e->ev_k[b], 0));   // chunk c-NB has read device buffer b
            VT_HIP(hipMemcpyAsync(d_in, rays + c * C, m * sizeof(vt_ray), hipMemcpyHostToDevice, e->s_in));
            VT_HIP(hipEventRecord(e->ev_in[b], e->s_in));
            VT_HIP(hipStreamWaitEvent(e->stream, e->ev_in[b], 0));
            if (c >= NB) VT_HIP(hipStreamWaitEvent(e->stream, e->ev_out[b], 0)); // chunk c-NB's results have left device buffer b
            rc = engine_launch(s, d_in, m, any_hit ? nullptr : d_res, any_hit ? d_res : nullptr, nullptr, any_hit, false, e->stream);
            if (rc != VT_OK) return rc;
            VT_HIP(hipEventRecord(e->ev_k[b], e->stream));
            VT_HIP(hipStreamWaitEvent(e->s_out, e->ev_k[b], 0));
            VT_HIP(hipMemcpyAsync(static_cast<char*>(out) + c * C * out_elem, d_res, m * out_elem, hipMemcpyDeviceToHost, e->s_out));
            VT_HIP(hipEventRecord(e->ev_out[b], e->s_out));
        }
        VT_HIP(hipStreamSynchronize(e->s_out));
        VT_HIP(hipStreamSynchronize(e->stream));
        return VT_OK;
    }
    auto drain = [&](uint64_t c) -> int {                    // chunk c's results: pinned -> caller
        const int b = int(c % NB);
        const uint64_t m = std::min(C, n - c * C);
        VT_HIP(hipEventSynchronize(e->ev_out[b]));
        parallel_copy(static_cast<char*>(out) + c * C * out_elem, e->h_stage_out[b], m * out_elem);
        return VT_OK;
    };
    for (uint64_t c = 0; c < nchunks; ++c) {
        const int b = int(c % NB);
        const uint64_t m = std::min(C, n - c * C);
        char* d_in = static_cast<char*>(e->d_rays) + size_t(b) * C * sizeof(vt_ray);
        char* d_res = static_cast<char*>(e->d_out) + size_t(b) * C * sizeof(vt_hit);
        if (c >= NB) VT_HIP(hipEventSynchronize(e->ev_in[b]));           // pinned input buffer b is free again
        parallel_copy(e->h_stage_in[b], rays + c * C, m * sizeof(vt_ray));
        if (c >= NB) VT_HIP(hipStreamWaitEvent(e->s_in, e->ev_k[b], 0));   // chunk c-NB has read device buffer b
        VT_HIP(hipMemcpyAsync(d_in, e->h_stage_in[b], m * sizeof(vt_ray), hipMemcpyHostToDevice, e->s_in));
        VT_HIP(hipEventRecord(e->ev_in[b], e->s_in));
        VT_HIP(hipStreamWaitEvent(e->stream, e->ev_in[b], 0));
        if (c >= NB) VT_HIP(hipStreamWaitEvent(e->stream, e->ev_out[b], 0)); // chunk c-NB's results have left device buffer b
        rc = engine_launch(s, d_in, m, any_hit ? nullptr : d_res, any_hit ? d_res : nullptr, nullptr, any_hit, false, e->stream);
        if (rc != VT_OK) return rc;
        VT_HIP(hipEventRecord(e->ev_k[b], e->stream));
        VT_HIP(hipStreamWaitEvent(e->s_out, e->ev_k[b], 0));
        VT_HIP(hipMemcpyAsync(e->h_stage_out[b], d_res, m * out_elem, hipMemcpyDeviceToHost, e->s_out));
        VT_HIP(hipEventRecord(e->ev_out[b], e->s_out));
        if (c >= LAG && (rc = drain(c - LAG)) != VT_OK) return rc;     // pinned output buffer (c - LAG) % NB is free again before chunk c - LAG + NB needs it
    }
    for (uint64_t c = nchunks > LAG ? nchunks - LAG : 0; c < nchunks; ++c)
        if ((rc = drain(c)) != VT_OK) return rc;
    VT_HIP(hipStreamSynchronize(e->stream));
    return VT_OK;
}

int vt_host_register(void* p, size_t bytes)
{
    if (!p || bytes == 0) return fail(VT_ERR_INVALID_ARG, "vt_host_register: empty range");
    const hipError_t err = VT_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    if (err == hipErrorHostMemoryAlreadyRegistered) { (void)hipGetLastError(); return VT_OK; }
    if (err != hipSuccess) { (void)hipGetLastError(); return fail(VT_ERR_HIP, std::string("vt_host_register: ") + hipGetErrorString(err)); }
    return VT_OK;
}

int vt_host_unregister(void* p)
{
    if (!p) return fail(VT_ERR_INVALID_ARG, "vt_host_unregister: NULL");
    const hipError_t err = hipHostUnregister(p);
    if (err != hipSuccess) { (void)hipGetLastError(); return fail(VT_ERR_HIP, std::string("vt_host_unregister: ") + hipGetErrorString(err)); }
    return VT_OK;
}

int vt_trace_closest(vt_scene* s, const vt_ray* rays, uint64_t n, vt_hit* hits)
{
    if (s && !s->replicas.empty() && n >= kMultiHostMin) return multi_trace_host(s, rays, n, hits, sizeof(vt_hit), false);
    return trace_host(s, rays, n, hits, sizeof(vt_hit), false);
}

int vt_trace_any(vt_scene* s, const vt_ray* rays, uint64_t n, uint8_t* occluded)
{
    if (s && !s->replicas.empty() && n >= kMultiHostMin) return multi_trace_host(s, rays, n, occluded, sizeof(uint8_t), true);
    return trace_host(s, rays, n, occluded, sizeof(uint8_t), true);
}

// ---- vt_batch: a traced batch that stays on the device (see the header) ------------------------------------------------
int vt_batch_trace_closest_ex(vt_scene* s, const vt_ray* rays, uint64_t n, uint32_t ray_image_width, uint32_t flags, uint64_t* bad_ray,
                              vt_batch** out)
{
    if (!out) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: out is NULL");
    *out = nullptr;
    if (bad_ray) *bad_ray = n;
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: the scene\'s engine has been closed");
    if (n != 0 && !rays) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: rays is NULL");
    if (flags & ~(VT_BATCH_CHECK_RANGES | VT_BATCH_FETCH_HITS)) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: unknown flag");
    if ((flags & VT_BATCH_CHECK_RANGES) && !bad_ray) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: VT_BATCH_CHECK_RANGES needs bad_ray");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_batch_trace_closest: hipSetDevice failed");
    vt_batch* b = nullptr;
    int rc = batch_new(s, n, &b);
    if (rc != VT_OK) return rc;
    uint64_t bad = n;
    if (n != 0) {
        {
            std::lock_guard<std::mutex> host_lock(e->host_mu);
            // the rays come from caller memory that may go away when this call returns (a Lua string): every upload is waited
            // for, the kernels behind them are not
            rc = batch_pipeline(s, b, rays, n, ray_image_width, flags, &bad);
            if (rc == VT_OK && bad == n) rc = batch_finish(s, b);
            if (rc != VT_OK || bad != n) {                   // nothing of this batch survives: wait for what was enqueued, give the blocks back
                (void)hipStreamSynchronize(e->s_in); (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->s_out);
            }
        }
        if (rc != VT_OK || bad != n) {
            batch_discard(e, b);
            if (rc != VT_OK) return rc;
            *bad_ray = bad;
            return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest: ray " + std::to_string(bad) + " fails the range checks (tMin < 0 or tMax <= tMin)");
        }
    }
    {
        std::lock_guard<std::mutex> lock(e->launch_mu);
        e->batches.push_back(b);
    }
    *out = b;
    return VT_OK;
}

// ---- a SET of batches: buffers are added one by one (staged and uploaded at once), then traced by ONE merged launch ----------
struct vt_batch_set {
    vt_scene* scene = nullptr;
    uint32_t flags = 0;
    std::vector<vt_batch*> batches;
    std::vector<uint32_t> widths;
    uint64_t stage_turn = 0;
};

int vt_batch_set_begin(vt_scene* s, uint32_t flags, vt_batch_set** out)
{
    if (!out) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_begin: out is NULL");
    *out = nullptr;
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_begin: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_begin: the scene\'s engine has been closed");
    if (flags & ~(VT_BATCH_CHECK_RANGES | VT_BATCH_FETCH_HITS)) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_begin: unknown flag");
    vt_batch_set* set = new vt_batch_set();
    set->scene = s;
    set->flags = flags;
    s->open_sets.push_back(set);                 // vt_scene_free detaches it: the set must not keep a dangling scene
    *out = set;
    return VT_OK;
}

// the set leaves its scene's list (it is about to be deleted)
static void set_unlink(vt_batch_set* set)
{
    if (vt_scene* s = set->scene) s->open_sets.erase(std::remove(s->open_sets.begin(), s->open_sets.end(), set), s->open_sets.end());
    set->scene = nullptr;
}

// the batches of a set whose engine or scene has gone: the engine released their device memory when it closed (they were
// registered with it when they were added); what is left are the shells and their pinned host arrays
static void set_free_batches(vt_batch_set* set)
{
    for (vt_batch* b : set->batches) vt_batch_free(b);
    set->batches.clear();
}

void vt_batch_set_abort(vt_batch_set* set)
{
    if (!set) return;
    if (vt_scene* s = set->scene; s && s->engine) {
        vt_engine* e = s->engine;
        DeviceGuard guard(e->device);
        if (e->s_in) { (void)hipStreamSynchronize(e->s_in); (void)hipStreamSynchronize(e->s_out); }
        (void)hipStreamSynchronize(e->stream);
    }
    set_free_batches(set);                       // with or without an engine: vt_batch_free knows both
    set_unlink(set);
    delete set;
}

int vt_batch_set_add(vt_batch_set* set, const vt_ray* rays, uint64_t n, uint32_t ray_image_width, uint64_t* bad_ray)
{
    if (!set) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: set is NULL");
    if (bad_ray) *bad_ray = n;
    vt_scene* s = set->scene;
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: the set's scene has been freed");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: the scene\'s engine has been closed");
    if (n != 0 && !rays) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: rays is NULL");
    if (n >= (uint64_t(1) << 32)) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: a batch of a set holds at most 2^32 - 1 rays");
    if ((set->flags & VT_BATCH_CHECK_RANGES) && !bad_ray) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: VT_BATCH_CHECK_RANGES needs bad_ray");
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_batch_set_add: hipSetDevice failed");
    vt_batch* b = nullptr;
    int rc = batch_new(s, n, &b);
    if (rc != VT_OK) return rc;
    uint64_t bad = n;
    if (n != 0) {
        std::lock_guard<std::mutex> host_lock(e->host_mu);
        rc = batch_pipeline(s, b, rays, n, 0, set->flags, &bad, false, &set->stage_turn);
        // the staging buffers are free again (and the caller's memory long before) when this returns: other host-pointer calls may
        // run between two adds
        if (hipStreamSynchronize(e->s_in) != hipSuccess && rc == VT_OK) rc = fail(VT_ERR_HIP, "vt_batch_set_add: hipStreamSynchronize failed");
        set->stage_turn = 0;
    }
    if (rc != VT_OK || bad != n) {
        batch_discard(e, b);
        if (rc != VT_OK) return rc;
        *bad_ray = bad;
        return fail(VT_ERR_INVALID_ARG, "vt_batch_set_add: ray " + std::to_string(bad) + " fails the range checks (tMin < 0 or tMax <= tMin)");
    }
    {
        // registered with the engine from now on: should the engine be closed while the set is open, it releases the batch's
        // device memory and detaches it like any other live batch
        std::lock_guard<std::mutex> lock(e->launch_mu);
        e->batches.push_back(b);
    }
    set->batches.push_back(b);
    set->widths.push_back(ray_image_width);
    return VT_OK;
}

uint32_t vt_batch_set_count(const vt_batch_set* set) { return set ? uint32_t(set->batches.size()) : 0; }

int vt_batch_set_trace(vt_batch_set* set, vt_batch** out)
{
    if (!set) return fail(VT_ERR_INVALID_ARG, "vt_batch_set_trace: set is NULL");
    const uint32_t nb = uint32_t(set->batches.size());
    if (nb != 0 && !out) { vt_batch_set_abort(set); return fail(VT_ERR_INVALID_ARG, "vt_batch_set_trace: out is NULL"); }
    vt_scene* s = set->scene;
    if (!s || !s->engine) {
        const bool freed = !s;
        vt_batch_set_abort(set);
        return fail(VT_ERR_INVALID_ARG, freed ? "vt_batch_set_trace: the set's scene has been freed" : "vt_batch_set_trace: the scene\'s engine has been closed");
    }
    vt_engine* e = s->engine;
    DeviceGuard guard(e->device);
    int rc = guard.ok ? VT_OK : fail(VT_ERR_HIP, "vt_batch_set_trace: hipSetDevice failed");
    uint64_t total = 0;
    for (const vt_batch* b : set->batches) total += b->n;
    // (a set of empty buffers only -- accel:TraverseBatch({""}) -- launches nothing and fetches nothing: the engine's host pipeline,
    // whose events the fetch uses, is only created by the first non-empty add)
    if (rc == VT_OK && nb != 0 && total != 0) {
        std::lock_guard<std::mutex> host_lock(e->host_mu);
        // ONE merged launch over all batches (one grid start, one drain: launch_batches); then per batch the download of its hit
        // records (VT_BATCH_FETCH_HITS) and its result kernels
        std::vector<BatchReq> reqs(nb);
        for (uint32_t k = 0; k < nb; ++k) reqs[k] = BatchReq{set->batches[k]->d_mem, set->batches[k]->d_hits, set->batches[k]->n, set->widths[k]};
        rc = launch_batches(s, reqs.data(), nb, nullptr, false, false, e->stream);
        if (rc == VT_OK && (set->flags & VT_BATCH_FETCH_HITS)) {
            hipError_t err = VT_TRY(hipEventRecord(e->ev_k[0], e->stream));
            if (err == hipSuccess) err = VT_TRY(hipStreamWaitEvent(e->s_out, e->ev_k[0], 0));
            for (uint32_t k = 0; k < nb && err == hipSuccess; ++k) {
                vt_batch* b = set->batches[k];
                if (b->n == 0) continue;
                err = VT_TRY(hipMemcpyAsync(b->h_hits.p, b->d_hits, b->n * sizeof(vt_hit), hipMemcpyDeviceToHost, e->s_out));
                if (err == hipSuccess) err = VT_TRY(hipEventRecord(b->hits_down, e->s_out));
                b->hits_in_flight = err == hipSuccess;
            }
            if (err != hipSuccess) rc = fail(VT_ERR_HIP, std::string("vt_batch_set_trace: ") + hipGetErrorString(err));
        }
        for (uint32_t k = 0; k < nb && rc == VT_OK; ++k) rc = batch_finish(s, set->batches[k]);
    }
    if (rc != VT_OK) { vt_batch_set_abort(set); return rc; }
    for (uint32_t k = 0; k < nb; ++k) out[k] = set->batches[k];     // already registered with the engine (vt_batch_set_add)
    set_unlink(set);
    delete set;
    return VT_OK;
}

int vt_batch_trace_closest_set(vt_scene* s, const vt_ray* const* rays, const uint64_t* n, const uint32_t* ray_image_widths, uint32_t nbatches,
                               uint32_t flags, uint32_t* bad_batch, uint64_t* bad_ray, vt_batch** out)
{
    if (!out && nbatches) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest_set: out is NULL");
    for (uint32_t k = 0; k < nbatches; ++k) out[k] = nullptr;
    if (bad_batch) *bad_batch = nbatches;
    if (nbatches != 0 && (!rays || !n)) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest_set: NULL argument");
    if ((flags & VT_BATCH_CHECK_RANGES) && (!bad_ray || !bad_batch)) return fail(VT_ERR_INVALID_ARG, "vt_batch_trace_closest_set: VT_BATCH_CHECK_RANGES needs bad_batch and bad_ray");
    vt_batch_set* set = nullptr;
    int rc = vt_batch_set_begin(s, flags, &set);
    if (rc != VT_OK) return rc;
    for (uint32_t k = 0; k < nbatches; ++k) {
        uint64_t bad = n[k];
        rc = vt_batch_set_add(set, rays[k], n[k], ray_image_widths ? ray_image_widths[k] : 0, &bad);
        if (rc != VT_OK) {
            if (bad < n[k] && bad_batch) { *bad_batch = k; *bad_ray = bad; }
            vt_batch_set_abort(set);
            return rc;
        }
    }
    return vt_batch_set_trace(set, out);
}

int vt_batch_trace_closest(vt_scene* s, const vt_ray* rays, uint64_t n, vt_batch** out)
{
    return vt_batch_trace_closest_ex(s, rays, n, s && s->engine ? s->engine->ray_image_width : 0, 0, nullptr, out);
}

uint64_t vt_batch_count(const vt_batch* b) { return b ? b->n : 0; }

// one array of a batch, device -> pinned host memory, once
static int batch_fetch(vt_batch* b, const void* d_src, size_t elem, vt_batch::HostArray& h, const void** out, const char* who)
{
    *out = nullptr;
    if (!h.have) {
        if (b->n != 0) {
            vt_engine* e = b->engine;
            if (!e || !d_src) return fail(VT_ERR_INVALID_ARG, std::string(who) + (e ? ": not materialised for this batch" : ": the engine has been closed"));
            DeviceGuard guard(e->device);
            if (!guard.ok) return fail(VT_ERR_HIP, std::string(who) + ": hipSetDevice failed");
            if (&h == &b->h_hits && b->hits_in_flight) {     // VT_BATCH_FETCH_HITS: the records came back behind the trace
                VT_HIP(hipEventSynchronize(b->hits_down));
                b->hits_in_flight = false;
            } else {
                const size_t need = b->n * elem;
                if (!h.p) { const int rc = take_pinned(e, need, h); if (rc != VT_OK) return rc; }
                VT_HIP(hipEventSynchronize(b->done));
                VT_HIP(hipMemcpy(h.p, d_src, need, hipMemcpyDeviceToHost));
            }
        }
        h.have = true;
    }
    *out = h.p;
    return VT_OK;
}

int vt_batch_rays(vt_batch* b, const vt_ray** rays)
{
    if (!b || !rays) return fail(VT_ERR_INVALID_ARG, "vt_batch_rays: NULL");
    return batch_fetch(b, b->d_mem, sizeof(vt_ray), b->h_rays, reinterpret_cast<const void**>(rays), "vt_batch_rays");
}

int vt_batch_hits(vt_batch* b, const vt_hit** hits)
{
    if (!b || !hits) return fail(VT_ERR_INVALID_ARG, "vt_batch_hits: NULL");
    return batch_fetch(b, b->d_hits, sizeof(vt_hit), b->h_hits, reinterpret_cast<const void**>(hits), "vt_batch_hits");
}

int vt_batch_attrs(vt_batch* b, const vt_hit_attrs** attrs)
{
    if (!b || !attrs) return fail(VT_ERR_INVALID_ARG, "vt_batch_attrs: NULL");
    return batch_fetch(b, b->d_attrs, sizeof(vt_hit_attrs), b->h_attrs, reinterpret_cast<const void**>(attrs), "vt_batch_attrs");
}

int vt_batch_shade(vt_batch* b, const vt_hit_shade** shade)
{
    if (!b || !shade) return fail(VT_ERR_INVALID_ARG, "vt_batch_shade: NULL");
    if (b->n != 0 && !b->h_shade.have && b->engine && !b->d_shade)
        return fail(VT_ERR_INVALID_ARG, "vt_batch_shade: the scene had no triangle attributes (vt_scene_set_tri_attribs) when the batch was traced");
    return batch_fetch(b, b->d_shade, sizeof(vt_hit_shade), b->h_shade, reinterpret_cast<const void**>(shade), "vt_batch_shade");
}

int vt_batch_tbn(vt_batch* b, const vt_hit_tbn** tbn)
{
    if (!b || !tbn) return fail(VT_ERR_INVALID_ARG, "vt_batch_tbn: NULL");
    if (b->n != 0 && !b->h_tbn.have && b->engine && !b->d_tbn)
        return fail(VT_ERR_INVALID_ARG, "vt_batch_tbn: the scene had no vertex frames (vt_scene_set_tri_frames) when the batch was traced");
    return batch_fetch(b, b->d_tbn, sizeof(vt_hit_tbn), b->h_tbn, reinterpret_cast<const void**>(tbn), "vt_batch_tbn");
}

void vt_batch_free(vt_batch* b)
{
    if (!b) return;
    if (vt_engine* e = b->engine) {
        DeviceGuard guard(e->device);
        if (b->done) { (void)hipEventSynchronize(b->done); (void)hipEventDestroy(b->done); }
        if (b->hits_down) { if (b->hits_in_flight) (void)hipEventSynchronize(b->hits_down); (void)hipEventDestroy(b->hits_down); }
        std::lock_guard<std::mutex> lock(e->launch_mu);
        if (b->d_mem) {                                  // kept for the next batches that fit; the oldest spare goes when the list is full
            if (e->device_spare.size() >= 32) { (void)hipFree(e->device_spare.front().first); e->device_spare.erase(e->device_spare.begin()); }
            e->device_spare.push_back({b->d_mem, b->d_mem_bytes});
        }
        for (vt_batch::HostArray* h : {&b->h_rays, &b->h_hits, &b->h_attrs, &b->h_shade, &b->h_tbn}) {
            if (!h->p) continue;
            if (e->pinned_spare.size() < 8) e->pinned_spare.push_back({h->p, h->bytes});
            else (void)hipHostFree(h->p);
            h->p = nullptr;
        }
        e->batches.erase(std::remove(e->batches.begin(), e->batches.end(), b), e->batches.end());
    }
    for (vt_batch::HostArray* h : {&b->h_rays, &b->h_hits, &b->h_attrs, &b->h_shade, &b->h_tbn})
        if (h->p) (void)hipHostFree(h->p);              // the engine is gone: nothing to hand the blocks back to
    delete b;
}

void* vt_engine_stream(vt_engine* e) { return e ? static_cast<void*>(e->stream) : nullptr; }

} // extern "C"

namespace vt {

void batch_sets_detach(vt_scene* s)
{
    for (vt_batch_set* set : s->open_sets) set->scene = nullptr;
    s->open_sets.clear();
}

int engine_trace_host(vt_scene* s, const vt_ray* rays, uint64_t n, void* out, size_t out_elem, bool any_hit)
{
    return trace_host(s, rays, n, out, out_elem, any_hit);
}

} // namespace vt
