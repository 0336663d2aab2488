// multi_gpu.hip -- multi-GPU behind the C ABI: replicated scene, sharded rays, ONE RCCL gather of hit records.
//
// rays never interact and the scene is read-only, so the path shards by independent units -- every
// device holds a full replica of the linearised BVH in its own HBM (1 M triangles = 104 MB, 10 M = 1.04 GB of 288 GB),
// traces a contiguous shard of the ray array, and the only exchange is the gather of the 16-byte hit records to the
// root device (`ncclGather`, /opt/rocm/include/rccl/rccl.h:745; RCCL implements it as N-1 direct sends into the root,
// one per xGMI link).  No all-reduce, no all-to-all.  Two ways to form the group:
//   * vt_engine_open_multi(devices, ndev): ONE process drives all devices (what the reference's C++ module would do:
//     a Lua state is one thread of one process) -- `ncclCommInitAll`, the gather calls of all devices in one group;
//   * vt_engine_comm_init_rank(e, nranks, rank, id): one process per GPU (bench.py under torch.distributed.run) --
//     `ncclCommInitRank` with an id from vt_comm_unique_id that the launcher distributes.
// The gather of batch b runs on a communication stream beside the trace of batch b+1 (send buffers are double-buffered;
// engine option "reserved_cus" keeps room on the CUs for RCCL's kernels while a persistent trace grid is resident).
// Inside ONE batch (engine option "gather_chunks" = K > 1) a device's shard is traced in K pieces and piece c crosses the
// links while piece c + 1 is traced: a one-shot call then costs about max(trace, gather) + one piece instead of their sum.
// ncclGather places rank r's data at recvbuff + r * count, so a piece of every shard cannot land at its final place (stride
// = shard capacity) through it; the pieces therefore move as what ncclGather is made of -- one ncclSend per peer and the
// matching ncclRecv's on the root, all in one group (rccl.h: "ncclGather ... implemented with ncclSend / ncclRecv").
// K = 1 keeps the single ncclGather per batch.
//
// RCCL is loaded with dlopen on first use: a single-GPU user never needs the library, and a process that already
// holds a copy (PyTorch bundles one) keeps using that one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "engine_internal.h"

using namespace vt;

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommInitAll)    CommInitAll = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclGather)         Gather = nullptr;
    decltype(&ncclSend)           Send = nullptr;
    decltype(&ncclRecv)           Recv = nullptr;
    decltype(&ncclGroupStart)     GroupStart = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

RcclApi& rccl_state()
{
    static RcclApi api;
    return api;
}

RcclApi* rccl()
{
    RcclApi& api = rccl_state();
    static std::once_flag once;
    std::call_once(once, [&api] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        const char* env = std::getenv("VT_RCCL_LIB");
        if (env && *env) api.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
        for (const char* n : names) {
            if (api.handle) break;
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.handle) {
            const char* why = dlerror();                  // one call: dlerror() clears the message it returns
            api.error = std::string("cannot load librccl.so: ") + (why ? why : "not found");
            return;
        }
        bool ok = true;
        auto sym = [&](auto& fn, const char* name) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, name));
            if (!fn) { ok = false; api.error = std::string("librccl.so lacks ") + name; }
        };
        sym(api.GetUniqueId, "ncclGetUniqueId");
        sym(api.CommInitRank, "ncclCommInitRank");
        sym(api.CommInitAll, "ncclCommInitAll");
        sym(api.CommDestroy, "ncclCommDestroy");
        sym(api.Gather, "ncclGather");
        sym(api.Send, "ncclSend");
        sym(api.Recv, "ncclRecv");
        sym(api.GroupStart, "ncclGroupStart");
        sym(api.GroupEnd, "ncclGroupEnd");
        sym(api.GetErrorString, "ncclGetErrorString");
        if (!ok) { api.handle = nullptr; }
    });
    return api.handle ? &api : nullptr;
}

int rccl_fail(const char* what, ncclResult_t r)
{
    RcclApi* R = rccl();
    return fail(VT_ERR_HIP, std::string(what) + ": " + (R ? R->GetErrorString(r) : "RCCL not loaded"));
}

#define VT_NCCL(call)                                                     \
    do {                                                                  \
        ncclResult_t r__ = (call);                                        \
        if (r__ != ncclSuccess) return rccl_fail(#call, r__);             \
    } while (0)

// comm stream + events of one engine (its device is current)
int ensure_comm_side(vt_engine* e)
{
    if (!e->s_comm) VT_HIP(hipStreamCreateWithFlags(&e->s_comm, hipStreamNonBlocking));
    for (int c = 0; c < kMaxGatherChunks; ++c)
        if (!e->ev_traced[c]) VT_HIP(hipEventCreateWithFlags(&e->ev_traced[c], hipEventDisableTiming));
    for (int b = 0; b < 2; ++b)
        if (!e->ev_sent[b]) VT_HIP(hipEventCreateWithFlags(&e->ev_sent[b], hipEventDisableTiming));
    if (!e->ev_g0) VT_HIP(hipEventCreate(&e->ev_g0));
    if (!e->ev_g1) VT_HIP(hipEventCreate(&e->ev_g1));
    return VT_OK;
}

// the devices of a group in group order: root first
std::vector<vt_engine*> group_of(vt_engine* root)
{
    std::vector<vt_engine*> g{root};
    g.insert(g.end(), root->peers.begin(), root->peers.end());
    return g;
}

// single-process group: one communicator per device, created together on first use
int ensure_group_comms(vt_engine* root)
{
    RcclApi* R = rccl();
    if (!R) return fail(VT_ERR_HIP, "multi-GPU gather: " + rccl_state().error);
    const std::vector<vt_engine*> g = group_of(root);
    // A communicator from vt_engine_comm_init_rank spans the ranks of a multi-process job, not this group: a gather on
    // it would deliver comm_size x cap records into a buffer sized for the group (or hang in a mismatched collective).
    if (root->comm && (!root->comm_from_init_all || root->comm_size != int(g.size())))
        return fail(VT_ERR_INVALID_ARG, "multi-GPU gather: the engine's communicator was made by vt_engine_comm_init_rank "
                                        "(one process per GPU); the single-process group needs its own engine (vt_engine_open_multi)");
    if (!root->comm) {
        std::vector<int> devs;
        for (vt_engine* e : g) devs.push_back(e->device);
        std::vector<ncclComm_t> comms(g.size(), nullptr);
        VT_NCCL(R->CommInitAll(comms.data(), int(g.size()), devs.data()));
        for (size_t k = 0; k < g.size(); ++k) {
            g[k]->comm = comms[k];
            g[k]->comm_from_init_all = true;
            g[k]->comm_rank = int(k);
            g[k]->comm_size = int(g.size());
        }
    }
    for (vt_engine* e : g) {                              // idempotent: a failure here is retried by the next call
        DeviceGuard guard(e->device);
        if (!guard.ok) return fail(VT_ERR_HIP, "multi-GPU gather: hipSetDevice failed");
        const int rc = ensure_comm_side(e);
        if (rc != VT_OK) return rc;
    }
    return VT_OK;
}

// This device's calls for moving records [lo, hi) of every rank's `cap`-record shard to `root` (inside the caller's group):
// the whole shard through ncclGather, a piece through the sends / receives ncclGather is made of (see the file comment).
// d_send = this rank's shard; d_recv = the root's ndev * cap records (root only).  The root's own piece is already in place
// when it traced into its slice; otherwise it is copied on the communication stream.
ncclResult_t move_part(RcclApi* R, vt_engine* e, const void* d_send, void* d_recv, uint64_t cap, uint64_t lo, uint64_t hi, bool whole, int root)
{
    const ncclComm_t comm = static_cast<ncclComm_t>(e->comm);
    if (whole) return R->Gather(d_send, d_recv, cap * sizeof(vt_hit), ncclUint8, root, comm, e->s_comm);
    if (hi <= lo) return ncclSuccess;
    const size_t bytes = (hi - lo) * sizeof(vt_hit);
    if (e->comm_rank != root) return R->Send(static_cast<const char*>(d_send) + lo * sizeof(vt_hit), bytes, ncclUint8, root, comm, e->s_comm);
    for (int r = 0; r < e->comm_size; ++r) {
        char* dst = static_cast<char*>(d_recv) + (uint64_t(r) * cap + lo) * sizeof(vt_hit);
        if (r == root) {
            const char* src = static_cast<const char*>(d_send) + lo * sizeof(vt_hit);
            if (src != dst && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, e->s_comm) != hipSuccess) return ncclUnhandledCudaError;
            continue;
        }
        const ncclResult_t rc = R->Recv(dst, bytes, ncclUint8, r, comm, e->s_comm);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

} // namespace

namespace vt {

void multi_release(vt_engine* e)
{
    if (e->comm) {
        if (RcclApi* R = rccl()) (void)R->CommDestroy(static_cast<ncclComm_t>(e->comm));
        e->comm = nullptr;
    }
    for (int b = 0; b < 2; ++b) {
        if (e->d_send[b]) (void)hipFree(e->d_send[b]);
        e->d_send[b] = nullptr; e->d_send_bytes[b] = 0;
        if (e->ev_sent[b]) (void)hipEventDestroy(e->ev_sent[b]);
        e->ev_sent[b] = nullptr;
    }
    for (int c = 0; c < kMaxGatherChunks; ++c) {
        if (e->ev_traced[c]) (void)hipEventDestroy(e->ev_traced[c]);
        e->ev_traced[c] = nullptr;
    }
    if (e->ev_g0) (void)hipEventDestroy(e->ev_g0);
    if (e->ev_g1) (void)hipEventDestroy(e->ev_g1);
    e->ev_g0 = e->ev_g1 = nullptr;
    e->gather_timed = false;
    if (e->s_comm) (void)hipStreamDestroy(e->s_comm);
    e->s_comm = nullptr;
}

// Host ray array -> contiguous shards -> one staging pipeline per device, side by side (one host thread each: the
// pipelines are synchronous per device).  Results land in the caller's array directly, so no collective is needed here.
int multi_trace_host(vt_scene* s, const vt_ray* rays, uint64_t n, void* out, size_t out_elem, bool any_hit)
{
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_trace: the scene's engine has been closed");
    if (!rays || !out) return fail(VT_ERR_INVALID_ARG, "vt_trace: NULL buffer");
    std::vector<vt_scene*> scenes{s};
    scenes.insert(scenes.end(), s->replicas.begin(), s->replicas.end());
    const int ndev = int(scenes.size());
    std::vector<int> rcs(size_t(ndev), VT_OK);
    std::vector<std::string> errs(static_cast<size_t>(ndev));
    std::vector<std::thread> workers;
    for (int g = 0; g < ndev; ++g) {
        uint64_t lo = 0, hi = 0;
        vt_shard_bounds(n, ndev, g, &lo, &hi);
        if (hi <= lo) continue;
        workers.emplace_back([&, g, lo, hi] {
            rcs[size_t(g)] = engine_trace_host(scenes[size_t(g)], rays + lo, hi - lo, static_cast<char*>(out) + lo * out_elem, out_elem, any_hit);
            if (rcs[size_t(g)] != VT_OK) errs[size_t(g)] = vt_last_error();      // the error string is thread-local
        });
    }
    for (std::thread& t : workers) t.join();
    for (int g = 0; g < ndev; ++g)
        if (rcs[size_t(g)] != VT_OK) return fail(rcs[size_t(g)], "device " + std::to_string(scenes[size_t(g)]->engine ? scenes[size_t(g)]->engine->device : -1) + ": " + errs[size_t(g)]);
    return VT_OK;
}

} // namespace vt

extern "C" {

int vt_engine_open_multi(const int* devices, int ndev, vt_engine** out)
{
    if (!out) return fail(VT_ERR_INVALID_ARG, "vt_engine_open_multi: out is NULL");
    *out = nullptr;
    if (!devices || ndev <= 0) return fail(VT_ERR_INVALID_ARG, "vt_engine_open_multi: no devices");
    // (test hook, dead without VT_ENABLE_TEST_HOOKS=1: with VT_TEST_ALLOW_DEVICE_ALIASES=1 a device may stand for several members of the group, so that the N > 1
    // control flow runs on a box with one GPU -- against tests/cpp/fake_rccl.cpp, real RCCL refuses such a group)
    const char* aliases = test_hook("VT_TEST_ALLOW_DEVICE_ALIASES");
    const bool allow_aliases = aliases && aliases[0] == '1';
    for (int a = 0; a < ndev && !allow_aliases; ++a)
        for (int b = a + 1; b < ndev; ++b)
            if (devices[a] == devices[b]) return fail(VT_ERR_INVALID_ARG, "vt_engine_open_multi: a device is listed twice");
    vt_engine* root = nullptr;
    int rc = vt_engine_open(devices[0], &root);
    if (rc != VT_OK) return rc;
    for (int k = 1; k < ndev; ++k) {
        vt_engine* p = nullptr;
        rc = vt_engine_open(devices[k], &p);
        if (rc != VT_OK) { vt_engine_close(root); return rc; }
        p->root = root;
        root->peers.push_back(p);
    }
    *out = root;
    return VT_OK;
}

int vt_engine_device_count(const vt_engine* e) { return e ? int(e->peers.size()) + 1 : 0; }

vt_engine* vt_engine_member(vt_engine* e, int g)
{
    if (!e || g < 0 || g > int(e->peers.size())) return nullptr;
    return g == 0 ? e : e->peers[size_t(g) - 1];
}

int vt_engine_device(const vt_engine* e, int g)
{
    if (!e || g < 0 || g > int(e->peers.size())) return -1;
    return g == 0 ? e->device : e->peers[size_t(g) - 1]->device;
}

// ---- device-resident shards + the gather --------------------------------------------------------------------------

int vt_trace_closest_gather_dev(vt_scene* s, const void* const* d_rays, uint64_t n, void* d_hits_root)
{
    if (!s) return fail(VT_ERR_INVALID_ARG, "vt_trace_closest_gather_dev: scene is NULL");
    if (!s->engine) return fail(VT_ERR_INVALID_ARG, "vt_trace_closest_gather_dev: the scene's engine has been closed");
    if (n == 0) return VT_OK;
    if (!d_rays || !d_hits_root) return fail(VT_ERR_INVALID_ARG, "vt_trace_closest_gather_dev: NULL argument");
    vt_engine* root = s->engine;
    if (root->root) return fail(VT_ERR_INVALID_ARG, "vt_trace_closest_gather_dev: call it on the group's root scene");
    int rc = ensure_group_comms(root);
    if (rc != VT_OK) return rc;
    RcclApi* R = rccl();
    const std::vector<vt_engine*> g = group_of(root);
    std::vector<vt_scene*> scenes{s};
    scenes.insert(scenes.end(), s->replicas.begin(), s->replicas.end());
    if (scenes.size() != g.size()) return fail(VT_ERR_INVALID_ARG, "vt_trace_closest_gather_dev: the scene is not replicated over the group");
    const int ndev = int(g.size());
    const uint64_t cap = vt_shard_capacity(n, ndev);
    const int buf = root->sched.next_buf();
    const bool buf_was_used = root->sched.sent_used[buf];

    // send buffers of the non-root devices (the root traces straight into its slice of the result: ncclGather in place,
    // sendbuff == recvbuff + rank * sendcount); growing one first waits for the gather that may still read it
    std::vector<void*> send(static_cast<size_t>(ndev), nullptr);
    send[0] = d_hits_root;
    for (int k = 0; k < ndev; ++k) {
        uint64_t lo = 0, hi = 0;
        vt_shard_bounds(n, ndev, k, &lo, &hi);
        if (hi > lo && !d_rays[k]) return fail(VT_ERR_INVALID_ARG, "vt_trace_closest_gather_dev: d_rays[" + std::to_string(k) + "] is NULL");
        if (k == 0) continue;
        vt_engine* e = g[size_t(k)];
        DeviceGuard guard(e->device);
        if (!guard.ok) return fail(VT_ERR_HIP, "vt_trace_closest_gather_dev: hipSetDevice failed");
        if (e->d_send_bytes[buf] < cap * sizeof(vt_hit)) {
            if (buf_was_used) VT_HIP(hipEventSynchronize(e->ev_sent[buf]));
            rc = ensure_bytes(&e->d_send[buf], &e->d_send_bytes[buf], cap * sizeof(vt_hit));
            if (rc != VT_OK) return rc;
        }
        send[size_t(k)] = e->d_send[buf];
    }

    // the batch, step by step, as gather_schedule.h plans it: per device [wait for the gather of two batches ago,] trace,
    // hand over to the communication stream; then ONE gather -- cap records from every device into the root's buffer,
    // shard g at record g * cap (ray order) --, all devices' calls in one group
    bool in_group = false;
    const int K = root->sched.effective_chunks(int(root->gather_chunks));
    for (const GatherStep& st : root->sched.plan(ndev, K)) {
        vt_engine* e = g[size_t(st.dev)];
        DeviceGuard guard(e->device);
        if (!guard.ok) { if (in_group) (void)R->GroupEnd(); return fail(VT_ERR_HIP, "vt_trace_closest_gather_dev: hipSetDevice failed"); }
        if (in_group && st.op != GatherOp::Gather) { in_group = false; VT_NCCL(R->GroupEnd()); }
        switch (st.op) {
        case GatherOp::WaitSent:
            VT_HIP(hipStreamWaitEvent(e->stream, e->ev_sent[st.buf], 0));
            break;
        case GatherOp::Trace: {
            // chunk st.chunk of this device's shard: records [clo, chi) of the shard, as far as the shard has rays
            uint64_t lo = 0, hi = 0, clo = 0, chi = 0;
            vt_shard_bounds(n, ndev, st.dev, &lo, &hi);
            gather_chunk_bounds(cap, K, st.chunk, &clo, &chi);
            const uint64_t have = hi - lo;
            if (clo < have) {
                const uint64_t m = std::min(chi, have) - clo;
                rc = engine_launch(scenes[size_t(st.dev)], static_cast<const char*>(d_rays[st.dev]) + clo * sizeof(vt_ray), m,
                                   static_cast<char*>(send[size_t(st.dev)]) + clo * sizeof(vt_hit), nullptr, nullptr, false, false, e->stream);
                if (rc != VT_OK) return rc;
            }
            break;
        }
        case GatherOp::RecordTraced:
            VT_HIP(hipEventRecord(e->ev_traced[st.chunk], e->stream));
            break;
        case GatherOp::WaitTraced:
            VT_HIP(hipStreamWaitEvent(e->s_comm, e->ev_traced[st.chunk], 0));
            break;
        case GatherOp::Gather: {
            // with timing on: the root's communication stream, from where the batch's first piece may start to behind its last
            // piece (the root receives every shard, so this is the gather's duration; vt_engine_last_gather_ms)
            if (root->timing && st.dev == 0 && st.chunk == 0) VT_HIP(hipEventRecord(root->ev_g0, root->s_comm));
            if (!in_group) { VT_NCCL(R->GroupStart()); in_group = true; }
            uint64_t clo = 0, chi = 0;
            gather_chunk_bounds(cap, K, st.chunk, &clo, &chi);
            const ncclResult_t r = move_part(R, e, send[size_t(st.dev)], d_hits_root, cap, clo, chi, K == 1, 0);
            if (r != ncclSuccess) { (void)R->GroupEnd(); return rccl_fail("gather", r); }
            break;
        }
        case GatherOp::RecordSent:
            VT_HIP(hipEventRecord(e->ev_sent[st.buf], e->s_comm));
            if (root->timing && st.dev == 0) { VT_HIP(hipEventRecord(root->ev_g1, root->s_comm)); root->gather_timed = true; }
            break;
        }
    }
    if (in_group) VT_NCCL(R->GroupEnd());
    return VT_OK;
}

// ---- one process per GPU ---------------------------------------------------------------------------------------

int vt_comm_unique_id(void* id128)
{
    if (!id128) return fail(VT_ERR_INVALID_ARG, "vt_comm_unique_id: NULL");
    RcclApi* R = rccl();
    if (!R) return fail(VT_ERR_HIP, "vt_comm_unique_id: " + rccl_state().error);
    static_assert(sizeof(ncclUniqueId) == 128, "vt_comm_unique_id hands out 128 bytes");
    ncclUniqueId id;
    VT_NCCL(R->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return VT_OK;
}

int vt_engine_comm_init_rank(vt_engine* e, int nranks, int rank, const void* id128)
{
    if (!e || !id128) return fail(VT_ERR_INVALID_ARG, "vt_engine_comm_init_rank: NULL");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(VT_ERR_INVALID_ARG, "vt_engine_comm_init_rank: bad rank");
    if (!e->peers.empty() || e->root) return fail(VT_ERR_INVALID_ARG, "vt_engine_comm_init_rank: the engine belongs to a single-process group");
    if (e->comm) return fail(VT_ERR_INVALID_ARG, "vt_engine_comm_init_rank: already initialised");
    RcclApi* R = rccl();
    if (!R) return fail(VT_ERR_HIP, "vt_engine_comm_init_rank: " + rccl_state().error);
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_engine_comm_init_rank: hipSetDevice failed");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    VT_NCCL(R->CommInitRank(&comm, nranks, id, rank));
    e->comm = comm;
    e->comm_rank = rank;
    e->comm_size = nranks;
    return ensure_comm_side(e);
}

void vt_gather_chunk_bounds(uint64_t count, int nchunks, int chunk, uint64_t* lo, uint64_t* hi)
{
    uint64_t a = 0, b = 0;
    if (chunk >= 0 && chunk < std::max(nchunks, 1)) gather_chunk_bounds(count, nchunks, chunk, &a, &b);
    if (lo) *lo = a;
    if (hi) *hi = b;
}

int vt_gather_hits_part_dev(vt_engine* e, const void* d_send, uint64_t count, int chunk, int nchunks, void* d_recv_root, int root, void* stream_)
{
    if (!e || !d_send) return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_dev: NULL");
    if (!e->comm || !e->peers.empty() || e->root) return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_dev: call vt_engine_comm_init_rank first");
    if (root < 0 || root >= e->comm_size) return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_dev: bad root");
    if (e->comm_rank == root && !d_recv_root) return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_dev: the root needs a receive buffer");
    if (nchunks < 1 || nchunks > kMaxGatherChunks || chunk < 0 || chunk >= nchunks)
        return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_part_dev: chunk / nchunks out of range (1 .. 16 pieces)");
    // an empty batch has no pieces to order and nothing to send (every rank passes the same count): all its calls are no-ops
    if (count == 0) return VT_OK;
    RcclApi* R = rccl();
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_gather_hits_dev: hipSetDevice failed");
    // The pieces of one batch come in order, every rank with the same nchunks: the batch's steps are planned at its first piece.
    // Piece 0 always starts a NEW batch: a batch the caller abandoned between two pieces (its own trace of piece c failed, say)
    // is dropped here -- the pieces it did hand over are on the communication stream, so its `sent` event is recorded behind them
    // and the next user of that send buffer waits for them as for a whole batch.
    if (chunk == 0 && !e->part_steps.empty()) {
        const int stale_buf = e->part_steps.back().buf;
        e->part_steps.clear(); e->part_next = 0; e->part_chunks = 0;
        VT_HIP(hipEventRecord(e->ev_sent[stale_buf], e->s_comm));
    }
    if (chunk != 0 && (e->part_steps.empty() || e->part_next != chunk || e->part_chunks != nchunks))
        return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_part_dev: the pieces of a batch must be handed over in order, 0 .. nchunks - 1 "
                                        "(piece 0 starts a new batch and drops an unfinished one)");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (chunk == 0) {
        if (e->sched.effective_chunks(nchunks) != nchunks)
            return fail(VT_ERR_INVALID_ARG, "vt_gather_hits_part_dev: engine option gather_overlap = 0 allows one piece per batch only");
        e->part_steps = e->sched.plan(1, nchunks);
        e->part_chunks = nchunks;
    }
    e->part_next = chunk + 1;
    uint64_t lo = 0, hi = 0;
    gather_chunk_bounds(count, nchunks, chunk, &lo, &hi);
    // The records were produced on `stream`; the transfer runs on the engine's comm stream so that the next trace on `stream`
    // does not queue behind it.  Steps as gather_schedule.h plans them for one rank; the wait and the trace in front of them
    // are the caller's (vt_gather_wait + its own launch).
    int rc = VT_OK;
    for (const GatherStep& st : e->part_steps) {
        if (st.chunk != chunk || rc != VT_OK) continue;
        switch (st.op) {
        case GatherOp::WaitSent: case GatherOp::Trace: break;
        case GatherOp::RecordTraced: if (hipEventRecord(e->ev_traced[chunk], stream) != hipSuccess) rc = fail(VT_ERR_HIP, "vt_gather_hits_dev: hipEventRecord failed"); break;
        case GatherOp::WaitTraced:   if (hipStreamWaitEvent(e->s_comm, e->ev_traced[chunk], 0) != hipSuccess) rc = fail(VT_ERR_HIP, "vt_gather_hits_dev: hipStreamWaitEvent failed"); break;
        case GatherOp::Gather: {
            if (e->timing && chunk == 0) (void)hipEventRecord(e->ev_g0, e->s_comm);     // around ALL pieces of the batch
            ncclResult_t r = nchunks == 1 ? ncclSuccess : R->GroupStart();
            if (r == ncclSuccess) r = move_part(R, e, d_send, d_recv_root, count, lo, hi, nchunks == 1, root);
            if (nchunks != 1) { const ncclResult_t r2 = R->GroupEnd(); if (r == ncclSuccess) r = r2; }
            if (r != ncclSuccess) rc = rccl_fail("gather", r);
            if (e->timing && chunk + 1 == nchunks) { (void)hipEventRecord(e->ev_g1, e->s_comm); e->gather_timed = true; }
            break;
        }
        case GatherOp::RecordSent:   if (hipEventRecord(e->ev_sent[st.buf], e->s_comm) != hipSuccess) rc = fail(VT_ERR_HIP, "vt_gather_hits_dev: hipEventRecord failed"); break;
        }
    }
    if (chunk + 1 == nchunks || rc != VT_OK) { e->part_steps.clear(); e->part_next = 0; e->part_chunks = 0; }
    return rc;
}

int vt_gather_hits_dev(vt_engine* e, const void* d_send, uint64_t count, void* d_recv_root, int root, void* stream)
{
    return vt_gather_hits_part_dev(e, d_send, count, 0, 1, d_recv_root, root, stream);
}

int vt_engine_last_gather_ms(vt_engine* e, float* ms)
{
    if (!e || !ms) return fail(VT_ERR_INVALID_ARG, "vt_engine_last_gather_ms: NULL");
    if (!e->gather_timed) return fail(VT_ERR_INVALID_ARG, "vt_engine_last_gather_ms: no timed gather yet (vt_engine_set_timing)");
    DeviceGuard guard(e->device);
    VT_HIP(hipEventSynchronize(e->ev_g1));
    VT_HIP(hipEventElapsedTime(ms, e->ev_g0, e->ev_g1));
    return VT_OK;
}

int vt_gather_wait(vt_engine* e, int batches_in_flight, void* stream_)
{
    if (!e) return fail(VT_ERR_INVALID_ARG, "vt_gather_wait: NULL");
    if (!e->s_comm || e->sched.batches == 0) return VT_OK;
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(VT_ERR_HIP, "vt_gather_wait: hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (batches_in_flight <= 0) {                       // everything: host wait
        VT_HIP(hipStreamSynchronize(e->s_comm));
        return VT_OK;
    }
    // allow one gather in flight: the one that read the send buffer the NEXT batch writes (two batches ago) must be over
    // before `stream` continues; with engine option gather_overlap = 0 the latest one too (diagnostic: no overlap)
    const int buf = e->sched.next_buf();
    if (e->sched.sent_used[buf]) VT_HIP(hipStreamWaitEvent(stream, e->ev_sent[buf], 0));
    if (!e->sched.overlap && e->sched.sent_used[buf ^ 1]) VT_HIP(hipStreamWaitEvent(stream, e->ev_sent[buf ^ 1], 0));
    return VT_OK;
}

} // extern "C"
