// fake_rccl.cpp -- a TEST DOUBLE for the ten RCCL entry points vistrace_amd/csrc/multi_gpu.hip binds (loaded instead of
// librccl.so through VT_RCCL_LIB).  It lets the N > 1 control flow of the product -- vt_engine_open_multi, scene replication,
// the per-device threads of the host path, vt_trace_closest_gather_dev's batch schedule with its double-buffered send buffers
// and K pieces per batch, and the one-process-per-GPU form (vt_engine_comm_init_rank + vt_gather_hits[_part]_dev, one thread per
// rank) -- run on a box with ONE GPU: every "device" of the group is device 0 (VT_TEST_ALLOW_DEVICE_ALIASES=1), and a transfer
// between two ranks is a device-to-device copy ordered between the two ranks' streams exactly as a send / receive pair is:
//
//   sender's stream:    ... work before the send | (the send completes when the receiver's copy has finished) | work after
//   receiver's stream:  ... work before the recv | wait for the sender's "ready" event, copy, record "done"   | work after
//
// What it does NOT model: RCCL's own kernels (their CUs, the xGMI links), ranks in other processes, error injection.
// Semantics kept: ncclGather = the sends / receives it is made of, root's own contribution copied unless in place; calls inside
// ncclGroupStart / ncclGroupEnd take effect at the outermost ncclGroupEnd; a receive matches the oldest unmatched send of
// its (source, destination) pair and must have the same size.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Transfer {
    const void* src = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;      // recorded on the sender's stream behind the work that produced src
    hipEvent_t done = nullptr;       // recorded on the receiver's stream behind its copy
    bool consumed = false;
    bool failed = false;
};

struct Group {
    int nranks = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::pair<int, int>, std::deque<std::shared_ptr<Transfer>>> box;   // (from, to) -> posted sends, oldest first
    long transfers = 0;
};

std::mutex g_registry_mu;
std::map<std::string, std::weak_ptr<Group>> g_registry;    // ncclCommInitRank: unique id -> group
long g_next_id = 1;

} // namespace

struct ncclComm {
    std::shared_ptr<Group> group;
    int rank = 0;
    int device = 0;
};

namespace {

struct Op {
    bool is_send;
    ncclComm* comm;
    const void* src;      // send
    void* dst;            // recv
    size_t bytes;
    int peer;
    hipStream_t stream;
    std::shared_ptr<Transfer> t;    // send: what was posted
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_pending;

struct Dev {
    int prev = -1;
    explicit Dev(int d) { (void)hipGetDevice(&prev); if (prev != d) (void)hipSetDevice(d); }
    ~Dev() { int cur = -1; (void)hipGetDevice(&cur); if (prev >= 0 && cur != prev) (void)hipSetDevice(prev); }
};

// FAKE_RCCL_RECV_DELAY_US: every receive first holds its stream for that long (a host function on the stream), so that a
// transfer is still in flight when the caller's later work is enqueued -- what a real link does to a 268 MB shard.
// FAKE_RCCL_FAULT=early_send_completion: a send no longer holds its stream until the data has left.  With the delay this
// makes premature reuse of a send buffer visible; the test uses it as its negative control.
long recv_delay_us()
{
    static const long us = [] { const char* e = std::getenv("FAKE_RCCL_RECV_DELAY_US"); return e ? std::atol(e) : 0L; }();
    return us;
}
bool early_send_completion()
{
    static const bool on = [] { const char* e = std::getenv("FAKE_RCCL_FAULT"); return e && std::strcmp(e, "early_send_completion") == 0; }();
    return on;
}
void sleep_on_stream(void*) { std::this_thread::sleep_for(std::chrono::microseconds(recv_delay_us())); }

ncclResult_t run(std::vector<Op>& ops)
{
    ncclResult_t result = ncclSuccess;
    // phase 1: post every send (its data is ready behind what its stream holds now)
    for (Op& op : ops) {
        if (!op.is_send) continue;
        Dev dev(op.comm->device);
        auto t = std::make_shared<Transfer>();
        t->src = op.src; t->bytes = op.bytes;
        if (hipEventCreateWithFlags(&t->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t->done, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(t->ready, op.stream) != hipSuccess)
            return ncclUnhandledCudaError;
        op.t = t;
        Group& g = *op.comm->group;
        std::lock_guard<std::mutex> lock(g.mu);
        g.box[{op.comm->rank, op.peer}].push_back(t);
        g.cv.notify_all();
    }
    // phase 2: every receive takes the oldest send of its pair (waiting for a sender on another thread)
    for (Op& op : ops) {
        if (op.is_send) continue;
        Group& g = *op.comm->group;
        std::shared_ptr<Transfer> t;
        {
            std::unique_lock<std::mutex> lock(g.mu);
            auto& q = g.box[{op.peer, op.comm->rank}];
            // a receive whose send never comes is a bug in the caller: fail the call instead of hanging the test
            if (!g.cv.wait_for(lock, std::chrono::seconds(30), [&] { return !q.empty(); })) {
                std::fprintf(stderr, "fake_rccl: rank %d waited 30 s for a send from rank %d\n", op.comm->rank, op.peer);
                result = ncclInternalError;
                continue;
            }
            t = q.front();
            q.pop_front();
        }
        Dev dev(op.comm->device);
        if (recv_delay_us() > 0) (void)hipLaunchHostFunc(op.stream, sleep_on_stream, nullptr);   // the link is slow: copies lag behind
        bool ok = t->bytes == op.bytes;
        if (!ok) std::fprintf(stderr, "fake_rccl: rank %d receives %zu bytes from rank %d, which sent %zu\n", op.comm->rank, op.bytes, op.peer, t->bytes);
        ok = ok && hipStreamWaitEvent(op.stream, t->ready, 0) == hipSuccess;
        ok = ok && hipMemcpyAsync(op.dst, t->src, op.bytes, hipMemcpyDeviceToDevice, op.stream) == hipSuccess;
        ok = ok && hipEventRecord(t->done, op.stream) == hipSuccess;
        {
            std::lock_guard<std::mutex> lock(g.mu);
            t->consumed = true;
            t->failed = !ok;
            ++g.transfers;
            g.cv.notify_all();
        }
        if (!ok) result = ncclInvalidArgument;
    }
    // phase 3: a send is complete on its stream when the receiver's copy is
    for (Op& op : ops) {
        if (!op.is_send) continue;
        Group& g = *op.comm->group;
        {
            std::unique_lock<std::mutex> lock(g.mu);
            if (!g.cv.wait_for(lock, std::chrono::seconds(30), [&] { return op.t->consumed; })) {
                std::fprintf(stderr, "fake_rccl: rank %d waited 30 s for rank %d to receive\n", op.comm->rank, op.peer);
                result = ncclInternalError;
                continue;
            }
        }
        Dev dev(op.comm->device);
        if (op.t->failed) result = ncclInvalidArgument;
        else if (early_send_completion()) {}     // injected fault: the sender's stream runs on before its data has left
        else if (hipStreamWaitEvent(op.stream, op.t->done, 0) != hipSuccess) result = ncclUnhandledCudaError;
        (void)hipEventDestroy(op.t->ready);      // released once the work that uses them has run
        (void)hipEventDestroy(op.t->done);
    }
    ops.clear();
    return result;
}

ncclResult_t submit(Op op)
{
    t_pending.push_back(op);
    if (t_depth > 0) return ncclSuccess;
    return run(t_pending);
}

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
    }
}

} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::memset(id, 0, sizeof(*id));
    std::lock_guard<std::mutex> lock(g_registry_mu);
    std::snprintf(id->internal, sizeof(id->internal), "fake-rccl-%ld", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const std::string key(id.internal, sizeof(id.internal));
    std::shared_ptr<Group> g;
    {
        std::lock_guard<std::mutex> lock(g_registry_mu);
        g = g_registry[key].lock();
        if (!g) { g = std::make_shared<Group>(); g->nranks = nranks; g_registry[key] = g; }
    }
    if (g->nranks != nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->group = g; c->rank = rank;
    (void)hipGetDevice(&c->device);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
    if (!comms || ndev <= 0) return ncclInvalidArgument;
    auto g = std::make_shared<Group>();
    g->nranks = ndev;
    for (int k = 0; k < ndev; ++k) {
        ncclComm* c = new ncclComm();
        c->group = g; c->rank = k; c->device = devlist ? devlist[k] : k;
        comms[k] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++t_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run(t_pending);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->group->nranks || peer == comm->rank) return ncclInvalidArgument;
    return submit(Op{true, comm, sendbuff, nullptr, count * type_bytes(datatype), peer, stream, nullptr});
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->group->nranks || peer == comm->rank) return ncclInvalidArgument;
    return submit(Op{false, comm, nullptr, recvbuff, count * type_bytes(datatype), peer, stream, nullptr});
}

ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || root < 0 || root >= comm->group->nranks) return ncclInvalidArgument;
    const size_t bytes = sendcount * type_bytes(datatype);
    if (comm->rank != root) return ncclSend(sendbuff, sendcount, datatype, root, comm, stream);
    ++t_depth;                                              // the root's receives form one group
    ncclResult_t r = ncclSuccess;
    for (int k = 0; k < comm->group->nranks && r == ncclSuccess; ++k) {
        char* dst = static_cast<char*>(recvbuff) + size_t(k) * bytes;
        if (k == root) {
            if (dst != sendbuff) {
                Dev dev(comm->device);
                if (hipMemcpyAsync(dst, sendbuff, bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) r = ncclUnhandledCudaError;
            }
        } else {
            r = ncclRecv(dst, sendcount, datatype, k, comm, stream);
        }
    }
    const ncclResult_t r2 = ncclGroupEnd();
    return r != ncclSuccess ? r : r2;
}

const char* ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
    case ncclInvalidArgument: return "invalid argument (fake RCCL)";
    case ncclInvalidUsage: return "invalid usage (fake RCCL)";
    default: return "error (fake RCCL)";
    }
}

// test introspection: transfers completed in the group of `comm`
long fakeRcclTransfers(ncclComm_t comm)
{
    if (!comm) return -1;
    std::lock_guard<std::mutex> lock(comm->group->mu);
    return comm->group->transfers;
}

} // extern "C"
