// gather_schedule.h -- the order of traces, event waits and gathers of the multi-GPU path, as data.
//
// multi_gpu.hip executes the steps this planner emits with HIP streams / events and RCCL calls; tests/cpp/
// test_gather_schedule.cpp executes the SAME steps on a simulated device set with randomly interleaved streams and
// checks the hazards of the scheme: a send buffer is never re-written before the gather that read it has finished, a
// gather never starts before the trace that fills the part it sends, and the parts of one batch do overlap.  Host-only,
// no HIP types.
//
// Per device: a trace stream and a communication stream; events `traced[c]` (one per chunk) and `sent[0..1]`.
//
// Between batches: batch b uses send buffer b & 1 on every device (the root's "send buffer" is its slice of the caller's
// result array: the gather is in place -- a caller that alternates two result arrays gets the same protection), so the
// gather of batch b (communication streams) runs beside the traces of batch b + 1 (trace streams) and batch b + 2 is the
// first to touch buffer b & 1 again -- behind a wait for sent[b & 1].
//
// Inside ONE batch (round 4): a device's shard is cut into `chunks` contiguous pieces.  Chunk c is traced, handed to the
// communication stream (traced[c]) and gathered while chunk c + 1 is being traced, so a single batch costs about
// max(trace, gather) + one chunk instead of trace + gather -- what decides a one-shot call like BASELINE configs[4]
// (one 128 Mi-ray batch over 8 GPUs: 3.5 ms of tracing and >= 4.2 ms on the wire per device).  chunks = 1 is the plain
// scheme: one gather per batch.
#pragma once

#include <cstdint>
#include <vector>

#ifndef VT_MUT   // mutation sites (vt_internal.h; scripts/mutants_host.sh builds the schedule test with -DVT_MUTANT=83)
#ifdef VT_MUTANT
#define VT_MUT(k, wrong, right) ((VT_MUTANT == (k)) ? (wrong) : (right))
#else
#define VT_MUT(k, wrong, right) (right)
#endif
#endif

namespace vt {

enum class GatherOp : uint8_t {
    WaitSent,     // trace stream of `dev` waits for event sent[buf] of `dev`        (hipStreamWaitEvent)
    Trace,        // trace stream of `dev`: trace chunk `chunk` of the device's shard into its send buffer `buf` (root: into the result)
    RecordTraced, // trace stream of `dev`: record event traced[chunk]                (hipEventRecord)
    WaitTraced,   // communication stream of `dev` waits for event traced[chunk]      (hipStreamWaitEvent)
    Gather,       // communication stream of `dev`: this device's call(s) of the collective that moves chunk `chunk` (reads buffer `buf`)
    RecordSent,   // communication stream of `dev`: record event sent[buf] (behind the batch's last chunk)
};

struct GatherStep {
    GatherOp op;
    int      dev;    // position in the group, 0 = root
    int      buf;    // send buffer / sent event index, 0 or 1
    int      chunk;  // piece of the shard, 0 .. chunks - 1
};

constexpr int kMaxGatherChunks = 16;

// State that survives between batches: how many batches were issued and which sent events have ever been recorded.
struct GatherSchedule {
    uint64_t batches = 0;
    bool     sent_used[2] = {false, false};
    bool     overlap = true;    // false: every trace also waits for the PREVIOUS batch's gather, and a batch is one chunk
                                // (no overlap of any kind: a diagnostic, step = trace + gather)

    int next_buf() const { return int(batches & 1); }

    // chunks a batch is really cut into: the request, clamped; 1 in the diagnostic mode
    int effective_chunks(int chunks) const
    {
        if (!overlap) return 1;
        return chunks < 1 ? 1 : (chunks > kMaxGatherChunks ? kMaxGatherChunks : chunks);
    }

    // The steps of the next batch for a group of `ndev` devices (single-process form: all devices; per-rank form:
    // ndev = 1 describes this rank's own two streams), its shards cut into `chunks` pieces.
    std::vector<GatherStep> plan(int ndev, int chunks = 1)
    {
        std::vector<GatherStep> s;
        const int buf = next_buf();
        const int K = effective_chunks(chunks);
        for (int c = 0; c < K; ++c) {
            for (int k = 0; k < ndev; ++k) {
                if (c == 0) {
                    // the gather that read this send buffer two batches ago must be over before the buffer is written again
                    if (sent_used[buf]) s.push_back({GatherOp::WaitSent, k, buf, 0});
                    if (!overlap && sent_used[buf ^ 1]) s.push_back({GatherOp::WaitSent, k, buf ^ 1, 0});
                }
                s.push_back({GatherOp::Trace, k, buf, c});
                s.push_back({GatherOp::RecordTraced, k, buf, c});
                s.push_back({GatherOp::WaitTraced, k, buf, c});
            }
            for (int k = 0; k < ndev; ++k) s.push_back({GatherOp::Gather, k, buf, c});   // one group: all devices' calls together
        }
        for (int k = 0; k < ndev; ++k) s.push_back({GatherOp::RecordSent, k, buf, K - 1});
        sent_used[buf] = true;
        ++batches;
        return s;
    }
};

// Chunk c of a shard of `cap` records cut into K pieces: records [lo, hi) of the shard.  Pieces are multiples of 64 records
// (a wave's worth) except the last; trailing chunks may be empty when cap is small.
inline void gather_chunk_bounds(uint64_t cap, int K, int c, uint64_t* lo, uint64_t* hi)
{
    if (K < 1) K = 1;
    const uint64_t per = ((cap + uint64_t(K) - 1) / uint64_t(K) + 63) / 64 * 64;
    const uint64_t a = per * uint64_t(c), b = a + per;
    *lo = a < cap ? a : cap;
    *hi = VT_MUT(83, b, b < cap ? b : cap);
}

} // namespace vt
