// gather_schedule.h -- the order of traces, event waits and gathers of the multi-GPU path, as data.
//
// multi_gpu.hip executes the steps this planner emits with HIP streams / events and ncclGather; tests/cpp/
// test_gather_schedule.cpp executes the SAME steps on a simulated device set with randomly interleaved streams and
// checks the two hazards of the scheme: a send buffer is never re-written before the gather that read it has
// finished, and a gather never starts before the trace that fills its send buffer.  Host-only, no HIP types.
//
// Per device: a trace stream and a communication stream; events `traced` and `sent[0..1]`.  Batch b uses send buffer
// b & 1 on every device (the root's "send buffer" is its slice of the caller's result array: ncclGather in place --
// a caller that alternates two result arrays gets the same protection), so the gather of batch b (communication
// streams) runs beside the traces of batch b + 1 (trace streams) and batch b + 2 is the first to touch buffer b & 1
// again -- behind a wait for sent[b & 1].
#pragma once

#include <cstdint>
#include <vector>

namespace vt {

enum class GatherOp : uint8_t {
    WaitSent,     // trace stream of `dev` waits for event sent[buf] of `dev`        (hipStreamWaitEvent)
    Trace,        // trace stream of `dev`: trace the device's shard into its send buffer `buf` (root: into the result)
    RecordTraced, // trace stream of `dev`: record event traced                       (hipEventRecord)
    WaitTraced,   // communication stream of `dev` waits for event traced             (hipStreamWaitEvent)
    Gather,       // communication stream of `dev`: this device's call of the collective (reads send buffer `buf`)
    RecordSent,   // communication stream of `dev`: record event sent[buf]
};

struct GatherStep {
    GatherOp op;
    int      dev;   // position in the group, 0 = root
    int      buf;   // send buffer / sent event index, 0 or 1
};

// State that survives between batches: how many batches were issued and which sent events have ever been recorded.
struct GatherSchedule {
    uint64_t batches = 0;
    bool     sent_used[2] = {false, false};
    bool     overlap = true;    // false: every trace also waits for the PREVIOUS batch's gather (no overlap: a diagnostic)

    int next_buf() const { return int(batches & 1); }

    // The steps of the next batch for a group of `ndev` devices (single-process form: all devices; per-rank form:
    // ndev = 1 describes this rank's own two streams).
    std::vector<GatherStep> plan(int ndev)
    {
        std::vector<GatherStep> s;
        const int buf = next_buf();
        for (int k = 0; k < ndev; ++k) {
            // the gather that read this send buffer two batches ago must be over before the buffer is written again
            if (sent_used[buf]) s.push_back({GatherOp::WaitSent, k, buf});
            if (!overlap && sent_used[buf ^ 1]) s.push_back({GatherOp::WaitSent, k, buf ^ 1});
            s.push_back({GatherOp::Trace, k, buf});
            s.push_back({GatherOp::RecordTraced, k, buf});
            s.push_back({GatherOp::WaitTraced, k, buf});
        }
        for (int k = 0; k < ndev; ++k) s.push_back({GatherOp::Gather, k, buf});      // one group: all devices' calls together
        for (int k = 0; k < ndev; ++k) s.push_back({GatherOp::RecordSent, k, buf});
        sent_used[buf] = true;
        ++batches;
        return s;
    }
};

} // namespace vt
