// test_gather_schedule.cpp -- the multi-GPU batch schedule (vistrace_amd/csrc/gather_schedule.h) on a simulated group.
//
// multi_gpu.hip executes the planner's steps with HIP streams, events and ncclGather; no multi-GPU node was available,
// so this program executes the SAME steps on a model of ndev devices x {trace stream, communication stream}:
//   * a stream runs its operations in order; which runnable stream advances next is drawn at random (every legal
//     interleaving of asynchronous streams is some such sequence);
//   * hipStreamWaitEvent blocks the stream until the event's latest record (at enqueue time) has executed;
//   * the collective completes on a device only once every device has reached its call (ncclGather);
//   * the per-rank form (one process per GPU, bench.py) runs the same steps, each process enqueueing its own device's.
// Checked while running: a Trace never starts while a Gather that reads its send buffer is unfinished (the double
// buffer), a Gather never starts before the Trace that fills its send buffer has finished (the hand-over), and with
// overlap on, some gather of batch b really runs beside a trace of batch b + 1.  A deliberately broken schedule (the
// wait for sent[buf] dropped) must trip the first check -- the checker checks itself.
// Built by `make -C tests/cpp schedule`; exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <random>
#include <vector>

#include "gather_schedule.h"

using vt::GatherOp;
using vt::GatherSchedule;
using vt::GatherStep;

namespace {

struct Op {
    GatherOp op;
    int dev, buf;
    uint64_t batch;
    long wait_gen = -1;     // WaitSent / WaitTraced: generation of the event that must have executed
};

struct Sim {
    int ndev;
    // streams: [dev][0] = trace, [dev][1] = communication
    std::vector<std::deque<Op>> q;
    // events: records enqueued / executed so far (generation counters), per device
    std::vector<long> traced_enq, traced_done;
    std::vector<long> sent_enq[2], sent_done[2];
    // buffers: per device and buffer, the batch whose trace last wrote it / whose gather is reading it
    std::vector<long> written_by[2];       // batch that last finished a trace into the buffer (-1 none)
    std::vector<int>  readers[2];          // gathers in flight that read the buffer
    // collective rendezvous: devices that have reached the gather of batch b
    std::vector<int> arrived;              // indexed by batch
    std::vector<int> gather_started;       // per device: batch whose gather has started but not completed (-1 none)
    int hazards_rewrite = 0, hazards_early = 0, overlaps = 0;
    std::vector<long> tracing;             // per device: batch being traced "now" (a Trace op occupies one tick)

    explicit Sim(int n) : ndev(n), q(size_t(n) * 2), traced_enq(n, 0), traced_done(n, 0), arrived(), gather_started(n, -1), tracing(n, -1)
    {
        for (int b = 0; b < 2; ++b) {
            sent_enq[b].assign(n, 0); sent_done[b].assign(n, 0);
            written_by[b].assign(n, -1); readers[b].assign(n, 0);
        }
    }

    // enqueue = what the host thread does when it walks the planner's steps (events capture their generation here)
    void enqueue(const std::vector<GatherStep>& steps, uint64_t batch)
    {
        if (arrived.size() <= batch) arrived.resize(batch + 1, 0);
        for (const GatherStep& s : steps) {
            Op o{s.op, s.dev, s.buf, batch, -1};
            switch (s.op) {
            case GatherOp::WaitSent:     o.wait_gen = sent_enq[s.buf][s.dev]; q[size_t(s.dev) * 2 + 0].push_back(o); break;
            case GatherOp::Trace:        q[size_t(s.dev) * 2 + 0].push_back(o); break;
            case GatherOp::RecordTraced: ++traced_enq[s.dev]; o.wait_gen = traced_enq[s.dev]; q[size_t(s.dev) * 2 + 0].push_back(o); break;
            case GatherOp::WaitTraced:   o.wait_gen = traced_enq[s.dev]; q[size_t(s.dev) * 2 + 1].push_back(o); break;
            case GatherOp::Gather:       q[size_t(s.dev) * 2 + 1].push_back(o); break;
            case GatherOp::RecordSent:   ++sent_enq[s.buf][s.dev]; o.wait_gen = sent_enq[s.buf][s.dev]; q[size_t(s.dev) * 2 + 1].push_back(o); break;
            }
        }
    }

    // one attempt to advance stream `si`; returns true when an operation executed (or a gather made progress)
    bool advance(size_t si)
    {
        if (q[si].empty()) return false;
        Op& o = q[si].front();
        const int d = o.dev;
        switch (o.op) {
        case GatherOp::WaitSent:
            if (sent_done[o.buf][d] < o.wait_gen) return false;
            break;
        case GatherOp::Trace: {
            if (readers[o.buf][d] != 0) ++hazards_rewrite;                           // a gather still reads this send buffer
            if (gather_started[d] >= 0 && uint64_t(gather_started[d]) + 1 == o.batch) ++overlaps;   // beside this device's previous gather
            written_by[o.buf][d] = long(o.batch);
            break;
        }
        case GatherOp::RecordTraced:
            traced_done[d] = o.wait_gen;
            break;
        case GatherOp::WaitTraced:
            if (traced_done[d] < o.wait_gen) return false;
            break;
        case GatherOp::Gather:
            if (gather_started[d] != long(o.batch)) {                                 // first visit: the device reaches its call
                if (written_by[o.buf][d] != long(o.batch)) ++hazards_early;           // its send buffer is not this batch's yet
                gather_started[d] = long(o.batch);
                ++readers[o.buf][d];
                ++arrived[o.batch];
                return true;
            }
            if (arrived[o.batch] < ndev) return false;                                // the collective waits for every device
            --readers[o.buf][d];
            gather_started[d] = -1;
            break;
        case GatherOp::RecordSent:
            sent_done[o.buf][d] = o.wait_gen;
            break;
        }
        q[si].pop_front();
        return true;
    }

    bool idle() const
    {
        for (const auto& s : q) if (!s.empty()) return false;
        return true;
    }
};

int fails = 0;
#define CHECK(c) do { if (!(c)) { ++fails; std::printf("FAIL line %d: %s\n", __LINE__, #c); } } while (0)

struct Outcome { int rewrite, early, overlaps; bool deadlock; };

// `batches` batches through a group of ndev devices; the host enqueues batch b + 1 at a random moment (it never waits
// for the device: the ABI call is asynchronous).  break_wait drops the WaitSent steps (negative control).
Outcome run(int ndev, int batches, unsigned seed, bool overlap, bool break_wait)
{
    std::mt19937 rng(seed);
    GatherSchedule sched;
    sched.overlap = overlap;
    Sim sim(ndev);
    int enq = 0;
    long guard = 0;
    while (enq < batches || !sim.idle()) {
        if (enq < batches && (sim.idle() || rng() % 4 == 0)) {
            std::vector<GatherStep> steps = sched.plan(ndev);
            if (break_wait) {
                std::vector<GatherStep> kept;
                for (const GatherStep& s : steps) if (s.op != GatherOp::WaitSent) kept.push_back(s);
                steps.swap(kept);
            }
            sim.enqueue(steps, uint64_t(enq));
            ++enq;
            continue;
        }
        // a random stream that can advance; scanning from a random start models arbitrary relative speeds
        const size_t nstreams = sim.q.size(), start = rng() % nstreams;
        bool moved = false;
        for (size_t k = 0; k < nstreams && !moved; ++k) moved = sim.advance((start + k) % nstreams);
        if (!moved && enq >= batches) return {sim.hazards_rewrite, sim.hazards_early, sim.overlaps, true};
        if (!moved && enq < batches) {                       // everything blocked on work not yet enqueued
            sim.enqueue(sched.plan(ndev), uint64_t(enq));
            ++enq;
        }
        if (++guard > 50000000) return {sim.hazards_rewrite, sim.hazards_early, sim.overlaps, true};
    }
    return {sim.hazards_rewrite, sim.hazards_early, sim.overlaps, false};
}

} // namespace

int main()
{
    for (int ndev : {1, 2, 4, 8}) {
        int overlaps = 0;
        for (unsigned seed = 1; seed <= 200; ++seed) {
            const Outcome o = run(ndev, 24, seed * 7919u + unsigned(ndev), true, false);
            CHECK(!o.deadlock);
            CHECK(o.rewrite == 0);
            CHECK(o.early == 0);
            overlaps += o.overlaps;
        }
        CHECK(overlaps > 0);                                  // the gather of batch b does run beside the trace of batch b + 1
        // diagnostic mode: no overlap at all
        int ov_off = 0;
        for (unsigned seed = 1; seed <= 50; ++seed) {
            const Outcome o = run(ndev, 12, seed * 31u, false, false);
            CHECK(!o.deadlock && o.rewrite == 0 && o.early == 0);
            ov_off += o.overlaps;
        }
        CHECK(ov_off == 0);
        // negative control: without the wait for sent[buf] the simulator must see a send buffer re-written under a gather
        if (ndev > 1) {
            int seen = 0;
            for (unsigned seed = 1; seed <= 200; ++seed) seen += run(ndev, 24, seed * 13u, true, true).rewrite;
            CHECK(seen > 0);
        }
        std::printf("ndev %d: ok (overlapping trace/gather pairs observed: %d)\n", ndev, overlaps);
    }
    std::printf(fails ? "%d checks FAILED\n" : "gather schedule: all checks passed\n", fails);
    return fails ? 1 : 0;
}
