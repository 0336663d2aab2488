// sim_waves.c -- CPU model of how a 64-lane wave executes the traversal kernel's NODE / TRI passes (dev tool).
//
// Counts wave-level passes and lane utilisation for scheduling policies of the persistent kernel, on the real tree and
// real rays, with the oracle's own slab and triangle arithmetic (this file includes oracle/vt_oracle.c: analysis
// tooling, never linked into the product).  Every policy must reproduce the oracle's hits exactly -- the program checks.
//
//   policy 0  the shipped kernel: one ray per lane, pending leaf range drained before the next node step, TRI pass when
//             >= tri_threshold lanes wait (or nobody can step), lanes re-filled when >= refill idle
//   policy 1  deferred triangle tests: a lane that finds a leaf queues it {first, second, range} and keeps walking with
//             its current tmax; queued leaves are tested in order in dense TRI passes (fired when >= fire lanes hold
//             work or a lane is blocked), each entry re-checked against the tmax of that moment (first <= min(second,
//             tmax)) -- exact by the containment argument in profiles/r2/notes.md
//   policy 2  the form a kernel can hold in registers: the live pending range as today + ONE parked group (the sibling
//             leaves of one step); a lane that tested does not step in the same iteration (one record per lane)
//
//   gcc -O2 -fopenmp -ffp-contract=off scripts/sim_waves.c -o /tmp/sim_waves -lm && /tmp/sim_waves <dir with *.bin> <policy> [params]
#include "../oracle/vt_oracle.c"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

#define LANES 64
#define QCAP 16

typedef struct { float first, second; uint32_t cur, end; int checked, pair_with_next, dead; } qentry;

typedef struct {
    int has_ray; uint64_t idx;
    vto_ray ray; node_isect ni; float tmax;
    vto_hit best;
    uint32_t node; int node_done;          /* pair walk on the v1 tree: `node` = index of the LEFT child of the pair */
    uint32_t stack[256]; int sp;
    uint32_t tri_cur, tri_end;             /* policy 0: pending range (slots into prim_indices) */
    qentry q[QCAP]; int qh, qn;            /* policy 1: FIFO of queued leaves */
    int gvalid; float gfl, gfr; uint32_t gstart, gcl, gcr;   /* policy 2: ONE queued group (the sibling leaves of one step) behind the live head range */
} lane_t;

static void* load(const char* dir, const char* name, size_t* bytes)
{
    char p[512]; snprintf(p, sizeof p, "%s/%s", dir, name);
    FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(1); }
    fseek(f, 0, SEEK_END); *bytes = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    void* b = malloc(*bytes); if (fread(b, 1, *bytes, f) != *bytes) exit(1); fclose(f); return b;
}

static const vto_node* N; static const uint32_t* PI; static const vto_tri* T; static const vto_ray* R; static vto_hit* OUT;

static void start(lane_t* L, uint64_t idx)
{
    memset(L, 0, offsetof(lane_t, stack));
    L->has_ray = 1; L->idx = idx; L->ray = R[idx]; L->tmax = R[idx].tmax;
    node_isect_init(&L->ni, &L->ray);
    L->best.prim = VTO_MISS;
    L->node = N[0].first; L->node_done = 0; L->sp = 0; L->tri_cur = L->tri_end = 0; L->qh = L->qn = 0; L->gvalid = 0;
}

/* one triangle test in slot order; returns 1 on an accepted hit */
static int test_slot(lane_t* L, uint32_t slot)
{
    const uint32_t prim = PI[slot];
    float t, u, v;
    if (vto_tri_intersect(&T[prim], L->ray.org, L->ray.dir, L->ray.tmin, L->tmax, &t, &u, &v)) {
        L->best.prim = prim; L->best.t = t; L->best.u = u; L->best.v = v; L->tmax = t;
        return 1;
    }
    return 0;
}

/* one NODE step; leaves that were hit are reported through (lf, ls, lc, le) x 2 in visiting order */
static int node_step(lane_t* L, float* f, float* s, uint32_t* c, uint32_t* e)
{
    const vto_node* left = &N[L->node]; const vto_node* right = left + 1;
    float fl, sl, fr, sr;
    node_slab(&L->ni, left, L->ray.tmin, L->tmax, &fl, &sl);
    node_slab(&L->ni, right, L->ray.tmin, L->tmax, &fr, &sr);
    int nleaf = 0;
    int go_l = 0, go_r = 0;
    if (fl <= sl) { if (left->prim_count) { f[nleaf] = fl; s[nleaf] = sl; c[nleaf] = left->first; e[nleaf] = left->first + left->prim_count; ++nleaf; } else go_l = 1; }
    if (fr <= sr) { if (right->prim_count) { f[nleaf] = fr; s[nleaf] = sr; c[nleaf] = right->first; e[nleaf] = right->first + right->prim_count; ++nleaf; } else go_r = 1; }
    if (go_l && go_r) {
        const vto_node *a = left, *b = right;
        if (fl > fr) { a = right; b = left; }
        L->stack[L->sp++] = b->first; L->node = a->first;
    } else if (go_l) L->node = left->first;
    else if (go_r) L->node = right->first;
    else if (L->sp) L->node = L->stack[--L->sp];
    else L->node_done = 1;
    return nleaf;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: sim_waves <dir> <policy> [tri_threshold|fire] [refill] [qcap]\n"); return 2; }
    const char* dir = argv[1]; const int policy = atoi(argv[2]);
    const int p1 = argc > 3 ? atoi(argv[3]) : (policy == 0 ? 4 : 24);
    const int refill = argc > 4 ? atoi(argv[4]) : 8;
    const int qcap = argc > 5 ? atoi(argv[5]) : 4;
    size_t nb, pb, tb, rb;
    N = load(dir, "nodes.bin", &nb); PI = load(dir, "pidx.bin", &pb); T = load(dir, "tris.bin", &tb); R = load(dir, "rays.bin", &rb);
    const uint64_t nrays = rb / sizeof(vto_ray);
    OUT = calloc(nrays, sizeof(vto_hit));
    if (N[0].prim_count) { fprintf(stderr, "root leaf\n"); return 2; }

    const uint64_t per_wave = 4096;                 /* rays a wave works through (re-filling as lanes retire) */
    uint64_t iters = 0, node_passes = 0, tri_passes = 0, node_lanes = 0, tri_lanes = 0, steps = 0, tests = 0, retests_failed = 0, stalled = 0;
    uint64_t idle_lanes = 0, wait_lanes = 0;     /* lane-iterations without a ray / with a pending triangle while no TRI pass runs */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : iters, node_passes, tri_passes, node_lanes, tri_lanes, steps, tests, retests_failed, stalled, idle_lanes, wait_lanes)
    for (int64_t w = 0; w < (int64_t)((nrays + per_wave - 1) / per_wave); ++w) {
        lane_t* L = calloc(LANES, sizeof(lane_t));
        uint64_t cur = (uint64_t)w * per_wave, end = cur + per_wave < nrays ? cur + per_wave : nrays;
        for (;;) {
            int idle = 0, active = 0;
            for (int l = 0; l < LANES; ++l) idle += !L[l].has_ray;
            if (cur < end && (idle >= refill || idle == LANES))
                for (int l = 0; l < LANES && cur < end; ++l) if (!L[l].has_ray) start(&L[l], cur++);
            for (int l = 0; l < LANES; ++l) active += L[l].has_ray;
            if (!active) break;
            ++iters;
            idle_lanes += (uint64_t)(LANES - active);
            if (policy == 2) {
                /* the kernel design: live head range [tri_cur, tri_end) as today + ONE queued, unchecked group */
                int want_tri = 0, can_step = 0, blocked = 0;
                int stepl[LANES] = {0};
                for (int l = 0; l < LANES; ++l) {
                    lane_t* q = &L[l];
                    if (!q->has_ray) continue;
                    const int head = q->tri_cur < q->tri_end;
                    want_tri += head;
                    if (!q->node_done && !q->gvalid) { stepl[l] = 1; ++can_step; } else if (head) ++blocked;
                }
                const int fire = want_tri && (want_tri >= p1 || blocked * 4 >= active || can_step == 0);
                int tested[LANES] = {0};
                if (fire) {
                    ++tri_passes; tri_lanes += want_tri;
                    for (int l = 0; l < LANES; ++l) {
                        lane_t* q = &L[l];
                        if (!q->has_ray || !(q->tri_cur < q->tri_end)) continue;
                        tested[l] = 1;
                        ++tests; test_slot(q, q->tri_cur++);
                        if (q->tri_cur >= q->tri_end && q->gvalid) {     /* promote: the reference's slab test of these leaves saw THIS tmax */
                            const int live_l = q->gcl && q->gfl <= q->tmax, live_r = q->gcr && q->gfr <= q->tmax;
                            retests_failed += (q->gcl && !live_l) + (q->gcr && !live_r);
                            q->tri_cur = q->gstart + (live_l ? 0 : q->gcl);
                            q->tri_end = q->gstart + q->gcl + (live_r ? q->gcr : 0);
                            if (!live_l && !live_r) q->tri_cur = q->tri_end = 0;
                            q->gvalid = 0;
                        }
                    }
                }
                int did = 0;
                for (int l = 0; l < LANES; ++l) {
                    lane_t* q = &L[l];
                    if (!stepl[l] || tested[l]) continue;       /* one record per lane and iteration: a lane that tested does not step */
                    float f[2], s2[2]; uint32_t c[2], e[2];
                    const vto_node* left = &N[q->node];
                    const int nl = node_step(q, f, s2, c, e);
                    ++did; ++steps;
                    if (!nl) continue;
                    /* which of the two children the reported leaves are */
                    uint32_t cl = 0, cr = 0, start = c[0]; float fl = 0, fr = 0;
                    for (int k = 0; k < nl; ++k) {
                        if (c[k] == left->first && left->prim_count) { cl = e[k] - c[k]; fl = f[k]; } else { cr = e[k] - c[k]; fr = f[k]; }
                    }
                    if (nl == 2 && e[0] != c[1]) { fprintf(stderr, "sibling leaves not contiguous\n"); exit(3); }
                    if (!(q->tri_cur < q->tri_end)) { q->tri_cur = start; q->tri_end = start + cl + cr; }
                    else { q->gvalid = 1; q->gfl = fl; q->gfr = fr; q->gstart = start; q->gcl = cl; q->gcr = cr; }
                }
                if (did) { ++node_passes; node_lanes += did; }
                stalled += blocked;
                for (int l = 0; l < LANES; ++l) {
                    lane_t* q = &L[l];
                    if (q->has_ray && q->node_done && !(q->tri_cur < q->tri_end) && !q->gvalid) { OUT[q->idx] = q->best; if (q->best.prim == VTO_MISS) OUT[q->idx].t = 0.f; q->has_ray = 0; }
                }
                continue;
            }
            {
                /* NODE pass.  policy 0: a lane with queued leaves waits (its walk continues only when they are drained);
                 * policy 1: it keeps walking with its current (possibly stale) tmax while its queue has room. */
                int did = 0, blocked = 0, holders = 0;
                int stepped[LANES] = {0};
                for (int l = 0; l < LANES; ++l) {
                    lane_t* q = &L[l];
                    if (!q->has_ray) continue;
                    if (q->node_done) { if (q->qn) ++blocked; continue; }
                    if (policy == 0 ? q->qn > 0 : q->qn >= qcap - 1) { ++blocked; continue; }   /* a step can add two leaves */
                    stepped[l] = 1;
                }
                /* policy 0 decides the TRI pass BEFORE the node pass (a lane does one or the other per iteration) */
                for (int l = 0; l < LANES; ++l) holders += L[l].has_ray && L[l].qn;
                int can_step = 0;
                for (int l = 0; l < LANES; ++l) can_step += stepped[l];
                const int fire = holders && (policy == 0 ? (holders >= p1 || can_step == 0)
                                                         : (holders >= p1 || blocked * 4 >= active || can_step == 0));
                if (!fire) wait_lanes += (uint64_t)holders;
                if (fire) {
                    int used = 0;
                    for (int l = 0; l < LANES; ++l) {
                        lane_t* q = &L[l];
                        if (!q->has_ray || !q->qn) continue;
                        qentry* en = &q->q[q->qh % QCAP];
                        if (!en->checked) {
                            /* the reference slab-tested this leaf (and its sibling found by the same step) with the tmax it had
                             * at that step = the tmax now, when every earlier triangle test has been done */
                            const int n2 = en->pair_with_next ? 2 : 1;
                            for (int k = 0; k < n2; ++k) {
                                qentry* x = &q->q[(q->qh + k) % QCAP];
                                const float second_now = x->second < q->tmax ? x->second : q->tmax;
                                x->checked = 1; x->dead = !(x->first <= second_now);
                            }
                        }
                        if (en->dead) { ++retests_failed; q->qh++; q->qn--; continue; }   /* costs the lane its turn, not a triangle test */
                        ++used; ++tests;
                        test_slot(q, en->cur++);
                        if (en->cur >= en->end) { q->qh++; q->qn--; }
                    }
                    if (used) { ++tri_passes; tri_lanes += used; }
                }
                for (int l = 0; l < LANES; ++l) {
                    lane_t* q = &L[l];
                    if (!stepped[l]) continue;
                    float f[2], s[2]; uint32_t c[2], e[2];
                    const int nl = node_step(q, f, s, c, e);
                    ++did; ++steps;
                    for (int k = 0; k < nl; ++k) {
                        qentry* en = &q->q[(q->qh + q->qn++) % QCAP];
                        en->first = f[k]; en->second = s[k]; en->cur = c[k]; en->end = e[k];
                        en->checked = 0; en->dead = 0; en->pair_with_next = (nl == 2 && k == 0);
                    }
                }
                if (did) { ++node_passes; node_lanes += did; }
                stalled += blocked;
            }
            for (int l = 0; l < LANES; ++l) {
                lane_t* q = &L[l];
                const int pending = q->qn;
                if (q->has_ray && q->node_done && !pending) { OUT[q->idx] = q->best; if (q->best.prim == VTO_MISS) OUT[q->idx].t = 0.f; q->has_ray = 0; }
            }
        }
        free(L);
    }
    /* exactness: compare with the oracle's own walk */
    uint64_t bad = 0, osteps = 0, otests = 0;
#pragma omp parallel for reduction(+ : bad, osteps, otests)
    for (int64_t i = 0; i < (int64_t)nrays; ++i) {
        vto_hit h; vto_stats st;
        vto_traverse(N, PI, T, &R[i], 0, &h, &st);
        osteps += st.steps; otests += st.tests;
        if (h.prim != OUT[i].prim || memcmp(&h.t, &OUT[i].t, 12) != 0) ++bad;
    }
    const double cost = node_passes * 78.0 + tri_passes * 61.0 + iters * 30.0;
    printf("policy %d p1=%d refill=%d qcap=%d: rays %llu  iterations %llu  NODE passes %llu (%.1f lanes)  TRI passes %llu (%.1f lanes)  steps/ray %.2f (oracle %.2f)  tests/ray %.2f (oracle %.2f)  "
           "failed re-tests/ray %.2f  blocked lane-iters/ray %.2f  VALU model/ray %.0f  mismatches vs oracle %llu\n",
           policy, p1, refill, qcap, (unsigned long long)nrays, (unsigned long long)iters, (unsigned long long)node_passes, (double)node_lanes / node_passes,
           (unsigned long long)tri_passes, tri_passes ? (double)tri_lanes / tri_passes : 0.0, (double)steps / nrays, (double)osteps / nrays, (double)tests / nrays,
           (double)otests / nrays, (double)retests_failed / nrays, (double)stalled / nrays, cost / nrays, (unsigned long long)bad);
    /* where a wave's 64 lanes are in an average iteration (policy 0 = the shipped kernel) */
    printf("  lanes per iteration: NODE step %.1f, TRI test %.1f, waiting for the TRI pass %.1f, no ray %.1f; TRI pass in %.0f %% of the iterations, "
           "lanes per TRI pass %.1f; VALU lane use (78 x NODE + 61 x TRI + 30 per iteration, lanes active / 64) %.3f\n",
           (double)node_lanes / iters, (double)tri_lanes / iters, (double)wait_lanes / iters, (double)idle_lanes / iters,
           100.0 * tri_passes / iters, tri_passes ? (double)tri_lanes / tri_passes : 0.0,
           (78.0 * node_lanes + 61.0 * tri_lanes + 30.0 * (double)(iters * LANES - idle_lanes)) / (64.0 * (78.0 * node_passes + 61.0 * tri_passes + 30.0 * iters)));
    return bad ? 1 : 0;
}
