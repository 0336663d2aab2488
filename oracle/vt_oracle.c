/*
 * vt_oracle.c -- CPU ORACLE (test infrastructure; see vt_oracle.h).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp  (oracle/Makefile)
 * All arithmetic is IEEE fp32, one rounding per operation, exactly as a
 * baseline-x86-64 build of the reference evaluates it (no FMA: the reference's
 * shipped binaries target baseline x86-64 and bvh v1's fast_multiply_add is a
 * plain a*b+c unless FP_FAST_FMAF is defined).
 *
 * PARITY STATUS: parity unpinned (no reference tests/vectors exist; libs/bvh
 * is absent).  See the header.
 *
 * RECALL-SENSITIVITY SWITCHES (scripts/recall_sensitivity.py, `make -C oracle alts`): each -DVTO_ALT_<X> builds a
 * VARIANT of this oracle in which ONE detail of the walk recalled from madmann91/bvh v1 (SURVEY.md section 3.2's
 * confidence table; libs/bvh is absent, reference call sites source/objects/AccelStruct.h:23-31) is read the OTHER
 * plausible way.  The variants exist to count how many rays would change if upstream differs from the shipped reading;
 * nothing but that script and tests/test_recall_sensitivity.py loads them, and the default build defines none of them
 * (vto_alt_mask() == 0 is asserted).
 *   VTO_ALT_PLAIN_INVERSE  inv_dir = 1/x (early v1: +-inf for zero components) instead of safe_inverse
 *   VTO_ALT_SWAP_GE        near/far swap on dL.first >= dR.first instead of >
 *   VTO_ALT_FMA            fast_multiply_add fused (FP_FAST_FMAF builds) instead of a*b then +c
 *   VTO_ALT_RETEST_RIGHT   right child's slab test AFTER the left leaf was intersected (sees the shrunk tmax)
 *   VTO_ALT_LEAF_DESC      leaf slots visited in descending instead of ascending order
 *   VTO_ALT_ACCEPT_LT      node accepted on first < second instead of <=
 *   VTO_ALT_PUSH_NODE_CULL stack holds the far NODE with its entry distance and a pop discards it when that
 *                          distance exceeds the current tmax (storing the node instead of its first child without
 *                          that test is the same walk by construction)
 *   VTO_ALT_FMINMAX        fmaxf/fminf (NaN-ignoring) instead of robust_max/robust_min (a > b ? a : b)
 */
#include "vt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- bvh v1 vector helpers [UPSTREAM-RECALL bvh/vector.hpp] ---------------
 * dot:   sum = a0*b0; sum += a1*b1; sum += a2*b2      (left to right)
 * cross: c[i] = a[j]*b[k] - a[k]*b[j], j=(i+1)%3, k=(i+2)%3            */
static inline float dot3(const float a[3], const float b[3])
{
    float s = a[0] * b[0];
    s = s + a[1] * b[1];
    s = s + a[2] * b[2];
    return s;
}

static inline void cross3(const float a[3], const float b[3], float c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* ---- Primitives.h:75-102 ---------------------------------------------------
 * ctor: p0(p0), e1(p0 - p1), e2(p2 - p0); ComputeNormalAndLoD: n = cross(e1,e2)
 * (LeftHandedNormal = true).  lod / nNorm are not read by intersect(). */
void vto_tri_setup(const float p0[3], const float p1[3], const float p2[3],
                   uint32_t flags, vto_tri* out)
{
    for (int a = 0; a < 3; ++a) {
        out->p0[a] = p0[a];
        out->e1[a] = p0[a] - p1[a];   /* Primitives.h:82 */
        out->e2[a] = p2[a] - p0[a];   /* Primitives.h:82 */
    }
    cross3(out->e1, out->e2, out->n); /* Primitives.h:93 */
    out->flags = flags;
}

/* ---- Primitives.h:168-215 --------------------------------------------------*/
/* ---- Primitives.h:196-208 + Utils.h:65-72; the texel lookup itself is defined in vt_oracle.h ----*/
static const vto_alpha_ctx* g_alpha = 0;

void vto_set_alpha(const vto_alpha_ctx* ctx) { g_alpha = ctx; }

static uint32_t wrap_index(float f, uint32_t n)
{
    int64_t i = (int64_t)f;                     /* f is integral and |f| < 1e9 */
    int64_t m = i % (int64_t)n;
    return (uint32_t)(m < 0 ? m + (int64_t)n : m);
}

float vto_alpha_sample(const vto_alpha_material* m, const uint8_t* texels, float s, float t)
{
    if (m->width == 0 || m->height == 0) return 1.0f;
    const float W = (float)m->width, H = (float)m->height;
    const uint8_t* img = texels + m->offset;
    float x = s * W, y = t * H;
    if (!(fabsf(x) < 1.0e9f)) x = 0.0f;         /* NaN, inf or absurd coordinates: texel (0, 0) */
    if (!(fabsf(y) < 1.0e9f)) y = 0.0f;
    if (m->filter == 0) {
        const uint32_t xi = wrap_index(floorf(x), m->width), yi = wrap_index(floorf(y), m->height);
        return (float)img[(size_t)yi * m->width + xi] / 255.0f;
    }
    const float fx = x - 0.5f, fy = y - 0.5f;
    const float x0 = floorf(fx), y0 = floorf(fy);
    const float ax = fx - x0, ay = fy - y0;
    const uint32_t i0 = wrap_index(x0, m->width), i1 = wrap_index(x0 + 1.0f, m->width);
    const uint32_t j0 = wrap_index(y0, m->height), j1 = wrap_index(y0 + 1.0f, m->height);
    const float a00 = (float)img[(size_t)j0 * m->width + i0], a10 = (float)img[(size_t)j0 * m->width + i1];
    const float a01 = (float)img[(size_t)j1 * m->width + i0], a11 = (float)img[(size_t)j1 * m->width + i1];
    const float top = a00 * (1.0f - ax) + a10 * ax;
    const float bot = a01 * (1.0f - ax) + a11 * ax;
    return (top * (1.0f - ay) + bot * ay) / 255.0f;
}

int vto_alpha_pass(const vto_alpha_ctx* ctx, uint32_t prim, float u, float v)
{
    const float* uv = ctx->tri_uv + (size_t)prim * 6;
    const uint32_t mi = ctx->tri_material[prim];
    if (mi >= ctx->nmats) return 1;
    const vto_alpha_material* m = &ctx->mats[mi];
    const float w = 1.0f - u - v;                                             /* :198 */
    const float tx = (w * uv[0] + u * uv[2]) + v * uv[4];
    const float ty = (w * uv[1] + u * uv[3]) + v * uv[5];
    /* TransformTexcoord: dot(vec4(texcoord, 1, 1), transform[r]) * scale; glm dot(vec4) = (x + y) + (z + w) */
    const float sx = ((tx * m->tex_mat[0][0] + ty * m->tex_mat[0][1]) + (m->tex_mat[0][2] + m->tex_mat[0][3])) * m->tex_scale;
    const float sy = ((tx * m->tex_mat[1][0] + ty * m->tex_mat[1][1]) + (m->tex_mat[1][2] + m->tex_mat[1][3])) * m->tex_scale;
    const float alpha = vto_alpha_sample(m, ctx->texels, sx, sy);             /* :202 (sampler defined here) */
    return alpha < m->alpha_ref ? 0 : 1;                                      /* :205 */
}

int vto_tri_intersect(const vto_tri* tri, const float org[3], const float dir[3],
                      float tmin, float tmax, float* t_out, float* u_out, float* v_out)
{
    float nDotDir = dot3(tri->n, dir);                               /* :173 */
    if ((tri->flags & VTO_TRI_CULL_BACKFACE) && nDotDir > 0.0f)      /* :174 */
        return 0;

    float c[3], r[3];
    for (int a = 0; a < 3; ++a) c[a] = tri->p0[a] - org[a];          /* :176 */
    cross3(dir, c, r);                                               /* :177 */
    float inv_det = 1.0f / nDotDir;                                  /* :178 */

    float u = dot3(r, tri->e2) * inv_det;                            /* :180 */
    float v = dot3(r, tri->e1) * inv_det;                            /* :181 */
    float w = 1.0f - u - v;                                          /* :182 */

    /* tolerance = 0 (NonZeroTolerance=false); NaN makes every test false :186-187 */
    if (u >= 0.0f && v >= 0.0f && w >= 0.0f) {
        float t = dot3(tri->n, c) * inv_det;                         /* :188 */
        if (t >= tmin && t <= tmax) {                                /* :189 */
            /* :196-208: a material with the alphatest flag discards the hit when the base texture's alpha at
             * the hit point is below the reference */
            if ((tri->flags & VTO_TRI_ALPHATEST) && g_alpha &&
                !vto_alpha_pass(g_alpha, (uint32_t)(tri - g_alpha->tris_base), u, v))
                return 0;
            *t_out = t; *u_out = u; *v_out = v;
            return 1;
        }
    }
    return 0;
}

/* ---- brute force (independent ground truth) -------------------------------*/
static void brute_one(const vto_tri* tris, uint32_t ntris, const vto_ray* ray,
                      int any_hit, vto_hit* hit)
{
    float tmax = ray->tmax;
    hit->prim = VTO_MISS; hit->t = 0.f; hit->u = 0.f; hit->v = 0.f;
    for (uint32_t i = 0; i < ntris; ++i) {
        float t, u, v;
        if (vto_tri_intersect(&tris[i], ray->org, ray->dir, ray->tmin, tmax, &t, &u, &v)) {
            hit->prim = i; hit->t = t; hit->u = u; hit->v = v;
            if (any_hit) return;
            tmax = t;
        }
    }
}

void vto_trace_brute(const vto_tri* tris, uint32_t ntris, const vto_ray* rays,
                     uint64_t nrays, int any_hit, vto_hit* hits, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads)
    for (int64_t i = 0; i < (int64_t)nrays; ++i)
        brute_one(tris, ntris, &rays[i], any_hit, &hits[i]);
}

uint32_t vto_min_t_set(const vto_tri* tris, uint32_t ntris, const vto_ray* ray,
                       float* tmin_hit, uint32_t* ids, uint32_t max_ids)
{
    float best = ray->tmax;
    int found = 0;
    for (uint32_t i = 0; i < ntris; ++i) {
        float t, u, v;
        if (vto_tri_intersect(&tris[i], ray->org, ray->dir, ray->tmin, best, &t, &u, &v)) {
            best = t; found = 1;
        }
    }
    if (!found) { *tmin_hit = 0.f; return 0; }
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < ntris; ++i) {
        float t, u, v;
        if (vto_tri_intersect(&tris[i], ray->org, ray->dir, ray->tmin, best, &t, &u, &v) && t == best) {
            if (cnt < max_ids) ids[cnt] = i;
            ++cnt;
        }
    }
    *tmin_hit = best;
    return cnt;
}

/* ---- bvh v1 utilities [UPSTREAM-RECALL bvh/utilities.hpp] -------------------
 * safe_inverse(x) = |x| <= FLT_EPSILON ? copysign(1/FLT_EPSILON, x) : 1/x
 * robust_max(a,b) = a > b ? a : b ;  robust_min(a,b) = a < b ? a : b
 * fast_multiply_add(a,b,c) = a*b + c  (unfused; FP_FAST_FMAF not defined)     */
static inline float safe_inverse(float x)
{
#ifdef VTO_ALT_PLAIN_INVERSE
    return 1.0f / x;
#else
    return fabsf(x) <= FLT_EPSILON ? copysignf(1.0f / FLT_EPSILON, x) : 1.0f / x;
#endif
}
#ifdef VTO_ALT_FMINMAX
static inline float robust_max(float a, float b) { return fmaxf(a, b); }
static inline float robust_min(float a, float b) { return fminf(a, b); }
#else
static inline float robust_max(float a, float b) { return a > b ? a : b; }
static inline float robust_min(float a, float b) { return a < b ? a : b; }
#endif
#ifdef VTO_ALT_FMA
static inline float fast_multiply_add(float a, float b, float c) { return fmaf(a, b, c); }
#else
static inline float fast_multiply_add(float a, float b, float c) { return a * b + c; }
#endif
#ifdef VTO_ALT_ACCEPT_LT
#define VTO_NODE_ACCEPT(first, second) ((first) < (second))
#else
#define VTO_NODE_ACCEPT(first, second) ((first) <= (second))
#endif

uint32_t vto_alt_mask(void)
{
    uint32_t m = 0;
#ifdef VTO_ALT_PLAIN_INVERSE
    m |= 1u;
#endif
#ifdef VTO_ALT_SWAP_GE
    m |= 2u;
#endif
#ifdef VTO_ALT_FMA
    m |= 4u;
#endif
#ifdef VTO_ALT_RETEST_RIGHT
    m |= 8u;
#endif
#ifdef VTO_ALT_LEAF_DESC
    m |= 16u;
#endif
#ifdef VTO_ALT_ACCEPT_LT
    m |= 32u;
#endif
#ifdef VTO_ALT_PUSH_NODE_CULL
    m |= 64u;
#endif
#ifdef VTO_ALT_FMINMAX
    m |= 128u;
#endif
    return m;
}

/* FastNodeIntersector [UPSTREAM-RECALL bvh/node_intersectors.hpp] */
typedef struct {
    int   oct[3];
    float inv_dir[3];
    float scaled_org[3];
} node_isect;

static inline void node_isect_init(node_isect* ni, const vto_ray* ray)
{
    for (int a = 0; a < 3; ++a) {
        ni->oct[a]        = signbit(ray->dir[a]) ? 1 : 0;
        ni->inv_dir[a]    = safe_inverse(ray->dir[a]);
        ni->scaled_org[a] = -ray->org[a] * ni->inv_dir[a];
    }
}

static inline void node_slab(const node_isect* ni, const vto_node* node,
                             float tmin, float tmax, float* first, float* second)
{
    float entry[3], exit_[3];
    for (int a = 0; a < 3; ++a) {
        entry[a] = fast_multiply_add(node->bounds[2 * a + ni->oct[a]],     ni->inv_dir[a], ni->scaled_org[a]);
        exit_[a] = fast_multiply_add(node->bounds[2 * a + 1 - ni->oct[a]], ni->inv_dir[a], ni->scaled_org[a]);
    }
    *first  = robust_max(entry[0], robust_max(entry[1], robust_max(entry[2], tmin)));
    *second = robust_min(exit_[0], robust_min(exit_[1], robust_min(exit_[2], tmax)));
}

#define VTO_STACK_CAP 256 /* upstream: 64, overflow unchecked; we abort instead */

/* intersect_leaf of SingleRayTraverser: ascending leaf-slot order; Closest:
 * best = hit, ray.tmax = t; Any: return at once.  Returns 1 if the traversal
 * must stop (any-hit found). */
static inline int leaf_isect(const vto_node* leaf, const uint32_t* prim_indices,
                             const vto_tri* tris, const vto_ray* ray, float* tmax,
                             int any_hit, vto_hit* best, int* found, uint64_t* tests)
{
    uint32_t begin = leaf->first, end = begin + leaf->prim_count;
    for (uint32_t k = begin; k < end; ++k) {
#ifdef VTO_ALT_LEAF_DESC
        const uint32_t i = end - 1 - (k - begin);
#else
        const uint32_t i = k;
#endif
        uint32_t idx = prim_indices[i];            /* ClosestPrimitiveIntersector, PreShuffled=false */
        float t, u, v;
        ++*tests;
        if (vto_tri_intersect(&tris[idx], ray->org, ray->dir, ray->tmin, *tmax, &t, &u, &v)) {
            best->prim = idx; best->t = t; best->u = u; best->v = v;
            *found = 1;
            if (any_hit) return 1;
            *tmax = t;
        }
    }
    return 0;
}

/* diagnostics for stack sizing (tests/scripts only): per-call maximum stack depth and push count */
static _Thread_local uint32_t g_last_max_sp, g_last_pushes;
void vto_last_stack_use(uint32_t* max_sp, uint32_t* pushes) { *max_sp = g_last_max_sp; *pushes = g_last_pushes; }

int vto_traverse(const vto_node* nodes, const uint32_t* prim_indices,
                 const vto_tri* tris, const vto_ray* ray, int any_hit,
                 vto_hit* best, vto_stats* stats)
{
    uint64_t steps = 0, tests = 0;
    int found = 0;
    g_last_max_sp = 0; g_last_pushes = 0;
    float tmax = ray->tmax;            /* traverse() takes the ray by value */
    best->prim = VTO_MISS; best->t = 0.f; best->u = 0.f; best->v = 0.f;

    if (nodes[0].prim_count != 0) {    /* root is a leaf: no slab test at all */
        leaf_isect(&nodes[0], prim_indices, tris, ray, &tmax, any_hit, best, &found, &tests);
        goto done;
    }

    {
        node_isect ni;
        node_isect_init(&ni, ray);
        uint32_t stack[VTO_STACK_CAP];
#ifdef VTO_ALT_PUSH_NODE_CULL
        float stack_first[VTO_STACK_CAP];
#endif
        int sp = 0;
        const vto_node* left = &nodes[nodes[0].first];
        for (;;) {
            const vto_node* right = left + 1;
            ++steps;
            float fl, sl, fr, sr;
            /* both children are slab-tested BEFORE either leaf is intersected */
            node_slab(&ni, left,  ray->tmin, tmax, &fl, &sl);
#ifndef VTO_ALT_RETEST_RIGHT
            node_slab(&ni, right, ray->tmin, tmax, &fr, &sr);
#endif

            if (VTO_NODE_ACCEPT(fl, sl)) {
                if (left->prim_count != 0) {
                    if (leaf_isect(left, prim_indices, tris, ray, &tmax, any_hit, best, &found, &tests))
                        goto done;
                    left = NULL;
                }
            } else left = NULL;

#ifdef VTO_ALT_RETEST_RIGHT
            node_slab(&ni, right, ray->tmin, tmax, &fr, &sr);
#endif
            if (VTO_NODE_ACCEPT(fr, sr)) {
                if (right->prim_count != 0) {
                    if (leaf_isect(right, prim_indices, tris, ray, &tmax, any_hit, best, &found, &tests))
                        goto done;
                    right = NULL;
                }
            } else right = NULL;

            if (left) {
                if (right) {
#ifdef VTO_ALT_SWAP_GE
                    if (fl >= fr) {
#else
                    if (fl > fr) {
#endif
                        const vto_node* tmp = left; left = right; right = tmp;
#ifdef VTO_ALT_PUSH_NODE_CULL
                        float tf = fl; fl = fr; fr = tf;
#endif
                    }
                    if (sp >= VTO_STACK_CAP) { fprintf(stderr, "vt_oracle: stack overflow\n"); abort(); }
#ifdef VTO_ALT_PUSH_NODE_CULL
                    stack_first[sp] = fr;
#endif
                    stack[sp++] = right->first;     /* far inner node's first-child index */
                    ++g_last_pushes;
                    if ((uint32_t)sp > g_last_max_sp) g_last_max_sp = (uint32_t)sp;
                }
                left = &nodes[left->first];
            } else if (right) {
                left = &nodes[right->first];
            } else {
#ifdef VTO_ALT_PUSH_NODE_CULL
                while (sp > 0 && stack_first[sp - 1] > tmax) --sp;   /* the stored node no longer reaches below tmax */
#endif
                if (sp == 0) break;
                left = &nodes[stack[--sp]];
            }
        }
    }
done:
    if (stats) { stats->steps = steps; stats->tests = tests; }
    return found;
}

int vto_traverse_batch(const vto_node* nodes, const uint32_t* prim_indices,
                       const vto_tri* tris, const vto_ray* rays, uint64_t nrays,
                       int any_hit, vto_hit* hits, uint32_t* per_ray_stats,
                       vto_stats* total, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    uint64_t tsteps = 0, ttests = 0;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nthreads) reduction(+ : tsteps, ttests)
    for (int64_t i = 0; i < (int64_t)nrays; ++i) {
        vto_stats st;
        vto_traverse(nodes, prim_indices, tris, &rays[i], any_hit, &hits[i], &st);
        if (per_ray_stats) {
            per_ray_stats[2 * i]     = (uint32_t)st.steps;
            per_ray_stats[2 * i + 1] = (uint32_t)st.tests;
        }
        tsteps += st.steps; ttests += st.tests;
    }
    if (total) { total->steps = tsteps; total->tests = ttests; }
    return nthreads;
}

/* ---- NUMA-aware batch driver (bench.py's cpu_baseline leg only) -------------------------------------------
 * The tree of the 1 M-triangle scene (nodes 64 MB + triangles 52 MB) is allocated by one thread of the caller, so it
 * sits on ONE socket's memory; on a two-socket host every thread of the other socket then chases pointers across
 * the inter-socket link.  A context holds one replica per NUMA node, each first-touched by a thread running there;
 * every worker walks the replica of the node it runs on (threads should be pinned: OMP_PROC_BIND).  Same arithmetic,
 * same results. */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

#define VTO_MAX_NUMA 16
struct vto_batch_ctx {
    vto_node* nodes[VTO_MAX_NUMA];
    uint32_t* prim_indices[VTO_MAX_NUMA];
    vto_tri*  tris[VTO_MAX_NUMA];
    int       replicas;
};

static int current_numa_node(void)
{
    unsigned cpu = 0, node = 0;
#ifdef SYS_getcpu
    if (syscall(SYS_getcpu, &cpu, &node, NULL) != 0) node = 0;
#endif
    return (int)(node < VTO_MAX_NUMA ? node : 0);
}

vto_batch_ctx* vto_batch_ctx_create(const vto_node* nodes, uint64_t nnodes, const uint32_t* prim_indices, uint64_t nprims,
                                    const vto_tri* tris, uint64_t ntris, int nthreads)
{
    vto_batch_ctx* c = (vto_batch_ctx*)calloc(1, sizeof(*c));
    if (!c) return NULL;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    int claimed[VTO_MAX_NUMA] = {0};
#pragma omp parallel num_threads(nthreads)
    {
        const int node = current_numa_node();
        int mine = 0;
#pragma omp critical(vto_ctx_claim)
        { if (!claimed[node]) { claimed[node] = 1; mine = 1; } }
        if (mine) {                       /* first thread seen on this node: allocate + copy here (first touch) */
            vto_node* n = (vto_node*)malloc((size_t)(nnodes ? nnodes : 1) * sizeof(vto_node));
            uint32_t* p = (uint32_t*)malloc((size_t)(nprims ? nprims : 1) * sizeof(uint32_t));
            vto_tri*  t = (vto_tri*)malloc((size_t)(ntris ? ntris : 1) * sizeof(vto_tri));
            if (n && p && t) {
                memcpy(n, nodes, (size_t)nnodes * sizeof(vto_node));
                memcpy(p, prim_indices, (size_t)nprims * sizeof(uint32_t));
                memcpy(t, tris, (size_t)ntris * sizeof(vto_tri));
                c->nodes[node] = n; c->prim_indices[node] = p; c->tris[node] = t;
            } else { free(n); free(p); free(t); }
        }
    }
    for (int k = 0; k < VTO_MAX_NUMA; ++k) c->replicas += c->nodes[k] != NULL;
    if (c->replicas == 0) { free(c); return NULL; }
    return c;
}

int vto_batch_ctx_replicas(const vto_batch_ctx* c) { return c ? c->replicas : 0; }

void vto_batch_ctx_destroy(vto_batch_ctx* c)
{
    if (!c) return;
    for (int k = 0; k < VTO_MAX_NUMA; ++k) { free(c->nodes[k]); free(c->prim_indices[k]); free(c->tris[k]); }
    free(c);
}

int vto_traverse_batch_ctx(const vto_batch_ctx* c, const vto_ray* rays, uint64_t nrays, int any_hit, vto_hit* hits,
                           uint32_t* per_ray_stats, vto_stats* total, int nthreads)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    int first = 0;
    while (first < VTO_MAX_NUMA && !c->nodes[first]) ++first;
    uint64_t tsteps = 0, ttests = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : tsteps, ttests)
    {
        int node = current_numa_node();
        if (!c->nodes[node]) node = first;
        const vto_node* nodes = c->nodes[node];
        const uint32_t* pidx = c->prim_indices[node];
        const vto_tri* tris = c->tris[node];
#pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < (int64_t)nrays; ++i) {
            vto_stats st;
            vto_traverse(nodes, pidx, tris, &rays[i], any_hit, &hits[i], &st);
            if (per_ray_stats) {
                per_ray_stats[2 * i]     = (uint32_t)st.steps;
                per_ray_stats[2 * i + 1] = (uint32_t)st.tests;
            }
            tsteps += st.steps; ttests += st.tests;
        }
    }
    if (total) { total->steps = tsteps; total->tests = ttests; }
    return nthreads;
}

/* ---- TraceResult.cpp:45-86, 255-262 ---------------------------------------*/
void vto_hit_attrs(const vto_tri* tri, const float dir[3], float u, float v, vto_attrs* out)
{
    /* AccelStruct.cpp:826 glm::normalize(dir) = dir * (1/sqrt(dot(dir,dir))) */
    float d2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
    float inv = 1.0f / sqrtf(d2);
    for (int a = 0; a < 3; ++a) out->wo[a] = -(dir[a] * inv);          /* TraceResult.cpp:56 */

    float w = 1.0f - u - v;                                            /* :70 */
    out->uvw[0] = u; out->uvw[1] = v; out->uvw[2] = w;

    /* Primitives.h:98-100 nNorm = n / length(n) */
    float len = sqrtf(dot3(tri->n, tri->n));
    for (int a = 0; a < 3; ++a) out->ngeo[a] = tri->n[a] / len;        /* :71 */

    /* v0 = p0, v1 = p0 - e1, v2 = p0 + e2 (Primitives.h:104-105, TraceResult.cpp:65-68)
     * pos = uvw.z*v0 + uvw.x*v1 + uvw.y*v2                              (:258) */
    for (int a = 0; a < 3; ++a) {
        float v0 = tri->p0[a], v1 = tri->p0[a] - tri->e1[a], v2 = tri->p0[a] + tri->e2[a];
        out->pos[a] = (w * v0 + u * v1) + v * v2;
    }
    out->front = dot3(out->wo, out->ngeo) >= 0.0f ? 1u : 0u;           /* :85 */
}

/* ---- TraceResult.cpp:70,73-74 -------------------------------------------------------*/
void vto_hit_shade(float u, float v, const float uvs[6], const float alphas[3], float tex_uv[2], float* blend)
{
    const float w = 1.0f - u - v;                                            /* :70 uvw.z */
    *blend = (w * alphas[0] + u * alphas[1]) + v * alphas[2];                /* :73 */
    tex_uv[0] = (w * uvs[0] + u * uvs[2]) + v * uvs[4];                      /* :74 (glm vec2: per component) */
    tex_uv[1] = (w * uvs[1] + u * uvs[3]) + v * uvs[5];
}

/* ---- VisTrace.cpp:1495-1517 ------------------------------------------------*/
void vto_calc_ray_origin(const float pos[3], const float normal[3], float out[3])
{
    const float origin = 1.f / 32.f;
    const float fScale = 1.f / 65536.f;
    const float iScale = 256.f;
    for (int a = 0; a < 3; ++a) {
        int32_t iOff = (int32_t)(normal[a] * iScale);          /* ivec3(normal * iScale) */
        int32_t bits;
        memcpy(&bits, &pos[a], 4);
        bits += (pos[a] < 0.f) ? -iOff : iOff;
        float iPos;
        memcpy(&iPos, &bits, 4);
        float fOff = normal[a] * fScale;
        out[a] = fabsf(pos[a]) < origin ? pos[a] + fOff : iPos;
    }
}

/* ---- BSDF.cpp:69-77 ---------------------------------------------------------*/
void vto_hemisphere_cos(float r1, float r2, float out[3])
{
    const float z = sqrtf(r1);
    const float sinTheta = sqrtf(1.f - r1);
    const float phi = 2.f * 3.14159265358979323846264338327950288f * r2;
    out[0] = sinTheta * cosf(phi);
    out[1] = sinTheta * sinf(phi);
    out[2] = z;
}

/* ---- AccelStruct.cpp:34-102: skinning -----------------------------------------
 * glm (un-vendored, version unpinned) supplies mat4*mat4, mat4*vec4 and vec4*scalar; the
 * scalar forms restated here are glm's generic (non-SIMD) ones:
 *   mat4*mat4: Result[c] = A[0]*B[c][0] + A[1]*B[c][1] + A[2]*B[c][2] + A[3]*B[c][3], left to right
 *   mat4*vec4: (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w)                                    */
void vto_skin_matrices(const float* bones, const float* binds, uint32_t nmat, float* out)
{
    for (uint32_t i = 0; i < nmat; ++i) {
        const float* A = bones + (size_t)i * 16;
        const float* B = binds + (size_t)i * 16;
        float* R = out + (size_t)i * 16;
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 4; ++r) {
                float acc = A[0 * 4 + r] * B[c * 4 + 0];
                acc = acc + A[1 * 4 + r] * B[c * 4 + 1];
                acc = acc + A[2 * 4 + r] * B[c * 4 + 2];
                acc = acc + A[3 * 4 + r] * B[c * 4 + 3];
                R[c * 4 + r] = acc;
            }
    }
}

/* TransformToBone :35-47 with angleOnly = false: vertex = (vec, 1), final = sum_i (M_i * vertex) * w_i */
static void vto_transform_to_bone(const float v[3], const vto_skin_vertex* sv, const float* mats, float out[3])
{
    float fin[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = 0; i < sv->num_bones; ++i) {
        const float* M = mats + (size_t)(int)sv->bone[i] * 16;
        for (int r = 0; r < 4; ++r) {
            const float a0 = M[0 * 4 + r] * v[0] + M[1 * 4 + r] * v[1];
            const float a1 = M[2 * 4 + r] * v[2] + M[3 * 4 + r] * 1.f;
            fin[r] = fin[r] + (a0 + a1) * sv->weight[i];
        }
    }
    out[0] = fin[0]; out[1] = fin[1]; out[2] = fin[2];
}

void vto_skin_verts(const float* bind_verts, const vto_skin_vertex* skin, const uint32_t* matrix_base,
                    uint32_t n, const float* mats, float* out_verts)
{
    for (uint32_t t = 0; t < n; ++t) {
        const float* b = bind_verts + (size_t)t * 9;
        float pos[3][3];
        for (int k = 0; k < 3; ++k) {                 /* Triangle ctor :82, then SkinTriangle :68-72 */
            const float p0 = b[k], e1 = b[k] - b[3 + k], e2 = b[6 + k] - b[k];
            pos[0][k] = p0; pos[1][k] = p0 - e1; pos[2][k] = p0 + e2;
        }
        const float* mats_t = mats + (size_t)matrix_base[t] * 16;
        for (int vi = 0; vi < 3; ++vi)
            vto_transform_to_bone(pos[vi], &skin[(size_t)t * 3 + vi], mats_t, out_verts + (size_t)t * 9 + vi * 3);
    }
}

/* ---- TraceResult.cpp:58-62, 89-103, 132-137, 175-186 -------------------------------------------*/
static inline void normalize3(float v[3])          /* glm::normalize: v * inversesqrt(dot(v, v)) */
{
    const float inv = 1.0f / sqrtf(dot3(v, v));
    v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv;
}

void vto_hit_tbn(const vto_tri* tri, const float dir[3], float distance, float u, float v,
                 const float normals[9], const float tangents[9], const float uvs[6],
                 float cone_width, float cone_angle, vto_tbn* out)
{
    vto_attrs at;
    vto_hit_attrs(tri, dir, u, v, &at);                       /* wo (:56), geometricNormal (:71), uvw (:70) */
    const float w = at.uvw[2];
    float vB[3][3];
    for (int i = 0; i < 3; ++i) cross3(tangents + 3 * i, normals + 3 * i, vB[i]);      /* :60 */
    float n[3], t[3], b[3];
    for (int a = 0; a < 3; ++a) {                             /* :134-136 uvw[2]*v[0] + uvw[0]*v[1] + uvw[1]*v[2] */
        n[a] = (w * normals[a] + u * normals[3 + a]) + v * normals[6 + a];
        t[a] = (w * tangents[a] + u * tangents[3 + a]) + v * tangents[6 + a];
        b[a] = (w * vB[0][a] + u * vB[1][a]) + v * vB[2][a];
    }
    normalize3(n); normalize3(t); normalize3(b);
    /* material.normalMap == nullptr: :139-173 skipped */
    const float kCosThetaThreshold = 0.1f;                    /* :175 */
    const float cosTheta = fabsf(dot3(at.wo, n));
    if (cosTheta <= kCosThetaThreshold) {
        float s = cosTheta * (1.f / kCosThetaThreshold);      /* :178 saturate */
        s = s < 0.f ? 0.f : s; s = s > 1.f ? 1.f : s;
        for (int a = 0; a < 3; ++a) n[a] = at.ngeo[a] * (1.f - s) + n[a] * s;          /* :179 lerp(geometricNormal, normal, t) */
        normalize3(n);
        const float tn = dot3(t, n);                          /* :181 */
        for (int a = 0; a < 3; ++a) t[a] = t[a] - n[a] * tn;
        normalize3(t);
        cross3(t, n, b);                                      /* :182 */
    }
    for (int a = 0; a < 3; ++a) { out->normal[a] = n[a]; out->tangent[a] = t[a]; out->binormal[a] = b[a]; }
    out->lod_info[0] = out->lod_info[1] = 0.f;
    out->lod_set = 0;
    if (!(cone_width < 0.f || cone_angle <= 0.f)) {           /* :54 mipOverride, :91 */
        const float cw = cone_angle * distance + cone_width;  /* :95 */
        const float normalTerm = dot3(at.wo, at.ngeo);        /* :97 */
        /* Primitives.h:97-103 */
        const float uv10x = uvs[2] - uvs[0], uv10y = uvs[3] - uvs[1];
        const float uv20x = uvs[4] - uvs[0], uv20y = uvs[5] - uvs[1];
        const float triUVArea = fabsf(uv10x * uv20y - uv20x * uv10y);
        const float len = sqrtf(dot3(tri->n, tri->n));
        out->lod_info[0] = 0.5f * log2f(triUVArea / len);
        out->lod_info[1] = (cw * cw) / (normalTerm * normalTerm);                       /* :99-102 */
        out->lod_set = 1;
    }
}

/* TransformToBone :35-47 with angleOnly = true: vertex = (vec, 0) */
static void vto_rotate_to_bone(const float v[3], const vto_skin_vertex* sv, const float* mats, float out[3])
{
    float fin[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t i = 0; i < sv->num_bones; ++i) {
        const float* M = mats + (size_t)(int)sv->bone[i] * 16;
        for (int r = 0; r < 4; ++r) {
            const float a0 = M[0 * 4 + r] * v[0] + M[1 * 4 + r] * v[1];
            const float a1 = M[2 * 4 + r] * v[2] + M[3 * 4 + r] * 0.f;
            fin[r] = fin[r] + (a0 + a1) * sv->weight[i];
        }
    }
    out[0] = fin[0]; out[1] = fin[1]; out[2] = fin[2];
}

void vto_skin_frames(const float* bind_frames, const vto_skin_vertex* skin, const uint32_t* matrix_base,
                     uint32_t n, const float* mats, float* out_frames)
{
    for (uint32_t t = 0; t < n; ++t) {
        const float* mats_t = mats + (size_t)matrix_base[t] * 16;
        for (int vi = 0; vi < 3; ++vi) {
            const vto_skin_vertex* sv = &skin[(size_t)t * 3 + vi];
            vto_rotate_to_bone(bind_frames + (size_t)t * 18 + vi * 3, sv, mats_t, out_frames + (size_t)t * 18 + vi * 3);           /* :82 normals */
            vto_rotate_to_bone(bind_frames + (size_t)t * 18 + 9 + vi * 3, sv, mats_t, out_frames + (size_t)t * 18 + 9 + vi * 3);   /* :88 tangents */
        }
    }
}
