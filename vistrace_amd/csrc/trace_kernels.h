// trace_kernels.h -- launch interface between engine.hip and trace_kernels.hip
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "vistrace_hip.h"

namespace vt {

constexpr uint32_t kBlockThreads = 256; // 4 waves of 64

constexpr uint32_t kExitWord = 120;   // of the 128 cursor words of a launch slot: waves that have left the persistent grid

// One batch of a launch.  A plain launch has one (TraceArgs::seg0, in the kernel arguments); a merged launch
// (vt_trace_closest_multi_dev) numbers the ray blocks of all its batches through, and a wave that acquires block b finds the
// batch with first_block <= b < end_block in TraceArgs::segs (device memory, nseg entries in block order) -- many small
// batches then share ONE grid start and ONE drain.
struct TraceSeg {
    const vt_ray* rays;         // ray 0 of the batch
    int64_t       out_off;      // result record of its ray 0, counted from TraceArgs::hits (16-B records) / occluded (bytes)
    uint64_t      n;            // rays
    // Image-order batches (engine option "ray_image_width", vt_batch_desc::ray_image_width): the rays are rows of tile_w rays.  A
    // wave then takes its 64 rays as a 4-wide, 16-high pixel tile instead of 64 consecutive rays of one row -- neighbours in both
    // directions walk the same nodes for longer (camera rays: -8..-18 % kernel time).  The ray and hit arrays keep their order: only
    // the lane <-> ray mapping changes.
    uint64_t      tiled_rays;   // the first tiled_rays rays (whole 16-row bands, < 2^32) are taken tile-wise, the rest in order
    uint32_t      first_block;  // its ray blocks [first_block, end_block) in the launch's numbering (persistent: block_rays rays
    uint32_t      end_block;    //   each, else 256); an empty batch has none
    uint32_t      tile_w;       // 0 = off; a multiple of 4
    uint32_t      pad;
};
static_assert(sizeof(TraceSeg) == 48, "TraceSeg layout");

struct TraceArgs {
    const void*         records;     // 64-B records: pairs [0, npairs), triangles from tri_base on
    TraceSeg            seg0;        // the batch of a plain launch (a merged launch starts from it: its batch 0)
    const TraceSeg*     segs;        // merged launch: its batches; NULL for a plain launch
    uint32_t            nseg;        // 0 = plain launch
    vt_hit*             hits;        // closest-hit output (or nullptr for any-hit)
    uint8_t*            occluded;    // any-hit output
    vt_ray_stats*       ray_stats;   // STATS kernels only
    uint32_t*           overflow;    // per-lane stack overflow area: entry k of thread g at [k*gstride + g]
    uint32_t*           block_cursor;// persistent mode: next block of rays to hand out, counted from cursor_base.  Zero when a launch
                                     // starts; the last wave to leave puts it back to zero (word kExitWord counts the leavers)
    uint32_t            cursor_base; // the first ray block the cursor hands out (the blocks before it are assigned statically)
    uint32_t            npairs;
    uint32_t            tri_base;    // record index of triangle 0
    uint32_t            root_leaf_count;
    uint32_t            lds_entries; // stack entries per lane kept in LDS
    uint32_t            block_rays;  // rays per block handed to a wave (multiple of 64)
    uint32_t            refill_threshold; // re-fill a wave once this many lanes are idle
    uint32_t            tri_threshold;    // run the TRI branch once this many lanes wait for it
    uint32_t            coherent_detect;  // DMA kernel: per-wave octant probe -> direct fetch, whole-wave re-fill
    float               coherent_radius2; // ... only for rays whose origins lie within this squared distance of the first
    // alpha test (ALPHA variants only): record index of the AlphaRec of triangle slot 0, the alpha planes
    uint32_t                  alpha_base;
    uint32_t                  alpha_threshold;   // run the AlphaRec -> texel-address block once this many lanes wait for it
    const uint8_t*            alpha_texels;
    uint32_t            xcd_cursors;      // persistent mode: 1 = one cursor per XCD (block_cursor[16 * xcd]), each over its own
                                          // eighth of the ray blocks, with stealing; 0 = one global cursor
    uint32_t            nblocks;          // ray blocks in the batch
    uint32_t            max_claim;        // most ray blocks one cursor atomic may claim (guided self-scheduling), >= 1
    const uint32_t*     reserved_cus;     // persistent mode: 1024-bit set of __smid() values of the reserved CUs, or NULL
    uint32_t*           cu_slots;         // 1024 counters (zeroed per launch): blocks that asked to stay on a reserved CU
    uint32_t            reserved_limit;   // blocks a reserved CU keeps (0 = none)
    const uint32_t*     live_n;           // trace_kernel_devn only: rays the launch's one batch really holds (<= seg0.n, which sized the launch)
};

// One per triangle slot of a scene with alpha-tested triangles, in the scene's record array behind the triangles (so
// the walk fetches it like any other record): what Primitives.h:196-208 needs of the triangle and of its material.
struct AlphaRec {
    float    uv[3][2];        // vt_tri_attribs::uv
    float    m[2][3];         // per row of Material::baseTexMat: [0], [1], [2] + [3]
    float    tex_scale, alpha_ref;
    uint32_t dims;            // width | height << 16; 0 = no texture (alpha 1); kAlphaAlwaysPass = material out of range
    uint32_t offset_filter;   // first texel of the plane (31 bits) | filter << 31
};
static_assert(sizeof(AlphaRec) == 64, "an AlphaRec is one 64-B record");
constexpr uint32_t kAlphaAlwaysPass = 0xFFFFFFFFu;
struct AlphaRecArgs { const vt_tri64* tris; const vt_tri_attribs* attribs; const vt_alpha_material* mats; uint32_t n_mats;
                      AlphaRec* out; uint32_t n; };
hipError_t launch_alpha_records(const AlphaRecArgs& a, hipStream_t stream);

struct HitAttrsArgs {
    const vt_tri64* tris;
    const uint32_t* prim_to_slot;
    const vt_ray*   rays;
    const vt_hit*   hits;
    vt_hit_attrs*   attrs;
    uint64_t        n;
};

struct HitShadeArgs {
    const vt_tri_attribs* attribs;   // original triangle order
    const vt_hit*         hits;
    vt_hit_shade*         out;
    uint64_t              n;
};

struct GenPrimaryArgs { vt_camera cam; vt_ray* rays; };
struct GenBounceArgs { const vt_hit_attrs* attrs; vt_ray* rays; uint64_t n; uint64_t seed; };
hipError_t launch_gen_primary(const GenPrimaryArgs& a, hipStream_t stream);
hipError_t launch_gen_bounce(const GenBounceArgs& a, hipStream_t stream);

// one depth of the bounce loop: m queue entries (rays_q[j], hits_q[j], path id ids_q[j] or j when NULL)
struct QueueArgs {
    const vt_tri64* tris; const uint32_t* prim_to_slot;
    const vt_ray* rays_q; const vt_hit* hits_q; const uint32_t* ids_q; uint64_t m;
    const uint32_t* m_dev;     // NULL, or: the queue really holds *m_dev <= m entries (m sized the launch)
    vt_hit* hits_out;          // this depth's row of the result (indexed by path), or NULL when hits_q already is it
    vt_ray* rays_next; uint32_t* ids_next;   // next queue, or NULL at the last depth
    uint32_t* block_offsets;   // scratch: one entry per 256 queue entries
    uint64_t seed;
};
hipError_t launch_queue_step(const QueueArgs& a, uint32_t* live_out, hipStream_t stream);
// every record of a result row becomes a miss -- unless (count != NULL) *count == n: then every path is still in the queue and the
// queue step is about to write the whole row
hipError_t launch_fill_miss(vt_hit* hits, uint64_t n, const uint32_t* count, hipStream_t stream);

struct RefitTrisArgs { const float* verts; const uint8_t* flags; const uint32_t* prim_to_slot; vt_tri64* tris; uint32_t n;
                       uint32_t* bad; /* counts triangles with a non-finite vertex */ };
struct RefitLevelArgs { vt_node_pair* pairs; const vt_tri64* tris; const uint32_t* level_pairs; uint32_t count; };
hipError_t launch_refit_tris(const RefitTrisArgs& a, hipStream_t stream);
struct SkinMatricesArgs { const float* bones; const float* binds; float* mats; uint32_t nmat; };
struct SkinTrisArgs {
    const float* bind_verts; const vt_skin_vertex* skin; const uint32_t* matrix_base; const float* mats;
    const uint32_t* prim_to_slot; vt_tri64* tris; uint32_t n; uint32_t nmat;
    uint32_t* bad;   // counts triangles with a non-finite vertex
};
hipError_t launch_skin_matrices(const SkinMatricesArgs& a, hipStream_t stream);
hipError_t launch_skin_tris(const SkinTrisArgs& a, hipStream_t stream);
hipError_t launch_refit_level(const RefitLevelArgs& a, hipStream_t stream);

size_t     trace_lds_bytes(uint32_t lds_entries, bool fetch_dma);
hipError_t launch_trace(const TraceArgs& a, bool any_hit, bool stats, bool persistent, bool fetch_dma, bool alpha,
                        uint32_t grid_blocks, size_t lds_bytes, hipStream_t stream);
hipError_t trace_blocks_per_cu(bool any_hit, bool stats, bool persistent, bool fetch_dma, bool alpha, size_t lds_bytes, int* out);
// ALPHA variants: false when the loaded code object gives one of them more VGPRs than the texel-register reservation allows
hipError_t alpha_kernels_within_budget(bool* ok);
hipError_t launch_hit_attrs(const HitAttrsArgs& a, hipStream_t stream);
// every CU that receives a block sets bit __smid() of the 1024-bit set `seen` (32 words, zeroed by the caller)
hipError_t launch_cu_probe(uint32_t* seen, uint32_t blocks, hipStream_t stream);
hipError_t launch_hit_shade(const HitShadeArgs& a, hipStream_t stream);

} // namespace vt
