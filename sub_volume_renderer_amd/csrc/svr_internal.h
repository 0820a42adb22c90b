// Internal declarations shared by the HIP translation units of libsvr_hip.so.
// gfx950 (MI355X) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/svr.h"

// Experiment surface: environment switches and the results-changing timing bits of svr_set_variant exist only in
// builds made with -DSVR_EXPERIMENTS (tools/ab_build.py name=-DSVR_EXPERIMENTS); the shipped library reads no
// environment variable that alters the march and rejects those variant bits.
#ifdef SVR_EXPERIMENTS
#include <stdlib.h>
static inline int  svr_exp_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool svr_exp_env_set(const char* name) { return getenv(name) != nullptr; }
#define SVR_EXP_VARIANT_BITS 0x1800      /* bit 11: do not wait for brick loads; bit 12: skip the march loop (WRONG pixels) */
#else
static inline int  svr_exp_env_int(const char*, int dflt) { return dflt; }
static inline bool svr_exp_env_set(const char*) { return false; }
#define SVR_EXP_VARIANT_BITS 0
#endif

// ---------------------------------------------------------------------------
// Kernel parameter block.  Passed BY VALUE as the single kernel argument: it
// lives in the kernarg segment and is read with scalar loads, so every field is
// wave-uniform SGPR data (the reference reads the same values from uniform
// buffers each step: raycast.wgsl:16-20, sample_vol.wgsl:7,14-15).
// ---------------------------------------------------------------------------
struct LodParams {
    const void*     density;   // ring texture r32float (or u8 storage), [z][y][x]
    const uint32_t* labels;    // ring texture r32uint
    int32_t  off[3];           // current_logical_offset_in_pixels (x,y,z)
    uint32_t shape[3];         // current_logical_shape_in_pixels
    uint32_t wrap0[3];         // off - floor(off/ring)*ring  (ring slot of the ROI's first voxel)
    uint32_t ring[3];          // ring extent
    float    scale[3];         // scale_factor
    uint32_t base_bytes;       // byte offset of this LOD's density ring inside the buffer resource (rbase, rbytes) below
    int32_t  addw[3];          // wrap0 - off: ring slot = wrap(ic + addw)
    uint32_t rx4;              // row pitch of the density ring in bytes (ring[0] * element size)
    float    ss[3];            // size * scale (the fused per-axis factor of the fast paths when scale = 2^-k)
    int32_t  slab;             // brick slab length in iterations (0: this LOD never stages bricks)
    // empty-space skipping: the 2x2x2-block maxima of this LOD's macro cells inside MarchParams::cells_all
    // the buffer resource this LOD's texels are fetched through: the one allocation of all LODs' rings while it is
    // below 4 GiB (one resource serves every lane of a mixed-LOD gather), else this LOD's ring alone
    const void* rbase;
    uint32_t rbytes;
    // a ring of 4 GiB or more (float voxels of a 2048^3-class volume) is addressed through `nparts` (<= 8) resources of
    // `zsplit` ring z planes each (the last one holds the rest): part p starts at rbase + p * part_bytes and takes
    // offsets relative to its own start; `rbytes` then is the size of a full part, `rbytes_last` that of the last one
    // (nparts = 1, zsplit = ring z where one resource reaches the whole ring)
    uint32_t nparts;
    uint32_t zsplit;
    uint32_t part_bytes;
    uint32_t rbytes_last;
    // svr_lod_desc::blocked_twin: a second copy of this ring in 128-byte micro-blocks (svr_blocked_index).  It is split into
    // parts exactly like the ring (same zsplit — a whole number of blocks —, part sizes and count)
    uint32_t twin;             // 0: no copy; 1: waves with many rows per gather take it instead of staging bricks; 2: only when they stage none
    uint32_t twin_rbytes;      // size of the copy's own buffer resource (of a full part of it where the ring is cut into parts)
    const void* twin_rbase;    // ... which starts here
    uint32_t cell_base;        // byte offset of the LOD's cell grid
    uint32_t cdim[3];          // cells per axis
    int32_t  cshift;           // log2 of the cell size (3 or 2)
    int32_t  skip_batches;     // batches of 8 iterations between two cell tests (0: never skip on this LOD)
};

struct MarchParams {
    float ndc_to_data[16];     // world_inv * cam_inv * proj_inv   (vs_main.wgsl:22)
    float pc[16];              // proj * cam                        (vs_main.wgsl:19)
    float world[16];           // world_transform                   (fs_main.wgsl:62)
    float size[3];             // volume_dimensions (x,y,z)
    float rel_step;            // fs_main.wgsl:20
    svr_frame frame;
    // u_material
    float clim0, clim1, gamma, opacity;
    float lmip_threshold, lmip_fall_off;
    uint32_t lmip_threshold_raw;   // integer rings: smallest value v with (float)v >= lmip_threshold (max + 1: none)
    int32_t lmip_max_samples;
    int32_t render_mode;           // svr_render_mode
    float weight_falloff;          // SVR_MODE_WEIGHTED_AVERAGE: w = max(1 - weight_falloff * d, 0)^2
    float fog_density;
    float fog_color[3];
    uint32_t color_count;
    const float* colors;       // device, color_count x vec4
    int32_t colorspace_srgb;
    int32_t num_lods;
    // Material.clipping_planes (pygfx.clipping_planes.wgsl, included at fs_main.wgsl:8)
    uint32_t clip_count;
    int32_t  clip_all;
    float    clip[SVR_MAX_CLIP_PLANES][4];
    // outputs (device)
    float*    rgba;
    float*    depth;
    uint32_t* label;
    uint8_t*  flags;
    uint32_t* steps;
    unsigned long long* pick; uint32_t pick_id;
    uint32_t* dbg;                 // batch census of the instrumented build (8 counters) or NULL
    // all LOD density rings live in ONE allocation so a single buffer resource
    // (32-bit byte offsets, hardware range check) addresses every LOD
    const void* cells_all;         // dilated macro-cell maxima of all LODs (null: no skipping)
    uint32_t cells_all_bytes;
    const void* density_all;
    uint32_t density_all_bytes;    // size of that allocation when it is below 4 GiB (per_lod_rsrc = 0)
    int32_t  span_ok;              // 0: some ring is beyond the span kernel's 32-bit / 24-bit addressing: straightforward kernel
    int32_t  per_lod_rsrc;         // 1: the rings together are 4 GiB or more: one buffer resource per LOD
    int32_t  density_esh;          // log2 of the density element size: 0 u8, 1 u16, 2 f32 (svr_lod_desc::density_storage)
    // block -> tile mapping
    int32_t tiles_x, tiles_y;
    int32_t tile_log2w;            // wave tile = (1 << tile_log2w) x (64 >> tile_log2w) pixels
    int32_t brick;                 // 0 never / 1 per-wave probe / 2 always: LDS bricks (u8 rings only)
    int32_t brick_lod_mask;        // LODs allowed to use bricks (bit l)
    int32_t slab_long;             // 1: brick slabs start at twice their plain length
    int32_t skip_flags;            // empty-space skipping policy bits (1: short march while lanes only follow a maximum; 2: MIP-like
                                   // uniforms — the machine never stops — so a lane that follows a maximum passes blocks that cannot beat it)
    int32_t brick_bytes;           // LDS bytes per wave for brick staging (also what caps the waves per CU: 160 KiB / it)
    int32_t brick_pow2;            // -DSVR_EXPERIMENTS builds only: round the brick row pitch up to a power of two (the round-1 layout)
    int32_t block_waves_log2;      // span kernel: block = (1 << this)^2 wave tiles (0: one wave per block)
    const uint32_t* tile_order;    // blockIdx -> tile index (null: contiguous run of tiles per XCD)
    int32_t dbg_nowait;            // -DSVR_EXPERIMENTS builds only (WRONG results): bit 0 do not wait for the brick loads, bit 1 skip the march loop
    int32_t brick_lines;           // probe threshold: estimated L1 lookups per wave-load above which bricks are staged
    int32_t orient;                // 1: lane order follows the screen direction of the volume's x axis
    float   xdir[4];               // clip-space image of the data-space direction (1,0,0,0)
    int32_t lod_pow2[SVR_MAX_LODS];// 1: all three scale factors of the LOD are powers of two and the
                                   //    voxel indices fit the 24-bit multiplier (fast path eligible)
    LodParams lod[SVR_MAX_LODS];
};

struct LodStorage {
    int32_t  ring[3];          // x,y,z
    size_t   voxels;
    void*     density;
    void*     twin;            // micro-block copy of `density` (svr_lod_desc::blocked_twin) or null
    int32_t   twin_policy;     // svr_lod_desc::blocked_twin as given: 1 instead of bricks, 2 where no bricks are staged
    uint32_t* labels;
    svr_lod_state state;
    // macro-cell maxima (8^3 or 4^3 slots per cell) for empty-space skipping; null when an extent is not a multiple of 8
    void*     cells_raw;
    void*     cells_dil;       // maxima over the 2 x 2 x 2 block of cells starting at each cell
    int32_t   cdim[3];
    int32_t   cshift;          // log2 of the cell size
    size_t    cell_base;       // element offset of this LOD's grid inside svr_ctx::cells_dil_all
};

struct StagingSlot {
    void* host;                // pinned
    void* dev;
    hipEvent_t done;           // scatter kernel that consumed this slot has finished
    bool used;
};

struct svr_ctx {
    int device;
    int num_lods;
    LodStorage lod[SVR_MAX_LODS];
    void*     density_all;           // one allocation, LOD rings at 256-byte aligned offsets
    void*     twin_all;              // the micro-block copies of the rings that keep one (svr_lod_desc::blocked_twin)
    size_t    twin_all_bytes;
    int       density_storage;       // ring element type: SVR_F32 (reference layout), SVR_U8 or SVR_U16
    int       density_u8;            // density_storage == SVR_U8
    int       no_labels;             // the volume has no segmentation: no label rings
    uint64_t  staged_bytes;          // bytes sent through the pinned staging slots so far (diagnostics)
    double    upload_seconds;        // host wall-clock time spent inside svr_upload_region
    uint32_t* labels_all;
    void*     cells_raw_all;         // macro-cell maxima of all LODs (element type = the density ring's)
    void*     cells_dil_all;
    size_t    cells_all_bytes;
    size_t    density_all_bytes;
    size_t    lod_base_bytes[SVR_MAX_LODS];
    hipStream_t render_stream;
    hipStream_t upload_stream;
    hipEvent_t  uploads_published;   // recorded on upload_stream by svr_publish_uploads
    bool        have_published;
    // material (device copy of the colour table)
    svr_material material;
    std::vector<float> colors_host;
    float* colors_dev;
    std::vector<float*> colors_retired;   // tables replaced while renders may still read them
    uint32_t colors_cap;
    bool material_set;
    int variant;
    // pinned staging ring for host uploads
    static constexpr int kSlots = 3;
    size_t slot_bytes;
    StagingSlot slot[kSlots];
    int next_slot;
    hipEvent_t ev_a, ev_b;           // timing
    // One event per stream that has carried a render: uploads must wait for EVERY render still in flight
    // (frames are kept in flight on several streams), not only for the most recently enqueued one.
    struct RenderMark { hipStream_t stream; hipEvent_t done; bool pending; };
    std::vector<RenderMark> render_marks;
    std::mutex marks_mu;             // render thread records, upload thread waits
    std::mutex upload_mu;            // the staging slots and the upload stream's enqueue order: one uploader at a time
    std::vector<float> clip_host;    // clipping planes (abcd) of the current material
    uint32_t*  dbg_dev;              // 8 diagnostic counters (instrumented renders)
    // block -> tile tables (one per frame tiling and policy; entries are never rewritten)
    struct TileOrder { int tiles_x, tiles_y, mode; uint32_t* dev; };
    std::vector<TileOrder> tile_orders;
    // cost-sorted block -> tile tables (default placement): one per stream that carries renders, rewritten (through
    // pinned memory, on that stream) whenever the camera or the frame region changes
    // (`drawn`: recorded behind every draw that reads the table — what a recycled slot waits for; a recycled entry's
    // stream handle is never used again: its owner may have destroyed the stream)
    struct CostOrder { hipStream_t stream; uint32_t* dev; uint32_t* host; size_t cap; hipEvent_t copied; uint64_t key; bool valid;
                       hipEvent_t drawn; bool drawn_set; };
    std::vector<CostOrder> cost_orders;
    hipEvent_t uploads_marker;       // svr_mark_uploads / svr_uploads_pending
    std::atomic<bool> marker_set;
    // svr_upload_ticket / svr_ticket_pending: ticket t lives in tickets[t % kTickets]
    static constexpr int kTickets = 64;
    hipEvent_t tickets[kTickets];
    uint64_t   next_ticket;          // next ticket to hand out (first is 1)
    std::mutex ticket_mu;
    // RCCL communicator of svr_comm_init (comm_rccl.hip); opaque here
    void* comm;
    int   comm_rank, comm_size;
};

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// error plumbing -------------------------------------------------------------
void svr_set_error(const std::string& msg);
#define SVR_HIP_TRY(expr)                                                            \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            svr_set_error(std::string(#expr) + ": " + hipGetErrorString(e_));        \
            return SVR_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

// kernels (march_kernel.hip / ring_kernels.hip) ------------------------------
hipError_t svr_launch_march(const MarchParams& p, int variant, hipStream_t stream);
hipError_t svr_launch_compose(const float* rgba, const float* depth, const uint8_t* flags, int w, int h,
                              const svr_compose_params& q, uint8_t* out_rgba8, float* zbuf, hipStream_t stream);

struct ScatterArgs {
    const void* src_density; int density_dtype; int64_t dstride[3];   // bytes per x,y,z step
    const void* src_labels;  int labels_dtype;  int64_t lstride[3];
    void* ring_density; uint32_t* ring_labels;
    void* ring_twin;               // micro-block copy of the density ring, written alongside (null: none)
    int32_t ring_storage;          // svr_dtype of the density ring: SVR_U8 / SVR_U16 / SVR_F32
    int32_t packed;                // the source is a staged block (x stride = element size, rows back to back)
    int32_t ring[3];
    int32_t dst_off[3];
    int32_t shape[3];
};
hipError_t svr_launch_scatter(const ScatterArgs& a, hipStream_t stream);
hipError_t svr_launch_cell_update(const void* ring, int storage, const int32_t ring_dims[3], void* raw, void* blk,
                                  const int32_t cdim[3], int cshift, const int32_t off[3], const int32_t shape[3],
                                  hipStream_t stream);
hipError_t svr_launch_gather(const void* ring_density, int ring_storage, const uint32_t* ring_labels, const int32_t ring[3],
                             const int32_t off[3], const int32_t shape[3],
                             float* out_density, uint32_t* out_labels, hipStream_t stream);
hipError_t svr_launch_untile_grid(const void* gathered, void* frame_out, int frame_w, int frame_h,
                                  int tile_w, int tile_h, int grid_x, int grid_y, int elem_bytes, hipStream_t stream);
hipError_t svr_launch_untile(const void* gathered, void* frame_out, int frame_w, int frame_h,
                             int band_h, int nranks, int out_h, int elem_bytes, hipStream_t stream);

hipError_t svr_launch_pool2x(const void* src, void* dst, const int32_t dims[3], int dtype, int mode, hipStream_t stream);

size_t svr_dtype_size(int dtype);

// Element index of ring slot (x, y, z) in the micro-block copy of a ring of Rx x Ry slots per plane (esh = log2 of the
// element size): 128-byte blocks of 8 x 4 x 4 (esh 0), 4 x 4 x 4 (1) or 4 x 4 x 2 (2) slots, blocks in [bz][by][bx] order,
// the slots of a block in [z][y][x] order.  Ring extents are multiples of (8, 4, 4).
__host__ __device__ inline size_t svr_blocked_index(int esh, uint32_t Rx, uint32_t Ry, uint32_t x, uint32_t y, uint32_t z) {
    const uint32_t XB = esh == 0 ? 3u : 2u, YB = 2u, ZB = esh == 2 ? 1u : 2u;
    const size_t blk = ((size_t)(z >> ZB) * (size_t)(Ry >> YB) + (size_t)(y >> YB)) * (size_t)(Rx >> XB) + (size_t)(x >> XB);
    const uint32_t inb = ((((z & ((1u << ZB) - 1u)) << YB) | (y & 3u)) << XB) | (x & ((1u << XB) - 1u));
    return (blk << (7 - esh)) + inb;
}
// integer rings hold the source values themselves: only sources of that very dtype may be uploaded
inline bool storage_accepts(int storage, int src_dtype) { return storage == SVR_F32 || src_dtype == storage; }
