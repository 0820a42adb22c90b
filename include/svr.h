/*
 * svr.h — C ABI of the MI355X-native sub-volume LMIP ray-march renderer.
 *
 * The reference (gyoge0/sub_volume_renderer) has no FFI boundary of its own:
 * its hot path sits behind pygfx's plugin hooks — uniform buffers, texture
 * bindings and a WGSL fragment shader (src/sub_volume/_shader.py:75-127).
 * This header is the C-ABI replacement of exactly those hooks.  Every entry
 * point cites the reference interface it replaces.  All vectors at this
 * boundary are in SHADER ORDER (x, y, z) = reversed numpy order, exactly as
 * the reference writes them into its uniform buffers
 * (_wrapping_buffer.py:84-96,113-115; _wobject.py:121-123).  Ring memory is
 * C-contiguous [z][y][x] (texel (x,y,z) == numpy data[z,y,x], KNOWLEDGE.md:81-133).
 *
 * Plain pointers and sizes only; no torch / HIP types in any signature
 * (streams travel as void*).  Every function returns 0 on success and a
 * negative svr_status on failure; svr_last_error() gives the message.
 *
 * Threads.  A context has ONE render thread: svr_set_lod_state, svr_set_material, svr_set_variant, svr_render,
 * svr_time_render, svr_gather_tiles and the svr_comm_* calls of one context must not run concurrently with one
 * another (renders may go to different HIP streams, one after the other).  svr_upload_region / _device,
 * svr_upload_ticket, svr_ticket_pending, svr_mark_uploads and svr_uploads_pending may be called from ONE other
 * thread at the same time (the streaming worker): uploads of a context are serialised by a mutex, and they are
 * ordered on the device behind every render still in flight.  Different contexts are independent.
 * svr_last_error() is per thread.
 */
#ifndef SVR_H
#define SVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVR_MAX_LODS 8
#define SVR_MAX_CLIP_PLANES 8
#define SVR_ABI_VERSION 7

typedef enum svr_status {
    SVR_OK = 0,
    SVR_ERR_INVALID = -1,   /* bad argument (maps to ValueError in the Python mirror) */
    SVR_ERR_HIP = -2,       /* a HIP runtime call failed */
    SVR_ERR_NOMEM = -3,     /* device or pinned allocation failed */
    SVR_ERR_RANGE = -4      /* region outside the ring / frame */
} svr_status;

/* element types accepted for source arrays (np.array(src, dtype=f32/u32) in
 * _wrapping_buffer.py:325,330 accepts anything numpy can cast) */
typedef enum svr_dtype {
    SVR_U8 = 0, SVR_U16 = 1, SVR_U32 = 2, SVR_U64 = 3,
    SVR_I8 = 4, SVR_I16 = 5, SVR_I32 = 6, SVR_I64 = 7,
    SVR_F32 = 8, SVR_F64 = 9
} svr_dtype;

/* One LOD's ring buffer = the two gfx.Texture objects of a WrappingBuffer
 * (_wrapping_buffer.py:50-59): r32float density + r32uint labels, both of
 * extent ring_dims (shader order), zero-initialised. */
typedef struct svr_lod_desc {
    int32_t ring_dims[3];          /* (x, y, z) voxels = reversed shape_in_pixels */
    int32_t density_storage;       /* SVR_F32: the reference's r32float texture.  SVR_U8 / SVR_U16: store
                                      the density ring in the sources' own integer type — allowed only when
                                      every upload's source has that dtype, so that texel values (exact in
                                      f32) and therefore all results are identical; 4x / 2x less HBM / L2 /
                                      LDS per voxel.  All LODs of a context use the same storage. */
    int32_t no_labels;             /* 1: the volume has no segmentation (FUTURE.md:178-193 "make this optional"): no label
                                      ring is allocated (4 of the 5 bytes per slot), uploads must pass labels = NULL,
                                      every hit gets label 0 (-> colors[0], like unlabelled voxels: FUTURE.md:170-176).
                                      All LODs of a context alike. */
    int32_t blocked_twin;          /* 1 or 2: keep a SECOND copy of this LOD's density ring, laid out in 128-byte micro-blocks that
                                      are compact in 3-D (8 x 4 x 4 one-byte voxels, 4 x 4 x 4 two-byte, 4 x 4 x 2 four-byte;
                                      blocks in [bz][by][bx] order) — the locality a texture unit's tiled 3-D layout gives the
                                      reference's textureLoad (sample_vol.wgsl:24).  Every upload writes both copies; the march
                                      reads the same texels from whichever copy suits a wave's view: waves whose gathers would
                                      touch many 128-byte ROWS per load (the per-wave probe that otherwise stages LDS bricks)
                                      gather from the micro-blocks — with 1 INSTEAD of staging bricks from the rows (what pays on
                                      the finest LOD), with 2 only where the wave stages none (the LOD cannot, or the wave's boxes
                                      stopped fitting its LDS region: what pays on the coarser LODs).  Costs one more density
                                      element per voxel of HBM and of upload traffic; results are identical.  The copies have an
                                      allocation of their own (they never push the rings over the 4 GiB below which one buffer
                                      resource reaches every LOD).  ring_dims must be multiples of (8, 4, 4), else SVR_ERR_INVALID.
                                      The Python mirror's "auto": 1 on the finest LOD; "all": also 2 on the others. */
} svr_lod_desc;

/* == u_wrapping_buffer_i uniform (_wrapping_buffer.py:15-19), shader order.
 * offset == shape == 0 means "ROI is None": nothing is in bounds (:90-96). */
typedef struct svr_lod_state {
    int32_t offset[3];             /* current_logical_offset_in_pixels */
    int32_t shape[3];              /* current_logical_shape_in_pixels  */
    float   scale[3];              /* scale_factor                     */
} svr_lod_state;

/* == u_material uniform (_material.py:6-24 + inherited VolumeMipMaterial
 * clim/gamma/opacity). colors is n x vec4 (h, s, v, pad) as the reference
 * pads it (_material.py:139-159). */
typedef struct svr_material {
    float    clim[2];
    float    gamma;
    float    opacity;
    float    lmip_threshold;
    float    lmip_fall_off;
    int32_t  lmip_max_samples;     /* i32 in the reference (_material.py:10-11) */
    float    fog_density;
    float    fog_color[3];
    uint32_t color_count;
    const float* colors;           /* host pointer, color_count * 4 floats */
    int32_t  colorspace_srgb;      /* 1: apply srgb2physical (raycast.wgsl:71-72; texture default) */
    /* Material.clipping_planes / clipping_mode of pygfx (the `pygfx.clipping_planes.wgsl` include at
     * fs_main.wgsl:8): world-space planes (a, b, c, d); the fragment — here: the whole ray, because the
     * tested position is the interpolated world position of the proxy box's BACK face (vs_main.wgsl:27) —
     * is discarded when dot(world_pos, abc) < d holds for ANY plane (clipping_mode_all = 0) or for ALL of
     * them (1).  No planes (the default): nothing is clipped.  pygfx text restated: assumption A6. */
    uint32_t clipping_plane_count; /* 0 .. SVR_MAX_CLIP_PLANES */
    int32_t  clipping_mode_all;
    const float* clipping_planes;  /* host pointer, clipping_plane_count * 4 floats (may be NULL when the count is 0) */
    /* Which raycast runs (FUTURE.md:111-120 "swappable rendering pipeline": a material-level switch).
     * SVR_MODE_LMIP: raycast.wgsl as it stands (MIP is LMIP with threshold -inf, fall-off 0 and no sample limit).
     * SVR_MODE_WEIGHTED_AVERAGE: the mode FUTURE.md:97-109 asks for and leaves without a formula; defined HERE
     * (oracle twin: oracle/lmip_oracle.c `raycast_weighted_average`) as
     *     d_i = f32(i) * |step|          distance of sample i from the ray's entry, in the units of the fog distance
     *     t_i = max(1 - weight_falloff * d_i, 0),  w_i = t_i * t_i
     *     value = (sum w_i * s_i) / (sum w_i)      over the samples of the ray, in order, f32
     * shown at the sample with the largest w_i * |s_i| (first one wins; it gives depth, label and fog distance);
     * nothing but zeros along the ray: a miss.  The ray ends at the first sample with t_i == 0. */
    int32_t  render_mode;          /* svr_render_mode */
    float    weight_falloff;       /* >= 0; only read in SVR_MODE_WEIGHTED_AVERAGE */
} svr_material;

typedef enum svr_render_mode { SVR_MODE_LMIP = 0, SVR_MODE_WEIGHTED_AVERAGE = 1 } svr_render_mode;

/* == the six mat4 of u_stdinfo / u_wobject that the shaders read
 * (vs_main.wgsl:18-22, fs_main.wgsl:62-63) + u_wobject.volume_dimensions
 * (_wobject.py:14-17).  Matrices are COLUMN-MAJOR float[16] (m[c*4+r]) like
 * WGSL mat4x4<f32>. */
typedef struct svr_camera {
    float world[16];
    float world_inv[16];
    float cam[16];
    float cam_inv[16];
    float proj[16];
    float proj_inv[16];
    float volume_dimensions[3];    /* shader order (x, y, z) */
} svr_camera;

/* Which pixels of the full frame this call renders.  Output row r (0..out_h)
 * and column c (0..out_w) map to the frame pixel
 *     x = x0 + c
 *     y = y0 + (r / band_h) * band_pitch + (r % band_h)
 * A plain tile is band_h = out_h (band_pitch unused).  Interleaved stripes
 * for multi-GPU load balance use band_h = stripe height and
 * band_pitch = stripe height * number of ranks.  Rows with y >= frame_h are
 * padding: written as "discarded". */
typedef struct svr_frame {
    int32_t frame_w, frame_h;      /* full frame: NDC uses these (vs_main.wgsl:19) */
    int32_t x0, y0;
    int32_t out_w, out_h;
    int32_t band_h, band_pitch;
} svr_frame;

/* pixel classification written to svr_outputs.flags */
#define SVR_PIX_DISCARD 0   /* no back-face fragment, or nsteps < 1 (fs_main.wgsl:44) */
#define SVR_PIX_MISS    1   /* fragment ran, no significant value (fs_main.wgsl:93-98) */
#define SVR_PIX_HIT     2   /* fragment ran, LMIP found a local maximum (fs_main.wgsl:56-87) */

/* DEVICE pointers, caller-allocated, each out_h*out_w elements (rgba: x4).
 * rgba is the fragment's out.color before blending (fs_main.wgsl:86,95);
 * discarded pixels get (0,0,0,0).  Any pointer except rgba may be NULL. */
typedef struct svr_outputs {
    float*    rgba;
    float*    depth;               /* out.depth (fs_main.wgsl:72,97) */
    uint32_t* label;               /* render_out.segmentation (raycast.wgsl:81) */
    uint8_t*  flags;               /* SVR_PIX_* */
    uint32_t* steps;               /* executed iterations of raycast.wgsl:29-62 per pixel
                                      (instrumented build of the kernel; NULL in production) */
    uint64_t* pick;                /* out.pick of the `write_pick` shader variant (fs_main.wgsl:89-92): the 64 bits of
                                      pygfx's rgba16uint pick target, component k = bits 16k..16k+15:
                                      [0,20) wobject id, [20,34) u32(coord.x*16383), [34,48) .y, [48,62) .z, each
                                      clipped to its width (pygfx `pick_pack`, restated); 0 where nothing was hit */
    uint32_t  pick_id;             /* u_wobject.id */
} svr_outputs;

typedef struct svr_ctx svr_ctx;

/* ---- lifecycle: replaces WrappingBuffer.__init__ texture/uniform creation
 * (_wrapping_buffer.py:47-70) for all LODs of one SubVolume (_wobject.py:72-91) */
int  svr_create(int device, int num_lods, const svr_lod_desc* lods, svr_ctx** out_ctx);
int  svr_destroy(svr_ctx* ctx);
const char* svr_last_error(void);
int  svr_abi_version(void);

/* ---- uniforms */
/* _current_logical_roi_in_pixels setter + scale_factor setter
 * (_wrapping_buffer.py:77-97,105-116).  Takes effect for renders enqueued
 * after the call. */
int  svr_set_lod_state(svr_ctx* ctx, int lod, const svr_lod_state* st);
int  svr_get_lod_state(svr_ctx* ctx, int lod, svr_lod_state* st);
/* SubVolumeMaterial uniform block (_material.py:60-159) */
int  svr_set_material(svr_ctx* ctx, const svr_material* m);

/* ---- texture upload: replaces texture.data[dst] = np.array(src, f32|u32);
 * texture.update_range(...) for BOTH textures (_wrapping_buffer.py:325-335).
 * dst_off/shape: ring voxels, shader order; must lie inside the ring.
 * density/labels: HOST pointers to element (0,0,0) of the source block;
 * strides in BYTES per (x, y, z) step.  Conversion is numpy's cast (to f32 /
 * to u32 with wrap-around).  The copy is staged through pinned memory and
 * enqueued on the context's upload stream; the call returns once the source
 * has been consumed (source may be freed), not when the device copy is done.
 * Either source may be NULL to leave that texture untouched. */
int  svr_upload_region(svr_ctx* ctx, int lod,
                       const int32_t dst_off[3], const int32_t shape[3],
                       const void* density, int density_dtype, const int64_t density_strides[3],
                       const void* labels,  int labels_dtype,  const int64_t labels_strides[3]);
/* same, but the sources are DEVICE pointers (backing volume resident in HBM) */
int  svr_upload_region_device(svr_ctx* ctx, int lod,
                       const int32_t dst_off[3], const int32_t shape[3],
                       const void* density, int density_dtype, const int64_t density_strides[3],
                       const void* labels,  int labels_dtype,  const int64_t labels_strides[3]);
/* Make everything uploaded so far visible to subsequent renders: the render
 * stream waits (device-side) on an event recorded on the upload stream.  The
 * reference has no equivalent: pygfx flushes update_range uploads before the
 * next draw on the same queue (FUTURE.md:47-58). */
int  svr_publish_uploads(svr_ctx* ctx);

/* ---- asynchronous streaming (the reference has none: FUTURE.md:3-67 lists it as future work).
 * Protocol, driven by the host (SubVolume.center_on_position(..., asynchronous=True)):
 *   1. svr_set_lod_state(ROI := old ROI intersected with the new one)  -- renders enqueued from now on
 *      never touch a ring slot that is about to be overwritten; coarser LODs cover the gap;
 *   2. svr_upload_region(...) for the new chunks, from a worker thread: the copies are ordered
 *      behind the last render that was enqueued with the old ROI, and run beside later renders;
 *   3. svr_mark_uploads(); when svr_uploads_pending() reports 0, svr_set_lod_state(ROI := new).
 * svr_render never waits for marked-but-unpublished uploads. */
int  svr_mark_uploads(svr_ctx* ctx);
int  svr_uploads_pending(svr_ctx* ctx, int* pending);
/* The same, for several loads in flight at once (one per LOD: the coarse levels are uploaded first and
 * published as soon as THEY have landed, FUTURE.md:86-95): svr_upload_ticket records an event behind
 * everything enqueued on the upload stream so far and returns its ticket (> 0); svr_ticket_pending
 * reports 1 while that event has not been reached.  Tickets older than the 64 most recent ones are
 * reported as done only after the oldest live one is (their event has been reused). */
/* diagnostics: bytes that went through the pinned staging slots so far and the host wall-clock seconds spent
 * inside svr_upload_region (packing rows, waiting for a free slot, enqueueing); reset = 1 zeroes both */
int  svr_upload_stats(svr_ctx* ctx, uint64_t* staged_bytes, double* seconds_in_calls, int reset);
int  svr_upload_ticket(svr_ctx* ctx, uint64_t* ticket);
int  svr_ticket_pending(svr_ctx* ctx, uint64_t ticket, int* pending);

/* ---- readback of a ring region into packed host arrays (shader-order
 * shape, x fastest): the texture.data numpy mirror the reference's tests read
 * (tests/wrapping_buffer/test_load_logical_roi.py:5-76).  Synchronous. */
int  svr_read_region(svr_ctx* ctx, int lod, const int32_t off[3], const int32_t shape[3],
                     float* density_out, uint32_t* labels_out);
/* zero both textures of one LOD (fresh WrappingBuffer state) */
int  svr_clear_lod(svr_ctx* ctx, int lod);

/* ---- the draw: replaces renderer.render(scene, camera) for the
 * (SubVolume, SubVolumeMaterial) pair — vs_main.wgsl:6-50 + fs_main.wgsl:4-101
 * + raycast.wgsl:11-88 + sample_vol.wgsl + hsv_selection.wgsl.
 * Enqueued on `stream` (a hipStream_t passed as void*; NULL = the device's default
 * stream), i.e. ordered with the caller's other work on that stream.  Asynchronous. */
int  svr_render(svr_ctx* ctx, const svr_camera* cam, const svr_frame* frame,
                const svr_outputs* out, void* stream);
/* kernel variant selector for A/B measurement (results are identical for every value):
 * bits 0-3  kernel: 0 = default span march (one wave per workgroup), 1 = straightforward one-load-per-step
 *            march, 2 = span march with 2x2 waves per workgroup; +4 = brick slabs never start longer
 *            than their plain length (default: twice), +8 = no empty-space skipping in LMIP mode (default: waves skip
 *            stretches whose macro-cell maxima stay below the threshold while no lane tracks a maximum)
 * bits 4-7  1 + log2(wave tile width): wave tile = 2^k x 64/2^k pixels (0 = default 8x8)
 * bit  8    never stage LDS bricks nor gather from a micro-block twin (linear gathers only); bit 9: always
 *            (default: per-wave probe)
 * bit  10   keep row-major lane order (default: lanes follow the projected x axis)
 * bits 11-12 reserved: SVR_ERR_INVALID.  (Builds made with -DSVR_EXPERIMENTS — tools/ab_build.py, never the shipped
 *            library — use them for timing experiments that render WRONG pixels, and read the SVR_* environment
 *            switches listed in tools/README.md; the shipped library reads only SVR_PACK_THREADS, the number of host
 *            threads that pack upload blocks.)
 * bits 13-15 block -> tile placement: 0 = 64x64-pixel chunks of tiles sorted by the length of their rays for the draw's
 *            camera, longest first, dealt to the XCDs in snake order (default),
 *            1 = one contiguous run of tiles per XCD, 2..6 = single tiles, 64x32, 32x32, 128x64, 32x16 chunks in raster order,
 *            7 = 64x64 chunks in raster order (camera-independent; about 1 % faster than 0 when several frames are kept in
 *            flight on separate streams, 4 % slower for one frame at a time)
 * bits 16-23 probe threshold in L1 lookups per wave-load (0 = default 32)
 * bits 24-31 mask of LODs allowed to stage bricks (0 = default: all) */
int  svr_set_variant(svr_ctx* ctx, int variant);

/* ---- multi-GPU helper: scatter a rank-major gathered stripe buffer back
 * into a full frame (device pointers).  gathered holds nranks blocks of
 * out_h*out_w*elem_bytes each, laid out by svr_frame with x0 = 0,
 * y0 = rank*band_h, band_pitch = band_h*nranks. */
int  svr_untile_stripes(svr_ctx* ctx, const void* gathered, void* frame_out,
                        int frame_w, int frame_h, int band_h, int nranks,
                        int out_h, int elem_bytes, void* stream);

/* The same for config 3's literal geometry: a grid of grid_x x grid_y tiles of tile_w x tile_h pixels, tile
 * (tx, ty) rendered by rank ty * grid_x + tx (svr_frame with x0 = tx * tile_w, y0 = ty * tile_h, band_h = out_h =
 * tile_h, out_w = tile_w); gathered holds grid_x * grid_y blocks of tile_h * tile_w elements.  Tiles of the
 * last column / row may hang over the frame edge (their excess pixels are padding). */
int  svr_untile_grid(svr_ctx* ctx, const void* gathered, void* frame_out, int frame_w, int frame_h,
                     int tile_w, int tile_h, int grid_x, int grid_y, int elem_bytes, void* stream);

/* ---- the collective (the reference has none): every rank's rendered region -> root, over RCCL
 * (xGMI on one node).  One process per GPU, one context per process.  Rank 0 makes an id
 * (svr_comm_unique_id == ncclGetUniqueId, /opt/rocm/include/rccl/rccl.h:187), hands its SVR_COMM_ID_BYTES bytes to
 * the other ranks by any channel (a file, MPI, torch.distributed), and every rank calls svr_comm_init
 * (ncclCommInitRank, rccl.h:220; collective: returns once all ranks have joined).
 * svr_gather_tiles enqueues on `stream`, for each of `nplanes` planes (RGBA, and depth / label / flags when they
 * are wanted), the transfer of bytes_per_rank[p] bytes from every rank's local[p] into root's
 * gathered[p] + rank * bytes_per_rank[p]: grouped ncclSend / ncclRecv (rccl.h:700,722 — what ncclGather, rccl.h:745,
 * is made of), all planes and peers in one group.  DEVICE pointers; `gathered` is only read on root.  The call is
 * ordered after the work already on `stream` (the render that wrote local) and asynchronous. */
#define SVR_COMM_ID_BYTES 128
int  svr_comm_unique_id(char out_id[SVR_COMM_ID_BYTES]);
int  svr_comm_init(svr_ctx* ctx, const char id[SVR_COMM_ID_BYTES], int rank, int nranks);
int  svr_comm_destroy(svr_ctx* ctx);
int  svr_gather_tiles(svr_ctx* ctx, int nplanes, const void* const* local, void* const* gathered,
                      const size_t* bytes_per_rank, int root, void* stream);

/* ---- LOD pyramid builder (device pointers): one 2x2x2 pooling step with the rules of the reference's
 * offline builders — mode 0 = mean (scripts/create_mouse_multiscale.py:23-54; SVR_U8: floor(sum/8),
 * SVR_F32: pairwise sum * 0.125), mode 1 = max (SVR_U32 labels, scripts/create_platynereis_multiscale.py:86-134).
 * src_dims (x, y, z) must be even; dst has half the extent per axis.  Enqueued on `stream`. */
int  svr_pool2x(int device, const void* src, void* dst, const int32_t src_dims[3], int dtype, int mode, void* stream);

/* ---- display side: blend one render "over" a vertical-gradient background (top row =
 * bg_top), optional depth test (fragment passes if depth < inout_depth, then writes it), linear -> sRGB,
 * 8 bits per channel.  Restates what pygfx does after the fragment shader (blending src_alpha /
 * one_minus_src_alpha, depth_compare "<", sRGB canvas); the reference's own tests draw over
 * gfx.Background(None, BackgroundMaterial(bottom, top)) (tests/conftest.py:17-22).  DEVICE pointers;
 * depth / flags / inout_depth may be NULL (NULL flags: every pixel is a fragment).  Parity unpinned. */
typedef struct svr_compose_params {
    float   bg_bottom[4];
    float   bg_top[4];
    int32_t srgb_encode;           /* 1: encode rgb with the sRGB OETF before quantising */
} svr_compose_params;
int  svr_compose(svr_ctx* ctx, const float* rgba, const float* depth, const uint8_t* flags, int width, int height,
                 const svr_compose_params* params, uint8_t* out_rgba8, float* inout_depth, void* stream);

/* ---- sync */
int  svr_sync(svr_ctx* ctx);                 /* both streams idle */
int  svr_sync_uploads(svr_ctx* ctx);         /* upload stream idle */
/* raw device pointers of one LOD's ring textures (for diagnostics / RCCL) */
int  svr_lod_device_ptrs(svr_ctx* ctx, int lod, void** density, void** labels);
/* the micro-block copy of the LOD's density ring (svr_lod_desc::blocked_twin), or NULL: tests compare it with the ring */
int  svr_lod_twin_ptr(svr_ctx* ctx, int lod, void** twin);

/* diagnostics: batch census accumulated by instrumented renders (outputs.steps != NULL):
 * [0] general batches, [1] direct fast batches, [2] brick batches, [3] brick slabs, [4] runs,
 * [5] all-zero batches, [6] waves, [7] batches skipped as empty space; batches are per wave, 8 iterations each */
int  svr_debug_counters(svr_ctx* ctx, uint32_t out[8], int reset);
/* diagnostics: shader-clock cycles of wave residency per kernel section, summed over the waves of the
 * instrumented renders: [0] prologue (ray set-up, event search), [1] span refresh + run length, [2] general
 * batches, [3] brick slab set-up + load issue, [4] wait for the brick loads, [5] brick batches, [6] direct
 * batches, [7] epilogue (shading, stores); [8..14] finer splits used while tuning (see march_kernel.hip);
 * [15] is a COUNT, not cycles: the direct batches that gathered from a micro-block copy (svr_lod_desc::blocked_twin) */
int  svr_debug_timers(svr_ctx* ctx, uint64_t out[16], int reset);

/* timing helper: run `iters` back-to-back renders on the context's render
 * stream bracketed by HIP events on that same stream; returns the average
 * kernel time in milliseconds (used by bench.py's roofline block). */
int  svr_time_render(svr_ctx* ctx, const svr_camera* cam, const svr_frame* frame,
                     const svr_outputs* out, int iters, float* avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* SVR_H */
