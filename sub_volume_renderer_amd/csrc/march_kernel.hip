// LMIP ray-march for gfx950 (MI355X): vs_main + fs_main + raycast + sample_vol +
// hsv_selection of the reference's WGSL (src/sub_volume/shaders/*.wgsl) as ONE
// HIP kernel.  One lane = one pixel = one fs_main invocation; one wave64 = an
// 8x8 pixel tile, so the 64 rays of a wave traverse neighbouring voxels.
//
// Arithmetic contract (must stay bit-identical with oracle/lmip_oracle.c):
// strict IEEE f32, NO fp contraction (-ffp-contract=off), expression order as
// written in the WGSL; see DESIGN.md "operation-order contract".
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "svr_internal.h"

namespace {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f4 mat_vec(const float* m, float x, float y, float z, float w) {
    f4 r;
    r.x = ((m[0] * x + m[4] * y) + m[8]  * z) + m[12] * w;
    r.y = ((m[1] * x + m[5] * y) + m[9]  * z) + m[13] * w;
    r.z = ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w;
    r.w = ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w;
    return r;
}

__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// Ring slot of an in-bounds voxel.  t = ic - off is in [0, shape), shape <= ring
// and wrap0 in [0, ring), so t + wrap0 is in [0, 2*ring): one conditional
// subtraction, done as an unsigned min (exact integer modulo; sample_vol.wgsl:22).
__device__ __forceinline__ uint32_t wrap(uint32_t t, uint32_t wrap0, uint32_t ring) {
    uint32_t w = t + wrap0;
    return min(w, w - ring);
}

// try_sample_scale_i addressing (sample_vol.wgsl:4-25): returns true and the
// texel index if the voxel under data coord d lies in LOD L's ROI.
__device__ __forceinline__ bool lod_texel(const LodParams& L, float dx, float dy, float dz, size_t& idx) {
    float sx = dx * L.scale[0], sy = dy * L.scale[1], sz = dz * L.scale[2];   // :7-8
    int ix = (int)sx, iy = (int)sy, iz = (int)sz;                             // vec3<i32>(): trunc
    uint32_t tx = (uint32_t)(ix - L.off[0]);
    uint32_t ty = (uint32_t)(iy - L.off[1]);
    uint32_t tz = (uint32_t)(iz - L.off[2]);
    // :17  offset <= ic && ic < offset + shape   (one unsigned compare per axis)
    if (!(tx < L.shape[0] && ty < L.shape[1] && tz < L.shape[2])) return false;
    uint32_t wx = wrap(tx, L.wrap0[0], L.ring[0]);
    uint32_t wy = wrap(ty, L.wrap0[1], L.ring[1]);
    uint32_t wz = wrap(tz, L.wrap0[2], L.ring[2]);
    idx = (size_t)(wz * L.ring[1] + wy) * (size_t)L.ring[0] + (size_t)wx;
    return true;
}

// sample_vol (sample_vol.wgsl:51-63,80-86): first LOD whose ROI holds the voxel wins.
template <int NL>
__device__ __forceinline__ float sample_density(const MarchParams& P, float cx, float cy, float cz) {
    float dx = cx * P.size[0], dy = cy * P.size[1], dz = cz * P.size[2];      // sample_vol.wgsl:6
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        size_t idx;
        if (lod_texel(P.lod[l], dx, dy, dz, idx))
            return P.density_esh == 0 ? (float)static_cast<const uint8_t*>(P.lod[l].density)[idx]
                 : P.density_esh == 1 ? (float)static_cast<const uint16_t*>(P.lod[l].density)[idx]
                                      : static_cast<const float*>(P.lod[l].density)[idx];
    }
    return 0.0f;
}

template <int NL>
__device__ __forceinline__ uint32_t sample_label(const MarchParams& P, float cx, float cy, float cz) {
    float dx = cx * P.size[0], dy = cy * P.size[1], dz = cz * P.size[2];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        size_t idx;
        if (lod_texel(P.lod[l], dx, dy, dz, idx)) return P.lod[l].labels ? P.lod[l].labels[idx] : 0u;   // no label rings: 0
    }
    return 0u;
}

// pick_pack(u32(c * 16383.0), 14): WGSL's u32() of an f32 clamps to the u32 range (NaN -> 0), pick_pack
// clips to the field width (pygfx std.wgsl, restated)
__device__ __forceinline__ uint32_t pick_field(float c) {
    const float f = c * 16383.0f;
    uint32_t u = 0u;
    if (f > 0.0f) u = f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f;
    return min(u, 16383u);
}

// hsv_selection.wgsl:7-41
__device__ __forceinline__ f3 hsv_to_rgb(float h, float s, float v) {
    f3 r;
    if (s == 0.0f) { r.x = v; r.y = v; r.z = v; return r; }
    float h_scaled = h * 6.0f;
    float fl = floorf(h_scaled);
    int sector = (int)fl;
    float fr = h_scaled - fl;
    float p = v * (1.0f - s);
    float q = v * (1.0f - s * fr);
    float t = v * (1.0f - s * (1.0f - fr));
    if (sector == 0)      { r.x = v; r.y = t; r.z = p; }
    else if (sector == 1) { r.x = q; r.y = v; r.z = p; }
    else if (sector == 2) { r.x = p; r.y = v; r.z = t; }
    else if (sector == 3) { r.x = p; r.y = q; r.z = v; }
    else if (sector == 4) { r.x = t; r.y = p; r.z = v; }
    else                  { r.x = v; r.y = p; r.z = q; }
    return r;
}

// pygfx std.wgsl srgb2physical (restated; see oracle header)
__device__ __forceinline__ float srgb2physical(float c) {
    float f = powf((c + 0.055f) / 1.055f, 2.4f);
    float t = c / 12.92f;
    return (c <= 0.04045f) ? t : f;
}

// The kernel's own argument block through a pointer the optimiser cannot see through: uniforms read this
// way are loaded where they are used and die there, instead of occupying SGPRs for the whole march loop
// (the SGPR file is what limits this kernel, not the VGPRs).  The pointer stays in the constant address
// space, so a read at a wave-uniform index is one scalar load.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const MarchParams __attribute__((address_space(4)))* kparams_t;
__device__ __forceinline__ kparams_t fresh_params(const MarchParams&) {
    kparams_t p = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#else
typedef const MarchParams* kparams_t;
__device__ inline kparams_t fresh_params(const MarchParams& P) { return &P; }
#endif

struct Ray {
    f3 start, step;     // normalised coords (fs_main.wgsl:47-48)
    int nsteps;
};

// vs_main.wgsl:36-47 + fs_main.wgsl:20-48 for the pixel (i, j) of the full frame.
// Returns false when no fragment runs for this pixel (discard).
__device__ __forceinline__ bool setup_ray(const MarchParams& P, int i, int j, Ray& R) {
    const float W = (float)P.frame.frame_w, H = (float)P.frame.frame_h;
    float px = (2.0f * ((float)i + 0.5f)) / W - 1.0f;
    float py = 1.0f - (2.0f * ((float)j + 0.5f)) / H;
    f4 n4 = mat_vec(P.ndc_to_data, px, py, -1.0f, 1.0f);
    f4 f4_ = mat_vec(P.ndc_to_data, px, py, 1.0f, 1.0f);
    f3 far_pos  = { f4_.x / f4_.w, f4_.y / f4_.w, f4_.z / f4_.w };            // fs_main.wgsl:24
    f3 near_pos = { n4.x / n4.w, n4.y / n4.w, n4.z / n4.w };                  // :25
    f3 dir = { far_pos.x - near_pos.x, far_pos.y - near_pos.y, far_pos.z - near_pos.z };
    float len = sqrtf(dot3(dir, dir));
    f3 ray = { dir.x / len, dir.y / len, dir.z / len };                       // :28 normalize

    // back_pos: exit of the ray from the proxy box [-0.5, size-0.5]^3
    const float lo = -0.5f;
    float hx = P.size[0] - 0.5f, hy = P.size[1] - 0.5f, hz = P.size[2] - 0.5f;
    float tx1 = (lo - near_pos.x) / ray.x, tx2 = (hx - near_pos.x) / ray.x;
    float ty1 = (lo - near_pos.y) / ray.y, ty2 = (hy - near_pos.y) / ray.y;
    float tz1 = (lo - near_pos.z) / ray.z, tz2 = (hz - near_pos.z) / ray.z;
    float t_exit  = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    float t_enter = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    if (!(t_enter <= t_exit)) return false;
    f3 back = { near_pos.x + ray.x * t_exit, near_pos.y + ray.y * t_exit, near_pos.z + ray.z * t_exit };
    f4 bw = mat_vec(P.world, back.x, back.y, back.z, 1.0f);
    f4 bc = mat_vec(P.pc, bw.x, bw.y, bw.z, bw.w);
    if (!(bc.w > 0.0f) || !(bc.z >= 0.0f) || !(bc.z <= bc.w)) return false;   // outside clip volume
    // pygfx.clipping_planes.wgsl (fs_main.wgsl:8): the fragment's varyings.world_pos is the world position
    // of the back face (vs_main.wgsl:27); discard when it lies behind ANY / ALL of the material's planes
    if (P.clip_count) {
        const kparams_t Pc = fresh_params(P);
        const uint32_t nplanes = Pc->clip_count;
        const bool all = Pc->clip_all != 0;
        bool clipped = all;
        for (uint32_t k = 0; k < nplanes; ++k) {
            const bool behind = ((bw.x * Pc->clip[k][0] + bw.y * Pc->clip[k][1]) + bw.z * Pc->clip[k][2]) < Pc->clip[k][3];
            clipped = all ? (clipped && behind) : (clipped || behind);
        }
        if (clipped) return false;
    }

    f3 nb = { near_pos.x - back.x, near_pos.y - back.y, near_pos.z - back.z };
    float dist = dot3(nb, ray);                                               // :32
    dist = fmaxf(dist, fminf((-0.5f - back.x) / ray.x, (P.size[0] - 0.5f - back.x) / ray.x));
    dist = fmaxf(dist, fminf((-0.5f - back.y) / ray.y, (P.size[1] - 0.5f - back.y) / ray.y));
    dist = fmaxf(dist, fminf((-0.5f - back.z) / ray.z, (P.size[2] - 0.5f - back.z) / ray.z));
    f3 front = { back.x + ray.x * dist, back.y + ray.y * dist, back.z + ray.z * dist };   // :39
    float nf = -dist / P.rel_step + 0.5f;                                     // :43
    if (!(nf >= 1.0f)) return false;                                          // :44 discard
    if (nf > 16777216.0f) nf = 16777216.0f;
    R.nsteps = (int)nf;
    float nstepsf = (float)R.nsteps;
    R.start = { (front.x + 0.5f) / P.size[0], (front.y + 0.5f) / P.size[1], (front.z + 0.5f) / P.size[2] };
    R.step = { ((back.x - front.x) / P.size[0]) / nstepsf,
               ((back.y - front.y) / P.size[1]) / nstepsf,
               ((back.z - front.z) / P.size[2]) / nstepsf };
    return true;
}

struct Hit {
    bool found;
    float sample;
    f3 offset, coord;
    uint32_t steps;
};

// fs_main.wgsl:56-98: fragment outputs from the raycast result.
template <int NL>
__device__ __forceinline__ void shade_and_store(const MarchParams& P, size_t o, bool has_fragment, const Hit& h) {
    float4 color = make_float4(0.f, 0.f, 0.f, 0.f);
    float depth = 0.f; uint32_t label = 0u; uint8_t cls = SVR_PIX_DISCARD;
    if (has_fragment) {
        if (!h.found) {                         // fs_main.wgsl:93-98
            color = make_float4(0.f, 0.f, 0.f, 1.f); cls = SVR_PIX_MISS;
        } else {
            float v = (h.sample - P.clim0) / (P.clim1 - P.clim0);        // sampled_value_to_color
            if (P.gamma != 1.0f) v = powf(v, P.gamma);                    // pow(v, 1) is v: the default gamma costs nothing
            float phys = P.colorspace_srgb ? srgb2physical(v) : v;        // raycast.wgsl:71-75
            label = sample_label<NL>(P, h.coord.x, h.coord.y, h.coord.z); // raycast.wgsl:81
            f4 wp = mat_vec(P.world, h.coord.x - 0.5f, h.coord.y - 0.5f, h.coord.z - 0.5f, 1.0f);
            f4 ndc = mat_vec(P.pc, wp.x, wp.y, wp.z, wp.w);
            depth = ndc.z / fmaxf(ndc.w, 0.001f);                         // fs_main.wgsl:72
            const float* hs = P.colors + 4u * (label % P.color_count);    // hsv_selection.wgsl:1-3
            f3 rgb = hsv_to_rgb(hs[0], hs[1], phys);
            float distance = sqrtf(dot3(h.offset, h.offset));             // fs_main.wgsl:82
            float fog = expf(-P.fog_density * distance);                  // :83
            float omf = 1.0f - fog;
            color.x = P.fog_color[0] * omf + rgb.x * fog;                 // :84 mix
            color.y = P.fog_color[1] * omf + rgb.y * fog;
            color.z = P.fog_color[2] * omf + rgb.z * fog;
            color.w = P.opacity;                                          // :86
            cls = SVR_PIX_HIT;
        }
    }
    reinterpret_cast<float4*>(P.rgba)[o] = color;
    if (P.depth) P.depth[o] = depth;
    if (P.label) P.label[o] = label;
    if (P.flags) P.flags[o] = cls;
    if (P.pick) {                                                         // fs_main.wgsl:89-92 (write_pick)
        unsigned long long pk = 0ull;
        if (cls == SVR_PIX_HIT)
            pk = (unsigned long long)min(P.pick_id, 0xFFFFFu) |
                 ((unsigned long long)pick_field(h.coord.x) << 20) | ((unsigned long long)pick_field(h.coord.y) << 34) |
                 ((unsigned long long)pick_field(h.coord.z) << 48);
        P.pick[o] = pk;
    }
}

// Block -> 16x16 pixel tile when no placement table is given (MarchParams::tile_order, built by the
// host: chunks of tiles dealt to the XCDs, see svr_api.hip).  Workgroups are dealt round-robin to the
// 8 XCDs (blockIdx b and b+8 share an XCD and its 4 MiB L2); this fallback gives each XCD one
// contiguous run of tiles: best L2 locality, but the runs (bands of the frame) differ in cost.
// Placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
    const int per = nblocks >> 3;            // full groups of 8
    const int body = per << 3;
    if (b >= body) return b;                 // tail blocks keep their index
    return (b & 7) * per + (b >> 3);
}

// ---------------------------------------------------------------------------
// Variant 1: straightforward march.  raycast.wgsl:29-62 verbatim, one texel
// fetch per step from global memory.
// ---------------------------------------------------------------------------
template <int NL, bool COUNT>
__global__ __launch_bounds__(256) void march_simple(const MarchParams P) {
    const int nblocks = P.tiles_x * P.tiles_y;
    const int t = P.tile_order ? (int)P.tile_order[blockIdx.x] : xcd_remap((int)blockIdx.x, nblocks);
    const int tile_x = t % P.tiles_x, tile_y = t / P.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // 4 waves = 2x2 sub-tiles of 8x8 pixels
    const int c = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    const int r = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (c >= P.frame.out_w || r >= P.frame.out_h) return;
    const size_t o = (size_t)r * (size_t)P.frame.out_w + (size_t)c;
    const int x = P.frame.x0 + c;
    const int y = P.frame.y0 + (r / P.frame.band_h) * P.frame.band_pitch + (r % P.frame.band_h);

    Ray R; Hit h;
    h.found = false; h.sample = 0.f; h.steps = 0u;
    h.offset = { 0.f, 0.f, 0.f }; h.coord = { 0.f, 0.f, 0.f };
    bool frag = (x < P.frame.frame_w && y < P.frame.frame_h) && setup_ray(P, x, y, R);
    if (frag) {
        float local_max = 0.f;
        int since = 0;
        const float nstepsf = (float)R.nsteps;
        for (float iter = 0.0f; iter < nstepsf; iter = iter + 1.0f) {                // raycast.wgsl:29
            if (COUNT) ++h.steps;
            f3 off = { iter * R.step.x, iter * R.step.y, iter * R.step.z };            // :30
            f3 coord = { R.start.x + off.x, R.start.y + off.y, R.start.z + off.z };    // :31
            float s = sample_density<NL>(P, coord.x, coord.y, coord.z);                // :32
            float inten = fabsf(s);                                                    // :33
            if (!h.found) {
                if (inten >= P.lmip_threshold) {                                       // :37-44
                    h.found = true; local_max = inten; h.sample = s; h.offset = off; h.coord = coord; since = 0;
                }
            } else {
                since += 1;                                                            // :47
                if (inten > local_max) { local_max = inten; h.sample = s; h.offset = off; h.coord = coord; }
                if (since >= P.lmip_max_samples || inten < local_max * P.lmip_fall_off) break;   // :58-60
            }
        }
    }
    shade_and_store<NL>(P, o, frag, h);
    if (COUNT && P.steps) P.steps[o] = h.steps;
}

// ---------------------------------------------------------------------------
// SVR_MODE_WEIGHTED_AVERAGE (include/svr.h; FUTURE.md:97-109 names the mode, the formula is this project's):
// every sample weighs w = max(1 - k d, 0)^2 with d its distance from the ray's entry (the unit of the fog
// distance, fs_main.wgsl:82); the pixel shows sum(w s) / sum(w) at the sample with the largest w |s|.
// Same ray, same sample positions and the same LOD fall-through as raycast.wgsl:29-32; sums in sample order, so
// the CPU twin (oracle/lmip_oracle.c raycast_weighted_average) reproduces every bit.  One texel fetch per
// step from global memory, like march_simple: the fallback where the span march (which carries this mode as a second
// batch reducer, see lmip_batch) cannot address the rings.
// ---------------------------------------------------------------------------
template <int NL, bool COUNT>
__global__ __launch_bounds__(256) void march_wavg(const MarchParams P) {
    const int nblocks = P.tiles_x * P.tiles_y;
    const int t = P.tile_order ? (int)P.tile_order[blockIdx.x] : xcd_remap((int)blockIdx.x, nblocks);
    const int tile_x = t % P.tiles_x, tile_y = t / P.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    const int r = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (c >= P.frame.out_w || r >= P.frame.out_h) return;
    const size_t o = (size_t)r * (size_t)P.frame.out_w + (size_t)c;
    const int x = P.frame.x0 + c;
    const int y = P.frame.y0 + (r / P.frame.band_h) * P.frame.band_pitch + (r % P.frame.band_h);

    Ray R; Hit h;
    h.found = false; h.sample = 0.f; h.steps = 0u;
    h.offset = { 0.f, 0.f, 0.f }; h.coord = { 0.f, 0.f, 0.f };
    bool frag = (x < P.frame.frame_w && y < P.frame.frame_h) && setup_ray(P, x, y, R);
    if (frag) {
        const float steplen = sqrtf(dot3(R.step, R.step));
        float num = 0.f, den = 0.f, best = 0.f;
        const float nstepsf = (float)R.nsteps;
        for (float iter = 0.0f; iter < nstepsf; iter = iter + 1.0f) {
            const float tw = 1.0f - P.weight_falloff * (iter * steplen);
            if (!(tw > 0.0f)) break;                                                   // this sample and all later ones weigh nothing
            if (COUNT) ++h.steps;
            f3 off = { iter * R.step.x, iter * R.step.y, iter * R.step.z };            // raycast.wgsl:30
            f3 coord = { R.start.x + off.x, R.start.y + off.y, R.start.z + off.z };    // :31
            const float s = sample_density<NL>(P, coord.x, coord.y, coord.z);          // :32
            const float w = tw * tw;
            num = num + w * s;
            den = den + w;
            const float contribution = w * fabsf(s);
            if (contribution > best) { best = contribution; h.offset = off; h.coord = coord; }
        }
        h.found = best > 0.0f;
        h.sample = h.found ? num / den : 0.0f;
    }
    shade_and_store<NL>(P, o, frag, h);
    if (COUNT && P.steps) P.steps[o] = h.steps;
}

// ---------------------------------------------------------------------------
// Variant 0 (default): span march.
//
// Exactness argument.  For one ray and one axis, the voxel index the reference
// computes at iteration i,
//     ic(i) = i32( ((start + f32(i)*step) * size) * scale )        (raycast.wgsl:30-31,
//                                                                   sample_vol.wgsl:6-8,17)
// is a composition of monotone functions of i (IEEE rounding is monotone), so it is
// monotone in i.  Therefore, per LOD l and axis:
//   * "ic lies in the ROI" holds on ONE contiguous iteration interval; intersecting the
//     three axes gives [A_l, B_l), and the cascade of sample_vol.wgsl:51-63 is
//     "smallest l with A_l <= i < B_l";
//   * the ring wrap  slot = (ic + addw) mod ring  (sample_vol.wgsl:22) changes its
//     constant at most once, at an iteration C_{l,axis}.
// All these iterations are found EXACTLY by evaluating that same f32 chain
// (first_cross).  Between two consecutive events a lane's texel address is
//     ((iz*Ry + iy)*Rx + ix)*4 + Kc        with a per-lane constant Kc,
// so the hot loop has no bounds tests, no cascade and no modulo; it computes U
// offsets, issues U independent buffer loads (hardware range check returns 0 where
// no LOD holds the voxel) and only then runs the sequential LMIP state machine —
// skipped entirely while no lane has reached the threshold.  Batches that contain an
// event for some lane take the general path, which evaluates every sample exactly.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int axis_voxel(int i, float start, float step, float size, float scale) {
    float off = (float)i * step;          // raycast.wgsl:30  (iter is an exact integer-valued f32)
    float coord = start + off;            // :31
    float d = coord * size;               // sample_vol.wgsl:6
    float sd = d * scale;                 // :7-8
    return (int)sd;                       // :17 vec3<i32>()
}

// First iteration j in [0, n] at which the (monotone) voxel index has crossed `thresh`
// in its direction of travel: ic(j) >= thresh for step > 0, ic(j) < thresh for step < 0.
// A zero step never crosses: returns 0 if the condition already holds, else n.  The result is only defined for lanes
// that pass wanted = true.
__device__ __forceinline__ int first_cross(int n, float start, float step, float size, float scale, int thresh, bool wanted = true) {
    const bool inc = step > 0.0f;
    auto pred = [&](int j) {
        const int v = axis_voxel(j, start, step, size, scale);
        return inc ? v >= thresh : v < thresh;
    };
    if (!(step > 0.0f) && !(step < 0.0f)) return pred(0) ? 0 : n;
    // only a starting point for the exact monotone fix-up below: fast reciprocals are fine
    // (v_rcp_f32, 1 ulp: __frcp_rn is a correctly rounded 1 / x, i.e. a full 11-instruction division)
    const float guess = ceilf(((float)thresh * __builtin_amdgcn_rcpf(size * scale) - start) * __builtin_amdgcn_rcpf(step));
    int j = (int)fminf(fmaxf(guess, 0.0f), (float)n);
    // The guess is nearly always the answer itself (the f32 chain moves it by one iteration at most): j is the first
    // crossing iff the index has not crossed at j - 1 and has at j.  Two evaluations settle that; the search loops —
    // which the compiler turns into 8 evaluations per trip, ~60 instructions even for a trip that finds nothing to
    // do — only run when some lane's guess was off.
    // (`wanted`: the caller only uses this lane's result then; the other lanes of the wave neither trigger nor walk the
    // loops, whatever their guess was)
    const bool settled = (j == 0 || !pred(j - 1)) && (j == n || pred(j));
    if (__builtin_amdgcn_ballot_w64(wanted && !settled) != 0) {
#pragma nounroll
        while (wanted && j > 0 && pred(j - 1)) --j;
#pragma nounroll
        while (wanted && j < n && !pred(j)) ++j;
    }
    return j;
}

// Byte offset (inside MarchParams::density_all) of the texel under data coord d for a
// voxel KNOWN to lie in LOD L's ROI; general form with the explicit ring wrap.
template <int ESH, typename LodT>
__device__ __forceinline__ uint32_t lod_offset_wrapped(const LodT& L, float dx, float dy, float dz) {
    float sx = dx * L.scale[0], sy = dy * L.scale[1], sz = dz * L.scale[2];
    uint32_t wx = (uint32_t)((int)sx + L.addw[0]);
    uint32_t wy = (uint32_t)((int)sy + L.addw[1]);
    uint32_t wz = (uint32_t)((int)sz + L.addw[2]);
    wx = min(wx, wx - L.ring[0]);
    wy = min(wy, wy - L.ring[1]);
    wz = min(wz, wz - L.ring[2]);
    return __umul24(__umul24(wz, L.ring[1]) + wy, L.rx4) + L.base_bytes + (wx << ESH);
}

// The same for a ring that is reached through several buffer resources (4 GiB or more: parts of LodParams::zsplit
// planes): the offset is relative to the part that holds the slot's z plane, `part` says which
template <int ESH, typename LodT>
__device__ __forceinline__ uint32_t lod_offset_wrapped_split(const LodT& L, float dx, float dy, float dz, uint32_t& part) {
    float sx = dx * L.scale[0], sy = dy * L.scale[1], sz = dz * L.scale[2];
    uint32_t wx = (uint32_t)((int)sx + L.addw[0]);
    uint32_t wy = (uint32_t)((int)sy + L.addw[1]);
    uint32_t wz = (uint32_t)((int)sz + L.addw[2]);
    wx = min(wx, wx - L.ring[0]);
    wy = min(wy, wy - L.ring[1]);
    wz = min(wz, wz - L.ring[2]);
    part = 0u;
    for (uint32_t k = 1; k < L.nparts; ++k) part += wz >= k * L.zsplit ? 1u : 0u;         // nparts <= 8, wave-uniform
    wz -= part * L.zsplit;
    return __umul24(__umul24(wz, L.ring[1]) + wy, L.rx4) + L.base_bytes + (wx << ESH);
}

// the buffer resource of part p (wave-uniform) of such a ring — or of its micro-block copy, which is cut the same way
template <typename LodT>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t part_rsrc(const LodT& L, uint32_t p, const void* ring_base) {
    const char* base = static_cast<const char*>(ring_base) + (size_t)p * (size_t)L.part_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(p + 1u == L.nparts ? L.rbytes_last : L.rbytes), 0x00020000);
}
template <typename LodT>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t part_rsrc(const LodT& L, uint32_t p) { return part_rsrc(L, p, L.rbase); }

// Raw texel as fetched: f32 rings (ESH = 2) hold the sample itself, u8 rings (ESH = 0) the byte,
// widened to f32 (exactly) only where the LMIP state machine needs the value.
template <int ESH> struct Texel { typedef float type; };
template <> struct Texel<0> { typedef uint32_t type; };
template <> struct Texel<1> { typedef uint32_t type; };
__device__ __forceinline__ float texel_value(float t) { return t; }
__device__ __forceinline__ float texel_value(uint32_t t) { return (float)t; }
__device__ __forceinline__ float texel_abs(float t) { return fabsf(t); }
__device__ __forceinline__ uint32_t texel_abs(uint32_t t) { return t; }
__device__ __forceinline__ float texel_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ uint32_t texel_max(uint32_t a, uint32_t b) { return max(a, b); }

// texel read from the LDS brick image at an ABSOLUTE LDS byte address (the base of the dynamic LDS block is folded into
// the wave-uniform part of the address: as `lds_all + a` every read carried a VALU add of the block's address)
template <int ESH>
__device__ __forceinline__ typename Texel<ESH>::type lds_texel(uint32_t a) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) const uint8_t* p8_t;
    typedef __attribute__((address_space(3))) const uint16_t* p16_t;
    typedef __attribute__((address_space(3))) const float* p32_t;
    if constexpr (ESH == 2) return *(p32_t)a;
    else if constexpr (ESH == 1) return (uint32_t)*(p16_t)a;
    else return (uint32_t)*(p8_t)a;
#else
    (void)a;
    return typename Texel<ESH>::type();       // device code only
#endif
}

// texel fetch through the range-checked buffer resource
// (soff: a wave-uniform byte offset that rides in the instruction's scalar-offset operand, for free)
template <int ESH>
__device__ __forceinline__ typename Texel<ESH>::type fetch_density(__amdgpu_buffer_rsrc_t rsrc, uint32_t off, uint32_t soff = 0u) {
    if constexpr (ESH == 2) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, (int)soff, 0));
    else if constexpr (ESH == 1) return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)off, (int)soff, 0);
    else return (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rsrc, (int)off, (int)soff, 0);
}

// a*b + c on the 24-bit integer multiplier (one full-rate-class VALU op); a, b < 2^24
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <int SH>
__device__ __forceinline__ uint32_t shl_add_c(uint32_t a, uint32_t c) {                // (a << SH) + c
    if (SH == 0) return a + c;
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(c));
    return r;
}

// Micro-block copy of a ring (svr_lod_desc::blocked_twin; host: svr_blocked_index): 128-byte blocks of
// 2^XB x 2^YB x 2^ZB slots in [bz][by][bx] order, the slots of a block in [z][y][x] order.
template <int ESH> struct TwinBlock { static constexpr int XB = ESH == 0 ? 3 : 2, YB = 2, ZB = ESH == 2 ? 1 : 2; };
// Byte offset of the texel of voxel (x, y, z) in that copy, for a lane whose ring slot is voxel + (kx, ky, kz) with
// constants that are multiples of the ring extents — and therefore of the block extents: the block coordinates of the
// slot are those of the voxel plus constants, the position inside the block is the voxel's own low bits:
//   offset = ((z >> ZB) * NBy + (y >> YB)) * NBx + (x >> XB) + Kb) * 128 + in-block part,   Kb = the constants' block number
// (all mod 2^32).  Computed from the packed form of the voxel (x, yz = y | z << 16, both below 2^16: LodParams::twin says
// so) in 9 VALU operations, where the form above takes 12 and a quarter-rate v_mad_u64_u32 (the compiler's choice for the
// inner multiply-add; the asm mad24 of the row-major form costs this loop 9 more VGPRs — one wave per SIMD less):
//   offset = 4row * (NBx * 32) + (x & ~xm) << (7 - XB)  +  [ (y & 3) * wy + (z & zm) * wz + ((x & xm) << ESH | KbS) ]
// with 4row = (y & ~3) + (z & ~zm) * (NBy * 4 >> ZB) from one v_dot2_u32_u16, the bracket from another (wy, wz: the byte
// strides of y and z inside a block) and KbS = Kb << 7 (its low 7 bits are free for the in-block part).
struct TwinConsts { uint32_t row_w, inb_w, nbx32; };           // wave-uniform: (1 | NBy * 4 >> ZB << 16), (wy | wz << 16), NBx * 32
template <int ESH>
__device__ __forceinline__ TwinConsts twin_consts(uint32_t Rx, uint32_t Ry) {
    typedef TwinBlock<ESH> B;
    TwinConsts c;
    c.row_w = 1u | (((Ry >> B::YB) * (4u >> B::ZB)) << 16);
    c.inb_w = (1u << (B::XB + ESH)) | ((1u << (B::XB + B::YB + ESH)) << 16);
    c.nbx32 = (Rx >> B::XB) * 32u;
    return c;
}
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
template <int ESH>
__device__ __forceinline__ uint32_t twin_offset_packed(uint32_t x, uint32_t yz, const TwinConsts& c, uint32_t KbS) {
    typedef TwinBlock<ESH> B;
    constexpr uint32_t xm = (1u << B::XB) - 1u, zm = (1u << B::ZB) - 1u, low = 3u | (zm << 16);
    const uint32_t xin = ESH == 0 ? ((x & xm) | KbS) : (((x & xm) << ESH) | KbS);
    const uint32_t inb = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, yz & low), __builtin_bit_cast(ushort2_t, c.inb_w), xin, false);
    const uint32_t row4 = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2_t, yz & ~low), __builtin_bit_cast(ushort2_t, c.row_w), 0u, false);
    return __umul24(row4, c.nbx32) + (((x & ~xm) << (7 - B::XB)) + inb);
}
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// Voxel indices of TWO consecutive samples (iterations it, it+1) of one ray, with packed f32 math
// (v_pk_mul_f32 / v_pk_add_f32 are IEEE-exact per component, no fusing): per axis
//   ic = i32((start + f32(i) * step) * ss),   ss = size * scale (scale = 2^-k, see the caller)
struct Idx2 { uint32_t x0, y0, z0, x1, y1, z1; };
__device__ __forceinline__ Idx2 voxel_pair(float sx, float sy, float sz, float tx, float ty, float tz, float2_t iter,
                                           float ssx, float ssy, float ssz) {
    const float2_t cx = (iter * tx + sx) * ssx;                  // -ffp-contract=off: mul, add, mul
    const float2_t cy = (iter * ty + sy) * ssy;
    const float2_t cz = (iter * tz + sz) * ssz;
    Idx2 r;
    r.x0 = (uint32_t)(int)cx.x; r.x1 = (uint32_t)(int)cx.y;
    r.y0 = (uint32_t)(int)cy.x; r.y1 = (uint32_t)(int)cy.y;
    r.z0 = (uint32_t)(int)cz.x; r.z1 = (uint32_t)(int)cz.y;
    return r;
}

// Same chain, but y and z of each sample come back packed as y | z << 16 (the second convert writes
// the upper half of the register directly): operand of the v_dot2_u32_u16 brick address below.
// Indices are < 2^15 here (checked by the caller through the packed-i16 box reduction).
struct Idx2p { uint32_t x0, yz0, x1, yz1; };
__device__ __forceinline__ Idx2p voxel_pair_packed(float sx, float sy, float sz, float tx, float ty, float tz, float2_t iter,
                                                   float ssx, float ssy, float ssz) {
    const float2_t cx = (iter * tx + sx) * ssx;
    const float2_t cy = (iter * ty + sy) * ssy;
    const float2_t cz = (iter * tz + sz) * ssz;
    Idx2p r;
    r.x0 = (uint32_t)(int)cx.x; r.x1 = (uint32_t)(int)cx.y;
    r.yz0 = (uint32_t)(int)cy.x; r.yz1 = (uint32_t)(int)cy.y;
    asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r.yz0) : "v"(cz.x));
    asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r.yz1) : "v"(cz.y));
    return r;
}
// a.lo * b.lo + a.hi * b.hi + c on 16-bit halves (one VALU op)
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
// a register with no defined value and no instruction (lanes whose value is never looked at)
template <typename T> __device__ __forceinline__ T undefined_value() { T v; asm volatile("" : "=v"(v)); return v; }

// Six wave64 reductions at once: min of a, b, c and max of d, e, f over all lanes; every lane of row 3
// (lane 63 is read) ends up with the results.  Written out with DPP-fused VOP2 ops and the six
// independent chains interleaved, so the two wait states a DPP read needs after a VALU write of its
// source are always covered by the other chains (the compiler emits mov + nop + mov_dpp + op per step).
__device__ __forceinline__ void wave_min3_max3(int& a, int& b, int& c, int& d, int& e, int& f) {
#define SVR_STEP6(ctrl)                              \
    "v_min_i32_dpp %0, %0, %0 " ctrl "\n"            \
    "v_min_i32_dpp %1, %1, %1 " ctrl "\n"            \
    "v_min_i32_dpp %2, %2, %2 " ctrl "\n"            \
    "v_max_i32_dpp %3, %3, %3 " ctrl "\n"            \
    "v_max_i32_dpp %4, %4, %4 " ctrl "\n"            \
    "v_max_i32_dpp %5, %5, %5 " ctrl "\n"
    asm volatile("s_nop 1\n"
                 SVR_STEP6("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 SVR_STEP6("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 SVR_STEP6("row_half_mirror row_mask:0xf bank_mask:0xf")
                 SVR_STEP6("row_mirror row_mask:0xf bank_mask:0xf")
                 SVR_STEP6("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 SVR_STEP6("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef SVR_STEP6
    a = __builtin_amdgcn_readlane(a, 63); b = __builtin_amdgcn_readlane(b, 63); c = __builtin_amdgcn_readlane(c, 63);
    d = __builtin_amdgcn_readlane(d, 63); e = __builtin_amdgcn_readlane(e, 63); f = __builtin_amdgcn_readlane(f, 63);
}

// wave64 min / max of an int via DPP (no LDS traffic); result is wave-uniform.  DPP-fused VOP2 ops
// (a DPP read needs two wait states after the VALU write of its source)
template <bool MAX>
__device__ __forceinline__ int wave_reduce(int v) {
#define SVR_STEP1(ctrl) "s_nop 1\n" "v_min_i32_dpp %0, %0, %0 " ctrl "\n"
#define SVR_STEP1X(ctrl) "s_nop 1\n" "v_max_i32_dpp %0, %0, %0 " ctrl "\n"
    if (MAX)
        asm volatile(SVR_STEP1X("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                     SVR_STEP1X("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                     SVR_STEP1X("row_half_mirror row_mask:0xf bank_mask:0xf")
                     SVR_STEP1X("row_mirror row_mask:0xf bank_mask:0xf")
                     SVR_STEP1X("row_bcast:15 row_mask:0xa bank_mask:0xf")
                     SVR_STEP1X("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     "s_nop 0" : "+v"(v));
    else
        asm volatile(SVR_STEP1("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                     SVR_STEP1("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                     SVR_STEP1("row_half_mirror row_mask:0xf bank_mask:0xf")
                     SVR_STEP1("row_mirror row_mask:0xf bank_mask:0xf")
                     SVR_STEP1("row_bcast:15 row_mask:0xa bank_mask:0xf")
                     SVR_STEP1("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     "s_nop 0" : "+v"(v));
#undef SVR_STEP1
#undef SVR_STEP1X
    return __builtin_amdgcn_readlane(v, 63);
}

// The uniforms of one LOD that the march loop needs (MarchParams::lod[l], read through fresh_params)
struct LodK {
    uint32_t ring[3];
    float    scale[3];
    int32_t  addw[3];
    uint32_t rx4, base_bytes;
    float    ss[3];
    int32_t  slab;
    const void* rbase;
    uint32_t rbytes;
    uint32_t nparts, zsplit, part_bytes, rbytes_last;      // rings of 4 GiB or more: parts of zsplit planes (LodParams)
    uint32_t twin;                                         // micro-block copy of the ring (LodParams::twin; its resource is read where
                                                           // a wave turns to it, not kept live across the brick loop)
};
__device__ __forceinline__ LodK load_lod(kparams_t p, int l) {
    LodK k;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        k.ring[a] = p->lod[l].ring[a]; k.scale[a] = p->lod[l].scale[a]; k.addw[a] = p->lod[l].addw[a]; k.ss[a] = p->lod[l].ss[a];
    }
    k.rx4 = p->lod[l].rx4; k.base_bytes = p->lod[l].base_bytes; k.slab = p->lod[l].slab;
    k.rbase = p->lod[l].rbase; k.rbytes = p->lod[l].rbytes;
    k.nparts = p->lod[l].nparts; k.zsplit = p->lod[l].zsplit; k.part_bytes = p->lod[l].part_bytes; k.rbytes_last = p->lod[l].rbytes_last;
    k.twin = p->lod[l].twin;
    return k;
}

// Per-ray event table of one LOD
struct LodEvents {
    int a, b;            // ROI interval [a, b)
    int cx, cy, cz;      // wrap-constant change iterations per axis
};

// LDS bricks: one private region of MarchParams::brick_bytes per wave (dynamic LDS, see launch_nl)

template <int NL, int U, bool COUNT, int ESH, bool BIG, bool WAVG = false>
__global__ __launch_bounds__(256) void march_span(const MarchParams P) {
    // dynamic LDS: kBrickBytes per wave of the block (u8 rings), see launch_nl
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    const int nblocks = P.tiles_x * P.tiles_y;
    const int tb = P.tile_order ? (int)P.tile_order[blockIdx.x] : xcd_remap((int)blockIdx.x, nblocks);
    const int tile_x = tb % P.tiles_x, tile_y = tb / P.tiles_x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // wave tile = (1 << lw) x (64 >> lw) pixels; a block is one wave tile (wl = 0: a finished wave frees
    // its slot at once) or 2 x 2 of them (wl = 1)
    const int lw = P.tile_log2w, wl = P.block_waves_log2;
    const int c0 = ((tile_x << wl) + (wave & wl)) << lw, r0 = ((tile_y << wl) + (wave >> 1)) << (6 - lw);
    // Lane order inside the wave tile.  The L1 serves a gather one lane-quad at a time and merges the
    // four lanes only when they hit the same line, so consecutive lanes should be neighbours along the
    // volume's contiguous axis x.  On screen, x points towards its vanishing point (the clip-space image
    // of (1,0,0,0)): run consecutive lanes along screen rows if that direction is mostly horizontal at
    // this tile, along screen columns otherwise.  A permutation of which lane renders which pixel only.
    int lc = lane & ((1 << lw) - 1), lr = lane >> lw;
    {
        const float pcx = (float)(P.frame.x0 + c0) + 0.5f * (float)(1 << lw);
        const float pcy = (float)(P.frame.y0 + (r0 / P.frame.band_h) * P.frame.band_pitch + (r0 % P.frame.band_h)) +
                          0.5f * (float)(64 >> lw);
        const float ncx = 2.0f * pcx / (float)P.frame.frame_w - 1.0f;
        const float ncy = 1.0f - 2.0f * pcy / (float)P.frame.frame_h;
        const float dx = (P.xdir[0] - ncx * P.xdir[3]) * (float)P.frame.frame_w;
        const float dy = (P.xdir[1] - ncy * P.xdir[3]) * (float)P.frame.frame_h;
        if (P.orient && fabsf(dy) > fabsf(dx)) { lr = lane & ((64 >> lw) - 1); lc = lane >> (6 - lw); }
    }
    const int c = c0 + lc;
    const int r = r0 + lr;
    const bool inside = c < P.frame.out_w && r < P.frame.out_h;
    const size_t o = (size_t)r * (size_t)P.frame.out_w + (size_t)c;
    const int x = P.frame.x0 + c;
    const int y = P.frame.y0 + (r / P.frame.band_h) * P.frame.band_pitch + (r % P.frame.band_h);

    // instrumented build: shader-clock cycles per section (svr_debug_timers), wave-uniform
    unsigned long long t_acc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long t_last = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
    auto lap = [&](int k) {
        if (COUNT) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_acc[k] += t - t_last;
            t_last = t;
        }
    };

    // (the ray lives in six scalar registers, assembled from selects: a struct written through a pointer under a
    // branch ends up in private memory as soon as the register allocator is under pressure)
    Ray Rs;
    Rs.nsteps = 0; Rs.start = { 0.f, 0.f, 0.f }; Rs.step = { 0.f, 0.f, 0.f };
    const bool frag = inside && (x < P.frame.frame_w && y < P.frame.frame_h) && setup_ray(P, x, y, Rs);
    const int nsteps = frag ? Rs.nsteps : 0;
    const float Rsx = frag ? Rs.start.x : 0.f, Rsy = frag ? Rs.start.y : 0.f, Rsz = frag ? Rs.start.z : 0.f;
    const float Rtx = frag ? Rs.step.x : 0.f, Rty = frag ? Rs.step.y : 0.f, Rtz = frag ? Rs.step.z : 0.f;

    // ---- exact per-LOD event iterations
    // ic is monotone along the ray, so its values at the first and last sample bound every other
    // sample: a threshold outside that range is crossed at iteration 0 or never, and the search
    // (first_cross) runs only for thresholds some lane of the wave really crosses.
    float d0[3], d1[3];                       // data coords (sample_vol.wgsl:6) of samples 0 and nsteps-1
    {
        const float lastf = (float)(nsteps - 1);
        d0[0] = (Rsx + 0.0f * Rtx) * P.size[0];  d1[0] = (Rsx + lastf * Rtx) * P.size[0];
        d0[1] = (Rsy + 0.0f * Rty) * P.size[1];  d1[1] = (Rsy + lastf * Rty) * P.size[1];
        d0[2] = (Rsz + 0.0f * Rtz) * P.size[2];  d1[2] = (Rsz + lastf * Rtz) * P.size[2];
    }
    LodEvents ev[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const LodParams& L = P.lod[l];
        int a = 0, b = nsteps, cr[3] = { 0, 0, 0 };
        if (L.shape[0] == 0 || L.shape[1] == 0 || L.shape[2] == 0) {      // ROI is None: never in bounds
            ev[l].a = 0; ev[l].b = 0; ev[l].cx = 0; ev[l].cy = 0; ev[l].cz = 0;
            continue;
        }
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const float st = ax == 0 ? Rsx : (ax == 1 ? Rsy : Rsz);
            const float sp = ax == 0 ? Rtx : (ax == 1 ? Rty : Rtz);
            const bool inc = sp > 0.0f;
            const int v0 = (int)(d0[ax] * L.scale[ax]), v1 = (int)(d1[ax] * L.scale[ax]);   // ic(0), ic(nsteps-1)
            // == first_cross(nsteps, st, sp, size, scale, T): pred(0) -> 0; !pred(nsteps-1) -> nsteps
            auto cross = [&](int T) {
                const bool p0 = inc ? v0 >= T : v0 < T;
                const bool p1 = inc ? v1 >= T : v1 < T;
                int j = p0 ? 0 : nsteps;
                const bool need = !p0 && p1;
                if (__builtin_amdgcn_ballot_w64(need) != 0) {
                    const int js = first_cross(nsteps, st, sp, P.size[ax], L.scale[ax], T, need);
                    j = need ? js : j;
                }
                return j;
            };
            const int lo = L.off[ax], hi = L.off[ax] + (int)L.shape[ax];
            const int jlo = cross(lo);
            const int jhi = cross(hi);
            // increasing: inside on [jlo, jhi); decreasing: inside on [jhi, jlo)
            int en = inc ? jlo : jhi, ex = inc ? jhi : jlo;
            if (!(sp > 0.0f) && !(sp < 0.0f)) {            // constant index: inside always or never
                en = 0; ex = (lo <= v0 && v0 < hi) ? nsteps : 0;
            }
            a = max(a, en); b = min(b, ex);
            // the slot constant changes only if the ROI really wraps on this axis (wave-uniform test);
            // otherwise encode "never": (n >= C) == inc must stay false
            if (L.wrap0[ax] + L.shape[ax] > L.ring[ax])
                cr[ax] = cross((int)L.ring[ax] - L.addw[ax]);
            else
                cr[ax] = inc ? nsteps : 0;
        }
        if (a >= b || L.shape[0] == 0) { a = 0; b = 0; }
        ev[l].a = a; ev[l].b = b; ev[l].cx = cr[0]; ev[l].cy = cr[1]; ev[l].cz = cr[2];
    }

    // One buffer resource over the allocation that holds every LOD's ring (32-bit byte offsets, hardware range
    // check) while it is below 4 GiB; BIG builds (rings of 4 GiB or more in total) take one resource per LOD instead.
    __amdgpu_buffer_rsrc_t rsrc_all = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(P.density_all), 0, (int)P.density_all_bytes, 0x00020000);

#ifdef SVR_EXPERIMENTS
    const int dbg_nowait = P.dbg_nowait, brick_pow2 = P.brick_pow2;       // timing experiments (wrong pixels) / round-1 brick layout
#else
    constexpr int dbg_nowait = 0, brick_pow2 = 0;                         // the shipped kernel carries neither
#endif
    bool found = false, finished = (dbg_nowait & 2) != 0;      // instruction-count experiments: prologue + epilogue only
    float local_max = 0.f, samp = 0.f;
    int hit_i = 0, since = 0;
    uint32_t steps = 0;

    // LMIP state machine over one batch of U samples starting at iteration nb (raycast.wgsl:35-61).
    // `live`: this lane executes the batch; `tail`: some of its samples may lie beyond nsteps.
    // Skipped entirely while no lane can change state (nothing found yet, nothing >= threshold).
    typedef typename Texel<ESH>::type texel_t;
    // "(f32)m >= threshold" on the raw texel: integer texels compare as integers against ceil(threshold)
    // (P.lmip_threshold_raw, one past the largest value when none can reach it), f32 texels against the threshold itself
    texel_t thr_raw;
    if constexpr (ESH != 2) thr_raw = P.lmip_threshold_raw; else thr_raw = P.lmip_threshold;
    const bool mip_like = (P.skip_flags & 2) != 0;           // wave-uniform
    // SVR_MODE_WEIGHTED_AVERAGE (include/svr.h) is another reducer over the same batches: local_max holds the largest
    // contribution so far, samp the weighted sum, den the sum of weights; the lane is finished at its first weightless sample
    // (its own instantiations, WAVG: the LMIP kernels do not carry a single instruction of it)
    float den = 0.f;
    auto lmip_batch = [&](const texel_t (&sv)[U], int nb, bool live, bool tail) {
        if constexpr (WAVG) {
            const float steplen = sqrtf((Rtx * Rtx + Rty * Rty) + Rtz * Rtz);
            const float k = fresh_params(P)->weight_falloff;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool act = live && !finished && (!tail || (nb + u) < nsteps);
                const float tw = 1.0f - k * ((float)(nb + u) * steplen);
                const bool go = act && tw > 0.0f;
                finished = finished || (act && !go);
                if (COUNT) steps += go ? 1u : 0u;
                const float val = texel_value(sv[u]);
                const float w = tw * tw;
                const float contribution = w * fabsf(val);
                const float num1 = samp + w * val, den1 = den + w;
                samp = go ? num1 : samp;
                den = go ? den1 : den;
                const bool take = go && contribution > local_max;
                local_max = take ? contribution : local_max;
                hit_i = take ? (nb + u) : hit_i;
            }
            return;
        }
        // u8: 0 is neutral (a live lane always has one real sample); f32: -1 < every |s|
        texel_t m;
        if constexpr (ESH != 2) m = 0u; else m = -1.0f;
        const texel_t neutral = m;
#pragma unroll
        for (int u = 0; u < U; ++u) m = texel_max(m, (tail && (nb + u) >= nsteps) ? neutral : texel_abs(sv[u]));
        // a lane that follows a maximum needs the batch too — unless the machine can never stop (MIP: no fall-off, no
        // sample limit; host: skip_flags bit 1) and nothing in the batch beats its maximum (raycast.wgsl:50 is a strict >)
        // (the same decision written branch-free, both conditions evaluated first, measured 1.5 % slower: the compiler
        // then keeps the lane masks as 0 / 1 in VGPRs)
        bool need;
        if (mip_like) need = live && (found ? texel_value(m) > local_max : m >= thr_raw);
        else need = live && (found || m >= thr_raw);
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool act = live && !finished && (!tail || (nb + u) < nsteps);
                const float val = texel_value(sv[u]);
                const float inten = fabsf(val);
                if (COUNT) steps += act ? 1u : 0u;
                const bool was_found = found;
                const bool first_hit = act && !was_found && inten >= P.lmip_threshold;     // :37
                const bool tracking = act && was_found;
                since += tracking ? 1 : 0;                                                 // :47
                const bool take = first_hit || (tracking && inten > local_max);            // :50
                local_max = take ? inten : local_max;
                samp = take ? val : samp;
                hit_i = take ? (nb + u) : hit_i;
                found = found || first_hit;
                const bool brk = tracking && (since >= P.lmip_max_samples || inten < local_max * P.lmip_fall_off);  // :58
                finished = finished || brk;
            }
        } else if (COUNT) {
            steps += live ? (uint32_t)(tail ? (min(nb + U, nsteps) - nb) : U) : 0u;
        }
    };

    // span state: iterations [.., E) use LOD `code` (NL = none) with address constant Kc
    int code = NL, E = 0;
    uint32_t Kc = 0xFFFFFFFFu;
    uint32_t Kb = 0u;            // the same for the micro-block copy of the span's LOD (twin_offset), kept by waves that gather from it
    int zth = 0;                 // BIG builds: first voxel index ic_z whose ring plane lies in the LOD's upper resource

    // Wave-static routing of fast runs (u8 rings).  LDS bricks pay when the samples a wave fetches
    // together fall into many different lines per lane-quad (the L1 does one lookup per distinct line
    // of each quad).  Probe it: at one iteration, count how many lanes hit a line that no lower lane of
    // their quad hits; > kBrickQuadLines lookups per load -> stage bricks.  P.brick: 0 never, 1 probe, 2 always.
    int brick_mode = 0;                              // wave-uniform: 0 gathers only, k > 0 brick slabs of 2^(k-1) times the plain length
    bool many_lines = false;                         // wave-uniform, for good: the probe's verdict (or "always").  On a LOD that
                                                     // keeps a micro-block copy such a wave gathers from the copy instead of staging bricks
    if (P.brick) {
        const float pf = (float)min(max(nsteps - 1, 0), 256);
        const int qx = (int)((Rsx + pf * Rtx) * P.size[0]) >> (7 - ESH);   // 128 bytes per line
        const int qy = (int)((Rsy + pf * Rty) * P.size[1]);
        const int qz = (int)((Rsz + pf * Rtz) * P.size[2]);
        const int key = frag ? ((qz * 4099 + qy) * 64 + (qx & 63)) : -1 - lane;
        const int k0 = __builtin_amdgcn_update_dpp(key, key, 0x00, 0xF, 0xF, false);   // quad_perm [0,0,0,0]
        const int k1 = __builtin_amdgcn_update_dpp(key, key, 0x55, 0xF, 0xF, false);   // [1,1,1,1]
        const int k2 = __builtin_amdgcn_update_dpp(key, key, 0xAA, 0xF, 0xF, false);   // [2,2,2,2]
        const int q = lane & 3;
        const bool fresh = frag && (q == 0 || (key != k0 && (q == 1 || (key != k1 && (q == 2 || key != k2)))));
        const int lines = __builtin_popcountll(__builtin_amdgcn_ballot_w64(fresh));
        const int quads = __builtin_popcountll(__builtin_amdgcn_ballot_w64(frag && q == 0)) + 1;
        const bool use_brick = P.brick >= 2 || lines * 16 > P.brick_lines * quads;   // lookups per full wave-load
        brick_mode = use_brick ? 1 + P.slab_long : 0;
        many_lines = use_brick;
        // (tried: keep the bricks for waves whose rays all run along the ring's rows, |dx| > 4 max(|dy|, |dz|) — it takes the
        //  -x view of C2 from +6 % to +1.5 % against rings without a copy, and costs K1 2.5 % and C5's -x view its 7 % gain)
    }
    const int wave_lds = wave * P.brick_bytes;

    // diagnostics of the instrumented build (COUNT): wave-level batch census
    int c_general = 0, c_direct = 0, c_brick = 0, c_slabs = 0, c_runs = 0, c_zero = 0, c_skip = 0;
    int n = 0;                                       // wave-uniform: every ray starts at iteration 0
    lap(0);
    while (true) {
        const bool alive = !finished && n < nsteps;
        const unsigned long long alive_mask = __builtin_amdgcn_ballot_w64(alive);
        if (alive_mask == 0) break;

        // ---- span refresh for the lanes whose span has ended
        if (__builtin_amdgcn_ballot_w64(alive && n >= E) != 0) {
            const kparams_t Pr = fresh_params(P);
            if (alive && n >= E) {
                code = NL; E = nsteps; Kc = 0xFFFFFFFFu;
                bool settled = false;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const bool in = !settled && n >= ev[l].a && n < ev[l].b;
                    if (in) {
                        const LodK L = load_lod(Pr, l);
                        code = l;
                        E = min(E, ev[l].b);
                        // wrap constants in force at iteration n, and when they change next
                        const bool px = (n >= ev[l].cx) == (Rtx > 0.0f);   // slot = ic + addw - ring ?
                        const bool py = (n >= ev[l].cy) == (Rty > 0.0f);
                        const bool pz = (n >= ev[l].cz) == (Rtz > 0.0f);
                        const uint32_t kx = (uint32_t)L.addw[0] - (px ? L.ring[0] : 0u);
                        const uint32_t ky = (uint32_t)L.addw[1] - (py ? L.ring[1] : 0u);
                        const uint32_t kz = (uint32_t)L.addw[2] - (pz ? L.ring[2] : 0u);
                        Kc = (kz * L.ring[1] + ky) * L.rx4 + (kx << ESH) + L.base_bytes;  // mod 2^32
                        if (many_lines && L.twin) {                           // (wave-uniform)
                            typedef TwinBlock<ESH> B;
                            Kb = (uint32_t)(((int)kz >> B::ZB) * (int)(L.ring[1] >> B::YB) + ((int)ky >> B::YB)) * (L.ring[0] >> B::XB) +
                                 (uint32_t)((int)kx >> B::XB);
                        }
                        if constexpr (BIG) zth = (int)L.zsplit - (int)kz;                 // slot plane ic_z + kz >= k * zsplit <=> ic_z >= zth + (k - 1) * zsplit
                        if (ev[l].cx > n) E = min(E, ev[l].cx);
                        if (ev[l].cy > n) E = min(E, ev[l].cy);
                        if (ev[l].cz > n) E = min(E, ev[l].cz);
                    } else if (!settled && ev[l].a > n) {
                        E = min(E, ev[l].a);                 // a finer LOD takes over there
                    }
                    settled = settled || in;
                }
            }
        }

        // ---- how many whole batches can every live lane run on one LOD with constant addressing?
        const int first = __builtin_amdgcn_readlane(code, (int)__builtin_ctzll(alive_mask));
        // a span that ends at the ray end may finish with a partial batch: its samples beyond nsteps
        // are fetched from harmless (range-checked / LDS) addresses and masked in the LMIP update
        int lane_run = (code == first) ? ((E == nsteps ? E - n + U - 1 : E - n) / U) : 0;
        if (first < NL && !fresh_params(P)->lod_pow2[first < NL ? first : 0]) lane_run = alive ? 0 : lane_run;   // fused constant needs 2^-k scales
        if (!alive) lane_run = 0x3fffffff;
        int run = wave_reduce<false>(lane_run);
        if (COUNT) ++c_runs;
        lap(1);

        if (run <= 0) {
            if (COUNT) ++c_general;
            // ---- general batch: each sample evaluated exactly (intervals + explicit ring wrap)
            const kparams_t Pg = fresh_params(P);
            texel_t s[U];
            const float basef = (float)n;
            if constexpr (!BIG) {
                // all rings in one buffer resource: one load per sample serves the lanes of every LOD
                uint32_t off[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    off[u] = 0xFFFFFFFFu;
                    const float iter = basef + (float)u;
                    const float dx = (Rsx + iter * Rtx) * Pg->size[0];
                    const float dy = (Rsy + iter * Rty) * Pg->size[1];
                    const float dz = (Rsz + iter * Rtz) * Pg->size[2];
                    bool done = !alive || (n + u) >= nsteps;
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const bool sel = !done && (n + u) >= ev[l].a && (n + u) < ev[l].b;
                        if (__builtin_amdgcn_ballot_w64(sel) != 0) {
                            const uint32_t ofs = lod_offset_wrapped<ESH>(load_lod(Pg, l), dx, dy, dz);
                            off[u] = sel ? ofs : off[u];
                        }
                        done = done || sel;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = fetch_density<ESH>(rsrc_all, off[u]);
            } else {
                // rings of 4 GiB or more in total: each LOD has a buffer resource of its own, so the lanes of a
                // sample are served LOD by LOD (the range check makes the other lanes read nothing)
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float iter = basef + (float)u;
                    const float dx = (Rsx + iter * Rtx) * Pg->size[0];
                    const float dy = (Rsy + iter * Rty) * Pg->size[1];
                    const float dz = (Rsz + iter * Rtz) * Pg->size[2];
                    bool done = !alive || (n + u) >= nsteps;
                    texel_t acc = 0;
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const bool sel = !done && (n + u) >= ev[l].a && (n + u) < ev[l].b;
                        if (__builtin_amdgcn_ballot_w64(sel) != 0) {
                            const LodK Lg = load_lod(Pg, l);
                            uint32_t part;
                            const uint32_t ofs = lod_offset_wrapped_split<ESH>(Lg, dx, dy, dz, part);
                            // (a ring of 4 GiB or more: one load per part that some lane's texel lies in)
                            for (unsigned long long todo = __builtin_amdgcn_ballot_w64(sel); todo != 0ull;) {
                                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)part, (int)__builtin_ctzll(todo));
                                const bool mine = sel && part == p;
                                const texel_t t = fetch_density<ESH>(part_rsrc(Lg, p), mine ? ofs : 0xFFFFFFFFu);
                                acc = mine ? t : acc;
                                todo &= ~__builtin_amdgcn_ballot_w64(mine);
                            }
                        }
                        done = done || sel;
                    }
                    s[u] = acc;
                }
            }
            lmip_batch(s, n, alive, true);
            n += U;
            lap(2);
            continue;
        }

        if (first == NL) {
            // no LOD holds these voxels: every sample is 0 (sample_vol.wgsl:62)
            texel_t s[U];
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] = 0;
            for (; run > 0; --run) {
                if (COUNT) ++c_zero;
                const bool lv = alive && !finished && n < nsteps;
                if (__builtin_amdgcn_ballot_w64(lv && n + U > nsteps) != 0) lmip_batch(s, n, lv, true);
                else lmip_batch(s, n, lv, false);
                n += U;
                if (__builtin_amdgcn_ballot_w64(alive && !finished && n < nsteps) == 0) break;
            }
            continue;
        }

        {
            // the LOD every live lane is on: its constants are read here, once per run
            const LodK L = load_lod(fresh_params(P), first);
            // scale is 2^-k here, so (coord*size)*scale == coord*(size*scale) bit for bit (scaling by a
            // power of two commutes with rounding): one multiply per axis instead of two
            const float ssx = L.ss[0], ssy = L.ss[1], ssz = L.ss[2];      // size * scale, multiplied on the host
            // the buffer resource this LOD's texels come through (hardware range check returns 0 beyond it)
            __amdgpu_buffer_rsrc_t rsrc = rsrc_all;
            if constexpr (BIG) rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(L.rbase), 0, (int)L.rbytes, 0x00020000);
            // (a ring of 4 GiB or more comes through nparts resources of L.zsplit planes each: part_rsrc)
            const bool ring_split = BIG && L.nparts > 1u;                // wave-uniform
            // this wave's gathers on this LOD go to the micro-block copy of the ring (and it stages no bricks from the ring)
            // (LodParams::twin 1: instead of staging bricks from the ring; 2: only where this wave stages no bricks — a LOD that
            //  cannot, or a wave whose boxes stopped fitting its LDS region and that would gather from the rows from here on)
            const bool use_twin = many_lines && L.twin == 1u;

            // ---- empty-space skipping (LMIP mode; host: MarchParams::cells_all).  Per LOD the host keeps, for
            // cells of S^3 ring slots (S = 8 or 4), the largest value stored in the 2 x 2 x 2 block of cells that
            // starts at each cell.  A lane looks at its samples of iterations i and i + 8 B (B = skip_batches): it
            // travels at most one cell per axis between them (checked per lane), and its voxel index is monotone
            // along the ray, so every sample in between lies in the block that starts at the smaller of the two
            // cell coordinates: a block maximum below the threshold proves them all insignificant.  When that
            // holds for every live lane over 4 such stretches, and no lane is tracking a maximum, the wave
            // advances 4 B batches without fetching a texel: results and executed-iteration counts are those of
            // the reference, sample for sample (raycast.wgsl:35-44 does nothing on such samples).
            int held = 0;                                    // batches of this run held back behind a vetoed test
            do {
            run += held; held = 0;
            if (P.cells_all_bytes != 0u) {
                const kparams_t Ps = fresh_params(P);
                const int sb = Ps->lod[first].skip_batches;
                if (sb > 0 && run >= 4 * sb) {
                    const float reach = (float)(8 * sb);
                    const uint32_t cs = (uint32_t)Ps->lod[first].cshift;
                    const bool slow = fmaxf(fmaxf(fabsf(Rtx * ssx), fabsf(Rty * ssy)), fabsf(Rtz * ssz)) * reach <=
                                      (float)(1u << cs) - 0.5f;
                    __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<void*>(Ps->cells_all), 0, (int)Ps->cells_all_bytes, 0x00020000);
                    const uint32_t cbase = Ps->lod[first].cell_base;
                    const uint32_t cdx = Ps->lod[first].cdim[0], cdy = Ps->lod[first].cdim[1], cdz = Ps->lod[first].cdim[2];
                    bool vetoed = false, tracking_only = false;
                    // cell coordinates of the sample at iteration n (not beyond the lane's last sample): border 0 of the first
                    // test; every later test inherits it from its predecessor's last border
                    uint32_t c0x, c0y, c0z;
                    {
                        const float i0 = fminf((float)n, (float)(nsteps - 1));
                        c0x = ((uint32_t)(int)((i0 * Rtx + Rsx) * ssx) + (uint32_t)L.addw[0]) >> cs;
                        c0y = ((uint32_t)(int)((i0 * Rty + Rsy) * ssy) + (uint32_t)L.addw[1]) >> cs;
                        c0z = ((uint32_t)(int)((i0 * Rtz + Rsz) * ssz) + (uint32_t)L.addw[2]) >> cs;
                    }
                    while (run >= 4 * sb) {
                        const bool live = alive && !finished && n < nsteps;
                        // unwrapped cell coordinates ((index + addw) >> cs, in [0, 2 * cells)) of the samples at the 5
                        // stretch borders: iterations n, n + 8B, n + 16B, n + 24B and n + 32B — the first border of the
                        // next test, which inherits it — or, when this run ends with the group, its LAST iteration,
                        // n + 32B - 1 (the next one lies beyond this run: another LOD, another wrap); none beyond the
                        // lane's last sample
                        const float lastf = fminf((float)(nsteps - 1), (float)n + 4.0f * reach - (run > 4 * sb ? 0.0f : 1.0f));
                        uint32_t px[5], py[5], pz[5];
                        px[0] = c0x; py[0] = c0y; pz[0] = c0z;
#pragma unroll
                        for (int g = 1; g < 5; g += 2) {
                            const float i0 = (float)n + reach * (float)g;
                            const float2_t iter = { fminf(i0, lastf), fminf(i0 + reach, lastf) };
                            const Idx2 q = voxel_pair(Rsx, Rsy, Rsz, Rtx, Rty, Rtz, iter, ssx, ssy, ssz);
                            px[g] = (q.x0 + (uint32_t)L.addw[0]) >> cs; py[g] = (q.y0 + (uint32_t)L.addw[1]) >> cs; pz[g] = (q.z0 + (uint32_t)L.addw[2]) >> cs;
                            px[g + 1] = (q.x1 + (uint32_t)L.addw[0]) >> cs; py[g + 1] = (q.y1 + (uint32_t)L.addw[1]) >> cs;
                            pz[g + 1] = (q.z1 + (uint32_t)L.addw[2]) >> cs;
                        }
                        c0x = px[4]; c0y = py[4]; c0z = pz[4];
                        texel_t v[4];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // the block that starts at the smaller cell coordinate per axis, wrapped onto the ring of cells
                            uint32_t bx = min(px[g], px[g + 1]), by = min(py[g], py[g + 1]), bz = min(pz[g], pz[g + 1]);
                            bx = min(bx, bx - cdx); by = min(by, by - cdy); bz = min(bz, bz - cdz);
                            v[g] = fetch_density<ESH>(crsrc, (__umul24(__umul24(bz, cdy) + by, cdx) + bx) << ESH, cbase);
                        }
                        const texel_t m = texel_max(texel_max(texel_abs(v[0]), texel_abs(v[1])), texel_max(texel_abs(v[2]), texel_abs(v[3])));
                        // a lane that tracks a maximum, moves too fast for the test, or passes a block that may hold a
                        // significant value vetoes the skip
                        // (MIP: a lane that follows a maximum only vetoes for a block that may beat it)
                        const bool occupied = live && !found && (!slow || m >= thr_raw);
                        const bool following = live && found && (!mip_like || !slow || texel_value(m) > local_max);
                        const unsigned long long occ_mask = __builtin_amdgcn_ballot_w64(occupied);
                        if (occ_mask != 0 || __builtin_amdgcn_ballot_w64(following) != 0) {
                            vetoed = true;
                            tracking_only = occ_mask == 0;         // every vetoing lane is already following a maximum
                            break;
                        }
                        if (COUNT) { steps += live ? (uint32_t)(min(n + 32 * sb, nsteps) - n) : 0u; c_skip += 4 * sb; }
                        n += 32 * sb;
                        run -= 4 * sb;
                        if (__builtin_amdgcn_ballot_w64(alive && !finished && n < nsteps) == 0) { run = 0; break; }
                    }
                    if (vetoed) {
                        // march one test's worth (a whole brick slab where bricks are staged), then test again
                        // (a lane that is following a maximum is done within lmip_max_samples more samples: march just
                        // long enough for that, with plain gathers, and come back to skipping)
                        int cap = 4 * sb;
                        if (brick_mode && L.slab > 0 && !use_twin) cap = max(cap, (L.slab << (brick_mode - 1)) / U);
                        if (tracking_only && (P.skip_flags & 1) && !mip_like) cap = min(cap, 2);     // (MIP lanes follow their maximum to the ray's end)
                        if (run > cap) { held = run - cap; run = cap; }
                    }
                }
                lap(10);
            }

            // ---- LDS brick slabs (u8 rings).  The exact bounding box of the wave's samples over a
            // slab (ic is monotone per axis: first and last sample bound the rest) is staged into LDS
            // with coalesced 16-byte loads; the slab is then gathered from LDS, not through the L1,
            // which serves gathers one lane-quad at a time.
            {
                // slab length (host: about 12 ring voxels of travel; 0 where this LOD cannot stage bricks)
                // (a wave starts with slabs of twice that length: the staged bytes per sample fall with the slab
                // length; at its first box that does not fit the LDS region it drops to the plain length)
                while (brick_mode && L.slab > 0 && !use_twin && run >= (L.slab << (brick_mode - 1)) / U) {
                    const int slab = L.slab << (brick_mode - 1);
                    const bool live = alive && !finished && n < nsteps;
                    // first and last existing sample of the slab, both at once with the packed chain
                    const float2_t it = { (float)n, (float)min(n + slab - 1, nsteps - 1) };
                    const float2_t ex = (it * Rtx + Rsx) * ssx;
                    const float2_t ey = (it * Rty + Rsy) * ssy;
                    const float2_t ez = (it * Rtz + Rsz) * ssz;
                    // exact box of the wave's samples: min / max per axis over the live lanes
                    const int big = 0x7fffffff;
                    int lx = live ? min((int)ex.x, (int)ex.y) : big, hx = live ? max((int)ex.x, (int)ex.y) : -big;
                    int ly = live ? min((int)ey.x, (int)ey.y) : big, hy = live ? max((int)ey.x, (int)ey.y) : -big;
                    int lz = live ? min((int)ez.x, (int)ez.y) : big, hz = live ? max((int)ez.x, (int)ez.y) : -big;
#if defined(SVR_EXPERIMENTS) && defined(SVR_EXP_NO_BOX_REDUCE)
                    // ablation (wrong pixels): lane 0's box stands for the wave's — the six DPP reductions are gone
                    lx = __builtin_amdgcn_readfirstlane(lx); ly = __builtin_amdgcn_readfirstlane(ly); lz = __builtin_amdgcn_readfirstlane(lz);
                    hx = lx + 12; hy = ly + 10; hz = lz + 10;
#else
                    wave_min3_max3(lx, ly, lz, hx, hy, hz);
#endif
                    if (COUNT) ++c_slabs;
                    lap(8);
                    if (lx == big) { run = 0; break; }                   // no live lane left
                    // 16-voxel groups aligned in RING space, so a group never straddles the wrap
                    constexpr int GSH = 4 - ESH;                             // log2 of the voxels in a 16-byte group
                    const int gx0 = lx - ((lx + L.addw[0]) & ((1 << GSH) - 1));
                    const int ngx = ((hx - gx0) >> GSH) + 1;
                    const int ny = hy - ly + 1, nz = hz - lz + 1;
                    // LDS image: rows of gp 16-byte groups, planes of pz groups.  Both pitches are ODD numbers of
                    // groups (one group = 4 banks): the 8 rows / planes a wave's 8 x 8 pixel tile touches at one
                    // iteration then start in 8 different bank quads instead of 1, 2 or 4 (power-of-two pitches:
                    // 70 % of the LDS cycles were bank conflicts), and only the ngx groups the box really spans
                    // are fetched (a pitch rounded up to a power of two fetched up to twice that).
                    const int gp = brick_pow2 ? (ngx <= 1 ? 1 : 1 << (32 - __builtin_clz(ngx - 1))) : (ngx | 1);
                    const int pz = brick_pow2 ? ny * gp : ((ny * gp) | 1);
                    if (ny >= 512 || pz * nz * 16 > P.brick_bytes) {         // does not fit
                        if (brick_mode > 1) { --brick_mode; continue; }       // retry with half the slab
                        brick_mode = 0; break;                                // march direct from here on
                    }
                    // One load instruction per (z plane, chunk of 64 / gp rows): lane = row * gp + group fills its
                    // 16-byte slot at base + lane * 16; a lane's source offset is a per-lane constant (its row y and
                    // group) plus a wave-uniform z term, so the loop body is one VALU add.  Lanes of a padding
                    // group, or beyond the plane's rows, are masked off (they leave their slot alone).
                    const float rgp = __builtin_amdgcn_rcpf((float)gp);     // 1 ulp is plenty: see below
                    const int rows_per = (int)(64.5f * rgp);                 // floor(64 / gp): x.5 / gp is never an integer
                    const int ly_lane = (int)(((float)lane + 0.5f) * rgp);   // lane / gp, exactly
                    const int g_lane = lane - ly_lane * gp;
                    uint32_t wx = (uint32_t)(gx0 + (g_lane << GSH) + L.addw[0]);
                    wx = min(wx, wx - L.ring[0]);
                    wx <<= ESH;                                              // byte offset inside the ring row
                    const uint32_t plane_bytes = (uint32_t)(pz << 4);
                    const uint32_t zpitch = L.ring[1] * L.rx4;                       // bytes per ring z plane
                    lap(9);
#if defined(SVR_EXPERIMENTS) && defined(SVR_EXP_NO_BRICK_LOADS)
                    // ablation (wrong pixels): everything of a slab but the LDS-DMA loads themselves — what ANY reduction of
                    // their number or bytes (shared boxes, loader waves) could gain at most (tools/exp_ablate.sh)
                    if (false)
#endif
                    for (int yc = 0; yc < ny; yc += rows_per) {
                        const int yy = yc + ly_lane;
                        uint32_t wy = (uint32_t)(ly + yy + L.addw[1]);
                        wy = min(wy, wy - L.ring[1]);
                        const uint32_t lane_src = mad24(wy, L.rx4, L.base_bytes + wx);
                        const uint32_t lds_chunk = (uint32_t)wave_lds + (uint32_t)((yc * gp) << 4);
                        if (yy < ny && g_lane < ngx && ly_lane < rows_per) {
                            for (int zz = 0; zz < nz; ++zz) {
                                uint32_t wz = (uint32_t)(lz + zz + L.addw[2]);      // wave-uniform
                                wz = min(wz, wz - L.ring[2]);
                                if (BIG && ring_split) {                             // the part this plane lives in
                                    uint32_t p = 0u;
                                    while (p + 1u < L.nparts && wz >= (p + 1u) * L.zsplit) ++p;
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                        part_rsrc(L, p), (__attribute__((address_space(3))) void*)(lds_all + lds_chunk + zz * plane_bytes),
                                        16, (int)(lane_src + (wz - p * L.zsplit) * zpitch), 0, 0, 0);
                                } else
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                    rsrc, (__attribute__((address_space(3))) void*)(lds_all + lds_chunk + zz * plane_bytes),
                                    16, (int)(lane_src + wz * zpitch), 0, 0, 0);
                            }
                        }
                    }
                    // LDS byte address of voxel (ix,iy,iz) = iy * (gp * 16) + iz * (pz * 16) + ix * element size + bk
                    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*)lds_all);
                    const uint32_t bk = lds_base + (uint32_t)wave_lds - (uint32_t)(((lz * pz + ly * gp) << 4) + (gx0 << ESH));
                    lap(3);
                    if (!(dbg_nowait & 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lap(4);
                    // one dot2 on the packed (y, z): y * rowpitch + z * planepitch + (x + bk)
                    const uint32_t kyz = (uint32_t)(gp << 4) | ((uint32_t)(pz << 4) << 16);
                    const bool slab_tail = __builtin_amdgcn_ballot_w64(live && n + slab > nsteps) != 0;   // some ray ends inside this slab
                    for (int k = 0; k < slab / U; ++k) {
                        texel_t s[U];
                        float2_t iter = { (float)n, (float)n + 1.0f };
#pragma unroll
                        for (int u = 0; u < U; u += 2) {
                            const Idx2p v = voxel_pair_packed(Rsx, Rsy, Rsz, Rtx, Rty, Rtz, iter, ssx, ssy, ssz);
                            iter += 2.0f;
                            const uint32_t a0 = dot2_u16(v.yz0, kyz, shl_add_c<ESH>(v.x0, bk));
                            const uint32_t a1 = dot2_u16(v.yz1, kyz, shl_add_c<ESH>(v.x1, bk));
                            s[u] = lds_texel<ESH>(a0);
                            s[u + 1] = lds_texel<ESH>(a1);
                        }
                        // pin the zero-extended bytes where they are loaded (ds_read_u8 already extends;
                        // otherwise the extension is re-done with a v_and per sample in the consumer block)
                        // (one statement for the whole batch: one wait for the 8 reads instead of 8)
                        if constexpr (ESH != 2 && U == 8)
                            asm("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]));
                        {
                            const bool lv = alive && !finished && n < nsteps;
                            // keep the no-tail case a compile-time constant (its per-sample tests fold away); a ray can only
                            // end inside this slab if slab_tail says so (one test per slab instead of one per batch)
                            if (slab_tail && __builtin_amdgcn_ballot_w64(lv && n + U > nsteps) != 0) lmip_batch(s, n, lv, true);
                            else lmip_batch(s, n, lv, false);
                        }
                        n += U;
                        if (COUNT) ++c_brick;
                    }
                    run -= slab / U;
                    lap(5);
                    asm volatile("" ::: "memory");      // the next slab's loads must not overtake these LDS reads
                }
            }

            // ---- direct fast batches: texel offset = ((iz*Ry + iy)*Rx + ix)*es + Kc, U loads in flight
            // (two copies of the loop, one per layout the wave gathers from: each keeps only its own constants live)
            __amdgpu_buffer_rsrc_t rsrc_twin = rsrc;              // (the copy's own resource and base: set where the wave turns to it)
            const void* twin_base = nullptr;
            auto direct_batches = [&](auto from_twin) {
            constexpr bool TW = decltype(from_twin)::value;
            for (; run > 0; --run) {
                const bool live = alive && !finished && n < nsteps;
                if (__builtin_amdgcn_ballot_w64(live) == 0) break;
                if (COUNT) { ++c_direct; if (TW) t_acc[15] += 1ull; }          // ([15]: a count, not cycles — svr_debug_timers)
                texel_t s[U];
                uint32_t off[U];
                float2_t iter = { (float)n, (float)n + 1.0f };
                if constexpr (TW) {
                    const TwinConsts tc = twin_consts<ESH>(L.ring[0], L.ring[1]);
                    const uint32_t KbS = Kb << 7;
#pragma unroll
                    for (int u = 0; u < U; u += 2) {
                        const Idx2p v = voxel_pair_packed(Rsx, Rsy, Rsz, Rtx, Rty, Rtz, iter, ssx, ssy, ssz);
                        iter += 2.0f;
                        off[u] = twin_offset_packed<ESH>(v.x0, v.yz0, tc, KbS);
                        off[u + 1] = twin_offset_packed<ESH>(v.x1, v.yz1, tc, KbS);
                    }
                } else {
#pragma unroll
                for (int u = 0; u < U; u += 2) {
                    const Idx2 v = voxel_pair(Rsx, Rsy, Rsz, Rtx, Rty, Rtz, iter, ssx, ssy, ssz);
                    iter += 2.0f;
                    off[u] = mad24(mad24(v.z0, L.ring[1], v.y0), L.rx4, shl_add_c<ESH>(v.x0, Kc));
                    off[u + 1] = mad24(mad24(v.z1, L.ring[1], v.y1), L.rx4, shl_add_c<ESH>(v.x1, Kc));
                }
                }
                // lanes that are not live fetch nothing; the LMIP update never looks at their samples
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = undefined_value<texel_t>();
                if (BIG && ring_split) {
                    // a ring of 4 GiB or more: a lane's texel comes through the resource of the part that holds its plane
                    // (offsets are mod 2^32 relative to part 0: minus p parts' sizes = relative to part p)
                    // ic_z is monotone along a ray: the parts of a lane's first and last sample of the batch bound those of the
                    // samples in between.  Nearly always every live lane's batch lies in ONE part: one resource, no per-sample test
                    auto part_of = [&](int icz) {
                        uint32_t q = 0u;
                        for (uint32_t k = 1; k < L.nparts; ++k) q += icz >= zth + (int)((k - 1u) * L.zsplit) ? 1u : 0u;
                        return q;
                    };
                    const float2_t ends = { (float)n, (float)min(n + U - 1, nsteps - 1) };     // (samples beyond the ray's end are never looked at)
                    const float2_t ez2 = (ends * Rtz + Rsz) * ssz;
                    const uint32_t pa = part_of((int)ez2.x), pb = part_of((int)ez2.y);
                    const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)pa, (int)__builtin_ctzll(__builtin_amdgcn_ballot_w64(live)));
                    if (__builtin_amdgcn_ballot_w64(live && (pa != p0 || pb != p0)) == 0) {
                        const __amdgpu_buffer_rsrc_t rp = part_rsrc(L, p0, TW ? twin_base : L.rbase);
                        const uint32_t sub = p0 * L.part_bytes;
#pragma unroll
                        for (int u = 0; u < U; ++u) s[u] = fetch_density<ESH>(rp, live ? off[u] - sub : 0xFFFFFFFFu);
                    } else {
                    float2_t it2 = { (float)n, (float)n + 1.0f };
#pragma unroll
                    for (int u = 0; u < U; u += 2) {
                        const float2_t cz = (it2 * Rtz + Rsz) * ssz;      // the z index of the pair again (this path is rare)
                        it2 += 2.0f;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int icz = h ? (int)cz.y : (int)cz.x;
                            uint32_t part = 0u;
                            for (uint32_t k = 1; k < L.nparts; ++k) part += icz >= zth + (int)((k - 1u) * L.zsplit) ? 1u : 0u;
                            const uint32_t rel = off[u + h] - part * L.part_bytes;
                            texel_t a = 0;
                            for (unsigned long long todo = __builtin_amdgcn_ballot_w64(live); todo != 0ull;) {
                                const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)part, (int)__builtin_ctzll(todo));
                                const bool mine = live && part == p;
                                const texel_t t = fetch_density<ESH>(part_rsrc(L, p, TW ? twin_base : L.rbase), mine ? rel : 0xFFFFFFFFu);
                                a = mine ? t : a;
                                todo &= ~__builtin_amdgcn_ballot_w64(mine);
                            }
                            s[u + h] = a;
                        }
                    }
                    }
                } else
                if (live) {
#pragma unroll
                    for (int u = 0; u < U; ++u) s[u] = fetch_density<ESH>(TW ? rsrc_twin : rsrc, off[u]);
                }
                if (__builtin_amdgcn_ballot_w64(live && n + U > nsteps) != 0) lmip_batch(s, n, live, true);
                else lmip_batch(s, n, live, false);
                n += U;
            }
            };
            if (use_twin || (many_lines && L.twin == 2u && (brick_mode == 0 || L.slab <= 0))) {
                const kparams_t Pt = fresh_params(P);
                twin_base = Pt->lod[first].twin_rbase;
                rsrc_twin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(twin_base), 0, (int)Pt->lod[first].twin_rbytes, 0x00020000);
                direct_batches(std::true_type{});
            } else {
                direct_batches(std::false_type{});
            }
            } while (held > 0 && __builtin_amdgcn_ballot_w64(alive && !finished && n < nsteps) != 0);
            lap(6);
        }
    }

    if (COUNT && P.dbg && lane == 0) {
        atomicAdd(P.dbg + 0, (uint32_t)c_general); atomicAdd(P.dbg + 1, (uint32_t)c_direct);
        atomicAdd(P.dbg + 2, (uint32_t)c_brick);   atomicAdd(P.dbg + 3, (uint32_t)c_slabs);
        atomicAdd(P.dbg + 4, (uint32_t)c_runs);    atomicAdd(P.dbg + 5, (uint32_t)c_zero);
        atomicAdd(P.dbg + 6, 1u);                  atomicAdd(P.dbg + 7, (uint32_t)c_skip);
    }
    Hit h;
    h.found = found; h.sample = samp; h.steps = steps;
    if constexpr (WAVG) { h.found = local_max > 0.0f; h.sample = h.found ? samp / den : 0.0f; }
    const float hit_f = (float)hit_i;
    h.offset = { hit_f * Rtx, hit_f * Rty, hit_f * Rtz };                    // raycast.wgsl:30
    h.coord = { Rsx + h.offset.x, Rsy + h.offset.y, Rsz + h.offset.z };   // :31
    if (inside) {
        // the epilogue re-reads its (cold) uniforms through a laundered kernarg pointer so that
        // they are not kept live in SGPRs across the march loop
#if defined(__HIP_DEVICE_COMPILE__)
        const MarchParams* Pk = (const MarchParams*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(Pk));
#else
        const MarchParams* Pk = &P;
#endif
        shade_and_store<NL>(*Pk, o, frag, h);
        if (COUNT && Pk->steps) Pk->steps[o] = h.steps;
    }
    if (COUNT && P.dbg) {
        lap(7);
        if (lane == 0) {
            unsigned long long* t_out = reinterpret_cast<unsigned long long*>(P.dbg + 8);
#pragma unroll
            for (int k = 0; k < 16; ++k) atomicAdd(t_out + k, t_acc[k]);
        }
    }
}

template <int NL>
hipError_t launch_nl(const MarchParams& p, int kind, hipStream_t stream) {
    const int nblocks = p.tiles_x * p.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    const bool simple = kind == 1 || !p.span_ok;                 // rings the 32-bit / 24-bit addressing of the span kernel cannot reach
    const bool wavg = p.render_mode == SVR_MODE_WEIGHTED_AVERAGE;
    if (wavg && (simple || p.per_lod_rsrc)) {
        if (p.steps) hipLaunchKernelGGL((march_wavg<NL, true>), dim3(nblocks), dim3(256), 0, stream, p);
        else         hipLaunchKernelGGL((march_wavg<NL, false>), dim3(nblocks), dim3(256), 0, stream, p);
    } else if (simple) {
        if (p.steps) hipLaunchKernelGGL((march_simple<NL, true>), dim3(nblocks), dim3(256), 0, stream, p);
        else         hipLaunchKernelGGL((march_simple<NL, false>), dim3(nblocks), dim3(256), 0, stream, p);
    } else {
        const int threads = 64 << (2 * p.block_waves_log2);
        const size_t lds = (size_t)p.brick_bytes * (threads / 64);
#define SVR_LAUNCH_SPAN(ESH_)                                                                                              \
    do {                                                                                                                   \
        if (p.per_lod_rsrc) {                                                                                              \
            if (p.steps) hipLaunchKernelGGL((march_span<NL, 8, true, ESH_, true>), dim3(nblocks), dim3(threads), lds, stream, p);   \
            else         hipLaunchKernelGGL((march_span<NL, 8, false, ESH_, true>), dim3(nblocks), dim3(threads), lds, stream, p);  \
        } else {                                                                                                           \
            if (p.steps) hipLaunchKernelGGL((march_span<NL, 8, true, ESH_, false>), dim3(nblocks), dim3(threads), lds, stream, p);  \
            else         hipLaunchKernelGGL((march_span<NL, 8, false, ESH_, false>), dim3(nblocks), dim3(threads), lds, stream, p); \
        }                                                                                                                  \
    } while (0)
#define SVR_LAUNCH_WAVG(ESH_)                                                                                              \
    do {                                                                                                                   \
        if (p.steps) hipLaunchKernelGGL((march_span<NL, 8, true, ESH_, false, true>), dim3(nblocks), dim3(threads), lds, stream, p);   \
        else         hipLaunchKernelGGL((march_span<NL, 8, false, ESH_, false, true>), dim3(nblocks), dim3(threads), lds, stream, p);  \
    } while (0)
        if (wavg) {
            if (p.density_esh == 0) SVR_LAUNCH_WAVG(0);
            else if (p.density_esh == 1) SVR_LAUNCH_WAVG(1);
            else SVR_LAUNCH_WAVG(2);
        }
        else if (p.density_esh == 0) SVR_LAUNCH_SPAN(0);
        else if (p.density_esh == 1) SVR_LAUNCH_SPAN(1);
        else SVR_LAUNCH_SPAN(2);
#undef SVR_LAUNCH_WAVG
#undef SVR_LAUNCH_SPAN
    }
    return hipGetLastError();
}

}  // namespace

// One translation unit per LOD count (compiled in parallel with -DSVR_NL=k): the
// reference specialises its shader on num_scales the same way (_shader.py:73).
#ifndef SVR_NL
#error "compile with -DSVR_NL=<number of LODs>"
#endif
#define SVR_CAT2(a, b) a##b
#define SVR_CAT(a, b) SVR_CAT2(a, b)
hipError_t SVR_CAT(svr_launch_march_nl, SVR_NL)(const MarchParams& p, int kind, hipStream_t stream) {
    return launch_nl<SVR_NL>(p, kind, stream);
}
