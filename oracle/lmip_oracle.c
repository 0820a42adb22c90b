/*
 * lmip_oracle.c — CPU restatement of the reference's LMIP ray-march shader.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sub_volume_renderer_amd/ may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the CPU baseline.
 *
 * PARITY STATUS: *render parity unpinned*.  The reference (gyoge0/
 * sub_volume_renderer @ 2025-08-15) cannot be imported or executed offline
 * (pygfx / wgpu / funlib.geometry absent, no WGSL runtime) and none of its own
 * tests pins a rendered pixel (tests/basic_volume/ is stale).
 * This file therefore follows the WGSL sources line by line; each function
 * cites the lines it restates.  Two pieces of arithmetic live in pygfx 0.12.0
 * (pixi.lock:1240), which is NOT under /root/reference, and are restated from
 * its published shader text:
 *   sampled_value_to_color (pygfx image_sample.wgsl):  v=(r-clim0)/(clim1-clim0); pow(v,gamma); grey
 *   srgb2physical          (pygfx std.wgsl):           c<=0.04045 ? c/12.92 : pow((c+0.055)/1.055, 2.4)
 *
 * Conventions fixed by this restatement (the HIP kernel implements the same
 * ones; see DESIGN.md "operation-order contract"):
 *   - strict IEEE f32, compiled with -ffp-contract=off, no fast-math;
 *   - mat4 are column-major m[c*4+r]; M*v = ((M0*x + M1*y) + M2*z) + M3*w;
 *     A*B column c = A * (column c of B);
 *   - dot(a,b) = (ax*bx + ay*by) + az*bz; length(v) = sqrtf(dot(v,v));
 *   - min/max are fminf/fmaxf (NaN-ignoring);
 *   - vec3<i32>(f) truncates toward zero;
 *   - float `%` by the (integer-valued) ring extent followed by vec3<i32>()
 *     is evaluated as exact integer modulo of the truncated coordinate
 *     (== exact fmodf; documented deviation from a lowering that evaluates
 *     x - y*trunc(x/y) in f32 — SURVEY.md §7 hard part 4);
 *   - the rasterised back face (vs_main.wgsl:36 + front-face culling) is
 *     replaced by the analytic exit point of the pixel's ray from the proxy
 *     box, kept only if it lies inside the clip volume (w>0, 0<=z<=w).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/svr.h"

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

typedef struct svr_oracle_lod {
    const float*    density;     /* ring texture, C-contiguous [z][y][x] */
    const uint32_t* labels;
    int32_t         ring_dims[3];/* (x,y,z) */
    svr_lod_state   st;          /* u_wrapping_buffer_i */
} svr_oracle_lod;

static v4 mat_vec(const float* m, v4 v) {
    v4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8]  * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9]  * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}

static void mat_mul(const float* a, const float* b, float* out) {
    for (int c = 0; c < 4; ++c) {
        v4 col = { b[c * 4 + 0], b[c * 4 + 1], b[c * 4 + 2], b[c * 4 + 3] };
        v4 r = mat_vec(a, col);
        out[c * 4 + 0] = r.x; out[c * 4 + 1] = r.y; out[c * 4 + 2] = r.z; out[c * 4 + 3] = r.w;
    }
}

static float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static float length3(v3 a) { return sqrtf(dot3(a, a)); }

static int floor_mod(int a, int n) { int r = a % n; return r < 0 ? r + n : r; }

/* try_sample_scale_i / try_sample_segmentations_scale_i: sample_vol.wgsl:4-25, :27-48.
 * Returns 1 and the linear texel index when the voxel is inside LOD i's ROI. */
static int lod_texel(const svr_oracle_lod* L, v3 coord, v3 sizef, size_t* idx) {
    /* :6  let data_coord = data_tex_coord * sizef; */
    v3 d = { coord.x * sizef.x, coord.y * sizef.y, coord.z * sizef.z };
    /* :7-8 scaled_data_coord = data_coord * scale_factor */
    v3 sd = { d.x * L->st.scale[0], d.y * L->st.scale[1], d.z * L->st.scale[2] };
    /* :17 in_bounds = all(offset <= vec3<i32>(sd)) && all(vec3<i32>(sd) < offset + shape) */
    int ix = (int)sd.x, iy = (int)sd.y, iz = (int)sd.z;
    const int32_t* o = L->st.offset; const int32_t* s = L->st.shape;
    int in_bounds = o[0] <= ix && o[1] <= iy && o[2] <= iz &&
                    ix < o[0] + s[0] && iy < o[1] + s[1] && iz < o[2] + s[2];
    if (!in_bounds) return 0;                                   /* :18-20 */
    /* :22-23 textureLoad(t, vec3<i32>(sd % ring_dims)) — exact integer modulo */
    int wx = floor_mod(ix, L->ring_dims[0]);
    int wy = floor_mod(iy, L->ring_dims[1]);
    int wz = floor_mod(iz, L->ring_dims[2]);
    *idx = ((size_t)wz * (size_t)L->ring_dims[1] + (size_t)wy) * (size_t)L->ring_dims[0] + (size_t)wx;
    return 1;
}

/* sample_vol / sample_vol_multi_scale: sample_vol.wgsl:51-63, :80-86.
 * First LOD whose ROI contains the voxel wins, even if its value is 0. */
static float sample_vol(int n, const svr_oracle_lod* lods, v3 coord, v3 sizef) {
    for (int i = 0; i < n; ++i) {
        size_t idx;
        if (lod_texel(&lods[i], coord, sizef, &idx)) return lods[i].density[idx];
    }
    return 0.0f;
}

/* sample_segmentations_vol: sample_vol.wgsl:65-77, :88-94 */
static uint32_t sample_seg(int n, const svr_oracle_lod* lods, v3 coord, v3 sizef) {
    for (int i = 0; i < n; ++i) {
        size_t idx;
        if (lod_texel(&lods[i], coord, sizef, &idx)) return lods[i].labels[idx];
    }
    return 0u;
}

/* hsv_to_rgb: hsv_selection.wgsl:7-41 */
static v3 hsv_to_rgb(float h, float s, float v) {
    v3 r;
    if (s == 0.0f) { r.x = v; r.y = v; r.z = v; return r; }      /* :13-15 */
    float h_scaled = h * 6.0f;                                    /* :18 */
    float fl = floorf(h_scaled);
    int sector = (int)fl;                                         /* :19 */
    float fractional = h_scaled - fl;                             /* :20 */
    float p = v * (1.0f - s);                                     /* :23 */
    float q = v * (1.0f - s * fractional);                        /* :24 */
    float t = v * (1.0f - s * (1.0f - fractional));               /* :25 */
    if (sector == 0)      { r.x = v; r.y = t; r.z = p; }
    else if (sector == 1) { r.x = q; r.y = v; r.z = p; }
    else if (sector == 2) { r.x = p; r.y = v; r.z = t; }
    else if (sector == 3) { r.x = p; r.y = q; r.z = v; }
    else if (sector == 4) { r.x = t; r.y = p; r.z = v; }
    else                  { r.x = v; r.y = p; r.z = q; }          /* :38-40 */
    return r;
}

/* pygfx std.wgsl srgb2physical (third-party, restated; see header) */
static float srgb2physical(float c) {
    float f = powf((c + 0.055f) / 1.055f, 2.4f);
    float t = c / 12.92f;
    return (c <= 0.04045f) ? t : f;
}

typedef struct {
    const svr_camera* cam;
    const svr_material* mat;
    int n; const svr_oracle_lod* lods;
    float ndc_to_data[16];   /* world_inv * cam_inv * proj_inv   (vs_main.wgsl:22) */
    float pc[16];            /* proj * cam                        (vs_main.wgsl:19, fs_main.wgsl:63) */
    v3 sizef;
    float rel_step;
} frame_ctx;

/* One pixel = one fs_main invocation (or none).  Returns the SVR_PIX_* class. */
static int shade_pixel(const frame_ctx* F, int W, int H, int i, int j,
                       float rgba[4], float* depth, uint32_t* label, uint32_t* steps, float hit_coord[3]) {
    const svr_camera* C = F->cam; const svr_material* M = F->mat;
    const v3 sizef = F->sizef;
    rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f; *depth = 0.0f; *label = 0u; *steps = 0u;

    /* pixel centre -> NDC; varyings data_near_pos / data_far_pos (vs_main.wgsl:44-47)
     * after perspective-correct interpolation = ndc_to_data * (px,py,-+1,1) up to scale */
    float px = (2.0f * ((float)i + 0.5f)) / (float)W - 1.0f;
    float py = 1.0f - (2.0f * ((float)j + 0.5f)) / (float)H;
    v4 n4 = mat_vec(F->ndc_to_data, (v4){ px, py, -1.0f, 1.0f });
    v4 f4 = mat_vec(F->ndc_to_data, (v4){ px, py,  1.0f, 1.0f });
    /* fs_main.wgsl:24-25 */
    v3 far_pos  = { f4.x / f4.w, f4.y / f4.w, f4.z / f4.w };
    v3 near_pos = { n4.x / n4.w, n4.y / n4.w, n4.z / n4.w };
    /* fs_main.wgsl:28 view_ray = normalize(far - near) */
    v3 dir = { far_pos.x - near_pos.x, far_pos.y - near_pos.y, far_pos.z - near_pos.z };
    float len = length3(dir);
    v3 ray = { dir.x / len, dir.y / len, dir.z / len };

    /* back_pos (fs_main.wgsl:23): exit point of the ray from the proxy box
     * [-0.5, size-0.5]^3 (volume_common.wgsl:26-27), i.e. the culled-front-face
     * rasterisation of vs_main.wgsl:36 done analytically. */
    float lo = -0.5f;
    float hx = sizef.x - 0.5f, hy = sizef.y - 0.5f, hz = sizef.z - 0.5f;
    float tx1 = (lo - near_pos.x) / ray.x, tx2 = (hx - near_pos.x) / ray.x;
    float ty1 = (lo - near_pos.y) / ray.y, ty2 = (hy - near_pos.y) / ray.y;
    float tz1 = (lo - near_pos.z) / ray.z, tz2 = (hz - near_pos.z) / ray.z;
    float t_exit  = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    float t_enter = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    if (!(t_enter <= t_exit)) return SVR_PIX_DISCARD;
    v3 back = { near_pos.x + ray.x * t_exit, near_pos.y + ray.y * t_exit, near_pos.z + ray.z * t_exit };
    /* the back-face fragment exists only inside the clip volume */
    v4 bw = mat_vec(C->world, (v4){ back.x, back.y, back.z, 1.0f });
    v4 bc = mat_vec(F->pc, bw);
    if (!(bc.w > 0.0f) || !(bc.z >= 0.0f) || !(bc.z <= bc.w)) return SVR_PIX_DISCARD;

    /* fs_main.wgsl:8  {$ include 'pygfx.clipping_planes.wgsl' $}  (pygfx 0.12, third party, restated from its
     * published text — assumption A6): with n_clipping_planes > 0
     *     var clipped = (clipping_mode == 'ANY') ? false : true;
     *     for each plane: plane_clipped = dot(varyings.world_pos, plane.xyz) < plane.w;
     *                     clipped = clipped || plane_clipped   (ANY)   /   clipped && plane_clipped   (ALL)
     *     if (clipped) { discard; }
     * varyings.world_pos is the world position of the BACK-face fragment (vs_main.wgsl:27), so a clipped
     * fragment removes the whole ray.  No planes (the default): the include expands to nothing. */
    if (M->clipping_plane_count > 0) {
        int clipped = M->clipping_mode_all ? 1 : 0;
        for (uint32_t k = 0; k < M->clipping_plane_count; ++k) {
            const float* pl = M->clipping_planes + 4u * k;
            v3 wpos = { bw.x, bw.y, bw.z }, abc = { pl[0], pl[1], pl[2] };
            int plane_clipped = dot3(wpos, abc) < pl[3];
            clipped = M->clipping_mode_all ? (clipped && plane_clipped) : (clipped || plane_clipped);
        }
        if (clipped) return SVR_PIX_DISCARD;
    }

    /* fs_main.wgsl:32-35 */
    v3 nb = { near_pos.x - back.x, near_pos.y - back.y, near_pos.z - back.z };
    float dist = dot3(nb, ray);
    dist = fmaxf(dist, fminf((-0.5f - back.x) / ray.x, (sizef.x - 0.5f - back.x) / ray.x));
    dist = fmaxf(dist, fminf((-0.5f - back.y) / ray.y, (sizef.y - 0.5f - back.y) / ray.y));
    dist = fmaxf(dist, fminf((-0.5f - back.z) / ray.z, (sizef.z - 0.5f - back.z) / ray.z));
    /* :39 */
    v3 front = { back.x + ray.x * dist, back.y + ray.y * dist, back.z + ray.z * dist };
    /* :43-44  nsteps = i32(-dist / relative_step_size + 0.5); if nsteps < 1 { discard; } */
    float nf = -dist / F->rel_step + 0.5f;
    if (!(nf >= 1.0f)) return SVR_PIX_DISCARD;
    if (nf > 16777216.0f) nf = 16777216.0f;       /* guard: keeps the f32 loop counter exact */
    int nsteps = (int)nf;
    /* :47-48 */
    v3 start = { (front.x + 0.5f) / sizef.x, (front.y + 0.5f) / sizef.y, (front.z + 0.5f) / sizef.z };
    float nstepsf = (float)nsteps;
    v3 step = { ((back.x - front.x) / sizef.x) / nstepsf,
                ((back.y - front.y) / sizef.y) / nstepsf,
                ((back.z - front.z) / sizef.z) / nstepsf };

    /* ---- raycast (raycast.wgsl:11-88) */
    float lmip_threshold = M->lmip_threshold, lmip_fall_off = M->lmip_fall_off;
    int lmip_max_samples = M->lmip_max_samples;
    float local_max_sample = 0.0f, local_max_intensity = 0.0f;
    v3 local_max_offset = { 0, 0, 0 }, local_max_coord = { 0, 0, 0 };
    int found = 0, since = 0;
    uint32_t executed = 0;
    if (M->render_mode == SVR_MODE_WEIGHTED_AVERAGE) {
        /* NOT in the reference: FUTURE.md:97-109 names a "weighted average" mode ("weight each sample by
         * distance", "sampling a finite number of points based on distance") and gives no formula.  The
         * definition is this project's (include/svr.h, SVR_MODE_WEIGHTED_AVERAGE); this loop is its CPU twin and
         * keeps the ray, the sample positions and the LOD fall-through of raycast.wgsl:29-32. */
        float steplen = length3(step);
        float num = 0.0f, den = 0.0f, best = 0.0f;
        for (float iter = 0.0f; iter < nstepsf; iter = iter + 1.0f) {
            float tw = 1.0f - M->weight_falloff * (iter * steplen);
            if (!(tw > 0.0f)) break;
            ++executed;
            v3 offset = { iter * step.x, iter * step.y, iter * step.z };
            v3 coord = { start.x + offset.x, start.y + offset.y, start.z + offset.z };
            float sample = sample_vol(F->n, F->lods, coord, sizef);
            float w = tw * tw;
            num = num + w * sample;
            den = den + w;
            float contribution = w * fabsf(sample);
            if (contribution > best) { best = contribution; local_max_offset = offset; local_max_coord = coord; }
        }
        found = best > 0.0f;
        local_max_sample = found ? num / den : 0.0f;
    } else
    for (float iter = 0.0f; iter < nstepsf; iter = iter + 1.0f) {                 /* :29 */
        ++executed;
        v3 offset = { iter * step.x, iter * step.y, iter * step.z };              /* :30 */
        v3 coord = { start.x + offset.x, start.y + offset.y, start.z + offset.z };/* :31 */
        float sample = sample_vol(F->n, F->lods, coord, sizef);                   /* :32 */
        float intensity = fabsf(sample);       /* :33 length((r,0,0)) for a 1-channel texture */
        if (!found) {
            if (intensity >= lmip_threshold) {                                    /* :37-44 */
                found = 1; local_max_intensity = intensity; local_max_sample = sample;
                local_max_offset = offset; local_max_coord = coord; since = 0;
            }
        } else {
            since += 1;                                                           /* :47 */
            if (intensity > local_max_intensity) {                                /* :50-55 */
                local_max_intensity = intensity; local_max_sample = sample;
                local_max_offset = offset; local_max_coord = coord;
            }
            if (since >= lmip_max_samples || intensity < local_max_intensity * lmip_fall_off) break; /* :58-60 */
        }
    }
    *steps = executed;

    if (!found) {                       /* fs_main.wgsl:93-98 */
        rgba[0] = 0.0f; rgba[1] = 0.0f; rgba[2] = 0.0f; rgba[3] = 1.0f; *depth = 0.0f;
        return SVR_PIX_MISS;
    }
    /* raycast.wgsl:69 sampled_value_to_color (pygfx, restated) */
    float v = (local_max_sample - M->clim[0]) / (M->clim[1] - M->clim[0]);
    v = powf(v, M->gamma);
    /* :71-75 */
    float phys = M->colorspace_srgb ? srgb2physical(v) : v;
    /* :81 */
    uint32_t seg = sample_seg(F->n, F->lods, local_max_coord, sizef);

    /* fs_main.wgsl:61-72 */
    v4 dp = { local_max_coord.x - 0.5f, local_max_coord.y - 0.5f, local_max_coord.z - 0.5f, 1.0f };
    v4 wp = mat_vec(C->world, dp);
    v4 ndc = mat_vec(F->pc, wp);
    *depth = ndc.z / fmaxf(ndc.w, 0.001f);
    /* :74-76 + hsv_selection.wgsl:1-3 */
    const float* hs = M->colors + 4u * (seg % M->color_count);
    v3 rgb = hsv_to_rgb(hs[0], hs[1], phys);
    /* :78-84 */
    float distance = length3(local_max_offset);
    float fog = expf(-M->fog_density * distance);
    float omf = 1.0f - fog;
    rgba[0] = M->fog_color[0] * omf + rgb.x * fog;      /* mix(a,b,t) = a*(1-t) + b*t */
    rgba[1] = M->fog_color[1] * omf + rgb.y * fog;
    rgba[2] = M->fog_color[2] * omf + rgb.z * fog;
    rgba[3] = M->opacity;                               /* :86 */
    *label = seg;
    hit_coord[0] = local_max_coord.x; hit_coord[1] = local_max_coord.y; hit_coord[2] = local_max_coord.z;
    return SVR_PIX_HIT;
}

/* fs_main.wgsl:89-92 (`write_pick` variant): pick_pack(id, 20) + pick_pack(u32(coord.x * 16383.0), 14) + ... y, z.
 * pygfx's pick_pack (std.wgsl, third party, restated from its published text) clips the value to `bits` bits and
 * places it at the running bit offset of a 64-bit word stored as four 16-bit components; WGSL's u32(f32) clamps
 * to the u32 range.  Parity unpinned (no pick fixture exists upstream). */
static uint32_t pick_field(float c) {
    const float f = c * 16383.0f;
    uint32_t u = 0u;
    if (f > 0.0f) u = f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f;
    return u < 16383u ? u : 16383u;
}
static uint64_t pick_word(uint32_t id, const float c[3]) {
    return (uint64_t)(id < 0xFFFFFu ? id : 0xFFFFFu) | ((uint64_t)pick_field(c[0]) << 20) |
           ((uint64_t)pick_field(c[1]) << 34) | ((uint64_t)pick_field(c[2]) << 48);
}

/* Render the pixels selected by `frame` (same row mapping as svr_frame in
 * include/svr.h).  Output arrays are HOST pointers, out_h*out_w elements
 * (rgba x4); any but rgba may be NULL.  nthreads <= 0: OpenMP default. */
int svr_oracle_render_pick(const svr_camera* cam, const svr_frame* fr,
                           int num_lods, const svr_oracle_lod* lods, const svr_material* mat,
                           float* rgba, float* depth, uint32_t* label, uint8_t* flags, uint32_t* steps,
                           uint64_t* pick, uint32_t pick_id, int nthreads);

int svr_oracle_render(const svr_camera* cam, const svr_frame* fr,
                      int num_lods, const svr_oracle_lod* lods, const svr_material* mat,
                      float* rgba, float* depth, uint32_t* label, uint8_t* flags, uint32_t* steps,
                      int nthreads) {
    return svr_oracle_render_pick(cam, fr, num_lods, lods, mat, rgba, depth, label, flags, steps, NULL, 0u, nthreads);
}

/* same, plus the pick plane (NULL: not wanted) */
int svr_oracle_render_pick(const svr_camera* cam, const svr_frame* fr,
                           int num_lods, const svr_oracle_lod* lods, const svr_material* mat,
                           float* rgba, float* depth, uint32_t* label, uint8_t* flags, uint32_t* steps,
                           uint64_t* pick, uint32_t pick_id, int nthreads) {
    if (!cam || !fr || !lods || !mat || !rgba || num_lods < 1 || mat->color_count == 0) return -1;
    frame_ctx F;
    F.cam = cam; F.mat = mat; F.n = num_lods; F.lods = lods;
    float tmp[16];
    mat_mul(cam->world_inv, cam->cam_inv, tmp);          /* vs_main.wgsl:22, left-assoc */
    mat_mul(tmp, cam->proj_inv, F.ndc_to_data);
    mat_mul(cam->proj, cam->cam, F.pc);                  /* vs_main.wgsl:19 */
    F.sizef = (v3){ cam->volume_dimensions[0], cam->volume_dimensions[1], cam->volume_dimensions[2] };
    /* fs_main.wgsl:20 */
    float mx = fmaxf(F.sizef.x, fmaxf(F.sizef.y, F.sizef.z));
    F.rel_step = fminf(fmaxf(sqrtf(mx) / 20.0f, 0.1f), 0.8f);

    const int band_h = fr->band_h > 0 ? fr->band_h : fr->out_h;
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
#else
    (void)nthreads;
#endif
    for (int r = 0; r < fr->out_h; ++r) {
        int y = fr->y0 + (r / band_h) * fr->band_pitch + (r % band_h);
        for (int c = 0; c < fr->out_w; ++c) {
            int x = fr->x0 + c;
            size_t o = (size_t)r * (size_t)fr->out_w + (size_t)c;
            float px[4], d, hc[3] = { 0.0f, 0.0f, 0.0f }; uint32_t lab, st; int cls = SVR_PIX_DISCARD;
            px[0] = px[1] = px[2] = px[3] = 0.0f; d = 0.0f; lab = 0u; st = 0u;
            if (x >= 0 && x < fr->frame_w && y >= 0 && y < fr->frame_h)
                cls = shade_pixel(&F, fr->frame_w, fr->frame_h, x, y, px, &d, &lab, &st, hc);
            if (pick) pick[o] = cls == SVR_PIX_HIT ? pick_word(pick_id, hc) : 0ull;
            rgba[o * 4 + 0] = px[0]; rgba[o * 4 + 1] = px[1]; rgba[o * 4 + 2] = px[2]; rgba[o * 4 + 3] = px[3];
            if (depth) depth[o] = d;
            if (label) label[o] = lab;
            if (flags) flags[o] = (uint8_t)cls;
            if (steps) steps[o] = st;
        }
    }
    return 0;
}

int svr_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
