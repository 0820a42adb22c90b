// LMIP ray-march for gfx950 (MI355X): vs_main + fs_main + raycast + sample_vol +
// hsv_selection of the reference's WGSL (src/sub_volume/shaders/*.wgsl) as ONE
// HIP kernel.  One lane = one pixel = one fs_main invocation; one wave64 = an
// 8x8 pixel tile, so the 64 rays of a wave traverse neighbouring voxels.
//
// Arithmetic contract (must stay bit-identical with oracle/lmip_oracle.c):
// strict IEEE f32, NO fp contraction (-ffp-contract=off), expression order as
// written in the WGSL; see DESIGN.md "operation-order contract".
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
        if (lod_texel(P.lod[l], dx, dy, dz, idx)) return P.lod[l].density[idx];
    }
    return 0.0f;
}

template <int NL>
__device__ __forceinline__ uint32_t sample_label(const MarchParams& P, float cx, float cy, float cz) {
    float dx = cx * P.size[0], dy = cy * P.size[1], dz = cz * P.size[2];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        size_t idx;
        if (lod_texel(P.lod[l], dx, dy, dz, idx)) return P.lod[l].labels[idx];
    }
    return 0u;
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
            v = powf(v, P.gamma);
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
}

// Block -> 16x16 pixel tile.  Workgroups are dealt round-robin to the 8 XCDs
// (blockIdx b and b+8 share an XCD and its 4 MiB L2), so give each XCD a
// contiguous run of tiles: neighbouring tiles sample neighbouring voxels.
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
    const int t = xcd_remap((int)blockIdx.x, nblocks);
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
// Variant 0 (default): batched march.
//
// Exactness argument.  For one ray and one axis, the voxel index the reference
// computes at iteration i,
//     ic(i) = i32( ((start + f32(i)*step) * size) * scale )        (raycast.wgsl:30-31,
//                                                                   sample_vol.wgsl:6-8,17)
// is a composition of monotone functions of i (IEEE rounding is monotone), so it is
// monotone in i.  Hence "voxel i lies in LOD l's ROI" holds on ONE contiguous
// iteration interval [A_l, B_l), whose ends are found exactly by evaluating that
// same f32 chain (first_true below).  The cascade of sample_vol.wgsl:51-63 then is
// "smallest l with A_l <= i < B_l".  With the intervals known, a wave whose lanes
// are all inside the same LOD for a whole batch of U steps needs no bounds tests,
// computes U texel offsets, issues U independent buffer loads (hardware range
// check returns 0 for "no LOD"), and only then runs the sequential LMIP state
// machine — skipped entirely while no lane has reached the threshold.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int axis_voxel(int i, float start, float step, float size, float scale) {
    float off = (float)i * step;          // raycast.wgsl:30  (iter is an exact integer-valued f32)
    float coord = start + off;            // :31
    float d = coord * size;               // sample_vol.wgsl:6
    float sd = d * scale;                 // :7-8
    return (int)sd;                       // :17 vec3<i32>()
}

// smallest i in [0, n] with pred(i), for a monotone (false.. true..) predicate
template <class Pred>
__device__ __forceinline__ int first_true(float guess, int n, Pred pred) {
    int i = (int)fminf(fmaxf(guess, 0.0f), (float)n);
    while (i > 0 && pred(i - 1)) --i;
    while (i < n && !pred(i)) ++i;
    return i;
}

// iterations for which lo <= ic(i) < hi on one axis: [enter, exit)
__device__ __forceinline__ void axis_interval(int n, float start, float step, float size, float scale,
                                              int lo, int hi, int& enter, int& exit) {
    auto g = [&](int i) { return axis_voxel(i, start, step, size, scale); };
    if (!(step > 0.0f) && !(step < 0.0f)) {           // zero (or NaN) step: constant along the ray
        const int v = g(0);
        const bool in = lo <= v && v < hi;
        enter = 0; exit = in ? n : 0;
        return;
    }
    const float k = size * scale;
    const float glo = ceilf(((float)lo / k - start) / step);
    const float ghi = ceilf(((float)hi / k - start) / step);
    if (step > 0.0f) {
        enter = first_true(glo, n, [&](int i) { return g(i) >= lo; });
        exit  = first_true(ghi, n, [&](int i) { return g(i) >= hi; });
    } else {
        enter = first_true(ghi, n, [&](int i) { return g(i) < hi; });
        exit  = first_true(glo, n, [&](int i) { return g(i) < lo; });
    }
}

// byte offset of the texel under data coord d inside MarchParams::density_all, for a
// voxel KNOWN to lie in LOD L's ROI (no bounds test)
__device__ __forceinline__ uint32_t lod_offset_inside(const LodParams& L, float dx, float dy, float dz) {
    float sx = dx * L.scale[0], sy = dy * L.scale[1], sz = dz * L.scale[2];
    uint32_t wx = (uint32_t)((int)sx + L.addw[0]);
    uint32_t wy = (uint32_t)((int)sy + L.addw[1]);
    uint32_t wz = (uint32_t)((int)sz + L.addw[2]);
    wx = min(wx, wx - L.ring[0]);
    wy = min(wy, wy - L.ring[1]);
    wz = min(wz, wz - L.ring[2]);
    return (wz * L.ring[1] + wy) * L.rx4 + L.base_bytes + (wx << 2);
}

template <int NL, int U, bool COUNT>
__global__ __launch_bounds__(256) void march_batched(const MarchParams P) {
    const int nblocks = P.tiles_x * P.tiles_y;
    const int t = xcd_remap((int)blockIdx.x, nblocks);
    const int tile_x = t % P.tiles_x, tile_y = t / P.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    const int r = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (c >= P.frame.out_w || r >= P.frame.out_h) return;
    const size_t o = (size_t)r * (size_t)P.frame.out_w + (size_t)c;
    const int x = P.frame.x0 + c;
    const int y = P.frame.y0 + (r / P.frame.band_h) * P.frame.band_pitch + (r % P.frame.band_h);

    Ray R;
    R.nsteps = 0; R.start = { 0.f, 0.f, 0.f }; R.step = { 0.f, 0.f, 0.f };
    const bool frag = (x < P.frame.frame_w && y < P.frame.frame_h) && setup_ray(P, x, y, R);
    const int nsteps = frag ? R.nsteps : 0;

    // exact per-LOD iteration intervals
    int A[NL], B[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        A[l] = 0; B[l] = 0;
        if (frag) {
            const LodParams& L = P.lod[l];
            int e0, x0, e1, x1, e2, x2;
            axis_interval(nsteps, R.start.x, R.step.x, P.size[0], L.scale[0], L.off[0], L.off[0] + (int)L.shape[0], e0, x0);
            axis_interval(nsteps, R.start.y, R.step.y, P.size[1], L.scale[1], L.off[1], L.off[1] + (int)L.shape[1], e1, x1);
            axis_interval(nsteps, R.start.z, R.step.z, P.size[2], L.scale[2], L.off[2], L.off[2] + (int)L.shape[2], e2, x2);
            const int a = max(e0, max(e1, e2)), b = min(x0, min(x1, x2));
            if (a < b) { A[l] = a; B[l] = b; }
        }
    }

    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(P.density_all), 0, (int)P.density_all_bytes, 0x00020000);

    bool found = false, finished = false;
    float local_max = 0.f, samp = 0.f;
    int hit_i = 0, since = 0;
    uint32_t steps = 0;

    for (int i = 0;; i += U) {
        const bool active = !finished && i < nsteps;
        if (__builtin_amdgcn_ballot_w64(active) == 0) break;
        float s[U];
        bool valid[U];
        if (active) {
            const int iend = min(i + U, nsteps);
            // which LOD serves this whole batch for this lane?  NL = "none", -1 = mixed
            int code = NL;
            bool settled = false;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const bool inb = i >= A[l] && iend <= B[l];
                const bool outb = iend <= A[l] || i >= B[l];
                if (!settled) {
                    if (inb) { code = l; settled = true; }
                    else if (!outb) { code = -1; settled = true; }
                }
            }
            const int first = __builtin_amdgcn_readfirstlane(code);
            const bool uniform = first >= 0 && __builtin_amdgcn_ballot_w64(code != first) == 0;
            uint32_t off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { off[u] = 0xFFFFFFFFu; valid[u] = (i + u) < nsteps; }
            const float basef = (float)i;
            if (uniform) {
                if (first < NL) {
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        if (first == l) {
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const float iter = basef + (float)u;
                                const float cx = R.start.x + iter * R.step.x;
                                const float cy = R.start.y + iter * R.step.y;
                                const float cz = R.start.z + iter * R.step.z;
                                off[u] = lod_offset_inside(P.lod[l], cx * P.size[0], cy * P.size[1], cz * P.size[2]);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float iter = basef + (float)u;
                    const float dx = (R.start.x + iter * R.step.x) * P.size[0];
                    const float dy = (R.start.y + iter * R.step.y) * P.size[1];
                    const float dz = (R.start.z + iter * R.step.z) * P.size[2];
                    bool done = false;
#pragma unroll
                    for (int l = 0; l < NL; ++l) {
                        const bool sel = !done && (i + u) >= A[l] && (i + u) < B[l];
                        if (__builtin_amdgcn_ballot_w64(sel) != 0) {
                            const uint32_t ofs = lod_offset_inside(P.lod[l], dx, dy, dz);
                            off[u] = sel ? ofs : off[u];
                        }
                        done = done || sel;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                s[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off[u], 0, 0));
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { s[u] = 0.f; valid[u] = false; }
        }

        // LMIP state machine (raycast.wgsl:35-61); skipped while nothing can change
        float m = -1.0f;
#pragma unroll
        for (int u = 0; u < U; ++u) m = fmaxf(m, valid[u] ? fabsf(s[u]) : -1.0f);
        const bool need = active && (found || m >= P.lmip_threshold);
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool act = active && valid[u] && !finished;
                const float inten = fabsf(s[u]);
                if (COUNT) steps += act ? 1u : 0u;
                const bool was_found = found;
                const bool first_hit = act && !was_found && inten >= P.lmip_threshold;     // :37
                const bool tracking = act && was_found;
                since += tracking ? 1 : 0;                                                 // :47
                const bool take = first_hit || (tracking && inten > local_max);            // :50
                local_max = take ? inten : local_max;
                samp = take ? s[u] : samp;
                hit_i = take ? (i + u) : hit_i;
                found = found || first_hit;
                const bool brk = tracking && (since >= P.lmip_max_samples || inten < local_max * P.lmip_fall_off);  // :58
                finished = finished || brk;
            }
        } else if (COUNT) {
            steps += active ? (uint32_t)(min(i + U, nsteps) - i) : 0u;
        }
    }

    Hit h;
    h.found = found; h.sample = samp; h.steps = steps;
    const float hit_f = (float)hit_i;
    h.offset = { hit_f * R.step.x, hit_f * R.step.y, hit_f * R.step.z };                    // raycast.wgsl:30
    h.coord = { R.start.x + h.offset.x, R.start.y + h.offset.y, R.start.z + h.offset.z };   // :31
    shade_and_store<NL>(P, o, frag, h);
    if (COUNT && P.steps) P.steps[o] = h.steps;
}

template <int NL>
hipError_t launch_nl(const MarchParams& p, int variant, hipStream_t stream) {
    const int nblocks = p.tiles_x * p.tiles_y;
    if (nblocks <= 0) return hipSuccess;
    const bool simple = variant == 1 || p.density_all_bytes == 0;   // >= 4 GiB of rings: 64-bit addressing
    if (simple) {
        if (p.steps) hipLaunchKernelGGL((march_simple<NL, true>), dim3(nblocks), dim3(256), 0, stream, p);
        else         hipLaunchKernelGGL((march_simple<NL, false>), dim3(nblocks), dim3(256), 0, stream, p);
    } else if (variant == 2) {
        if (p.steps) hipLaunchKernelGGL((march_batched<NL, 4, true>), dim3(nblocks), dim3(256), 0, stream, p);
        else         hipLaunchKernelGGL((march_batched<NL, 4, false>), dim3(nblocks), dim3(256), 0, stream, p);
    } else {
        if (p.steps) hipLaunchKernelGGL((march_batched<NL, 8, true>), dim3(nblocks), dim3(256), 0, stream, p);
        else         hipLaunchKernelGGL((march_batched<NL, 8, false>), dim3(nblocks), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t svr_launch_march(const MarchParams& p, int variant, hipStream_t stream) {
    switch (p.num_lods) {
        case 1: return launch_nl<1>(p, variant, stream);
        case 2: return launch_nl<2>(p, variant, stream);
        case 3: return launch_nl<3>(p, variant, stream);
        case 4: return launch_nl<4>(p, variant, stream);
        case 5: return launch_nl<5>(p, variant, stream);
        case 6: return launch_nl<6>(p, variant, stream);
        case 7: return launch_nl<7>(p, variant, stream);
        case 8: return launch_nl<8>(p, variant, stream);
        default: return hipErrorInvalidValue;
    }
}
