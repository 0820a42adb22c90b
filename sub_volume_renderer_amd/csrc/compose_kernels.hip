// Display-side output: what a canvas shows for this object over a background.
// The fragment outputs of the march (out.color before blending, out.depth; fs_main.wgsl:86-87) are blended
// "over" a vertical-gradient background (the reference's tests add gfx.Background(None,
// BackgroundMaterial(bottom, top)), tests/conftest.py:17-22), depth-tested against an optional existing depth
// plane, encoded linear -> sRGB and quantised to 8 bits.  Blending and the final encode are pygfx's, restated
// (parity unpinned).  Pure HBM streaming: 21 B read + 4 B written per pixel, one thread per pixel.
#include <algorithm>

#include "svr_internal.h"

namespace {

__device__ __forceinline__ float srgb_encode(float c) {          // IEC 61966-2-1 OETF
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}

__device__ __forceinline__ uint32_t quant8(float v) {
    return (uint32_t)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f);
}

__global__ __launch_bounds__(256) void compose_kernel(const float4* __restrict__ rgba, const float* __restrict__ depth,
                                                      const uint8_t* __restrict__ flags, int w, int h,
                                                      svr_compose_params q, uint32_t* __restrict__ out8,
                                                      float* __restrict__ zbuf) {
    const size_t n = (size_t)w * (size_t)h;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / (size_t)w);
        const float t = ((float)y + 0.5f) / (float)h;              // 0 at the top row
        float c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = q.bg_top[k] * (1.0f - t) + q.bg_bottom[k] * t;
        bool draw = flags ? flags[i] != SVR_PIX_DISCARD : true;
        if (draw && zbuf && depth) draw = depth[i] < zbuf[i];      // depth_compare "<"
        if (draw) {
            const float4 s = rgba[i];
            const float a = s.w;
            c[0] = s.x * a + c[0] * (1.0f - a);                    // src_alpha, one_minus_src_alpha
            c[1] = s.y * a + c[1] * (1.0f - a);
            c[2] = s.z * a + c[2] * (1.0f - a);
            c[3] = a + c[3] * (1.0f - a);
            if (zbuf && depth) zbuf[i] = depth[i];
        }
        if (q.srgb_encode) { c[0] = srgb_encode(c[0]); c[1] = srgb_encode(c[1]); c[2] = srgb_encode(c[2]); }
        out8[i] = quant8(c[0]) | (quant8(c[1]) << 8) | (quant8(c[2]) << 16) | (quant8(c[3]) << 24);
    }
}

}  // namespace

hipError_t svr_launch_compose(const float* rgba, const float* depth, const uint8_t* flags, int w, int h,
                              const svr_compose_params& q, uint8_t* out_rgba8, float* zbuf, hipStream_t stream) {
    const size_t n = (size_t)w * (size_t)h;
    if (n == 0) return hipSuccess;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(compose_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(rgba), depth,
                       flags, w, h, q, reinterpret_cast<uint32_t*>(out_rgba8), zbuf);
    return hipGetLastError();
}
