// 2x pooling kernels for building LOD pyramids on the GPU: the rules of the
// reference's offline builders — 2x2x2 mean for density (scripts/create_mouse_multiscale.py:23-54),
// 2x2x2 max for labels (scripts/create_platynereis_multiscale.py:86-134).  Pure HBM streaming:
// every source voxel is read once with 8/16-byte loads along x, every output written once.
#include "svr_internal.h"

namespace {

// u8 mean, floor((a0+..+a7)/8): one thread makes 4 consecutive x outputs from 4 rows of 8 source bytes
__global__ __launch_bounds__(256) void pool_mean_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                    int sx, int sy, int sz) {
    const int ox = sx >> 1, oy = sy >> 1, oz = sz >> 1;
    const int qx = (ox + 3) >> 2;                                   // groups of 4 outputs per row
    const size_t n = (size_t)qx * oy * oz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % qx);
        const size_t q = i / qx;
        const int y = (int)(q % oy), z = (int)(q / oy);
        const int x0 = g * 4;
        uint32_t acc[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const uint8_t* row = src + ((size_t)(2 * z + dz) * sy + (size_t)(2 * y + dy)) * sx + 2 * x0;
                if (x0 + 4 <= ox && (((uintptr_t)row) & 7) == 0) {
                    const uint2 v = *reinterpret_cast<const uint2*>(row);
                    acc[0] += (v.x & 0xFF) + ((v.x >> 8) & 0xFF);
                    acc[1] += ((v.x >> 16) & 0xFF) + (v.x >> 24);
                    acc[2] += (v.y & 0xFF) + ((v.y >> 8) & 0xFF);
                    acc[3] += ((v.y >> 16) & 0xFF) + (v.y >> 24);
                } else {
                    for (int k = 0; k < 4 && x0 + k < ox; ++k) acc[k] += row[2 * k] + row[2 * k + 1];
                }
            }
        uint8_t* o = dst + ((size_t)z * oy + y) * ox + x0;
        for (int k = 0; k < 4 && x0 + k < ox; ++k) o[k] = (uint8_t)(acc[k] >> 3);
    }
}

// f32 mean: ((((a000+a001)+(a010+a011)) + ((a100+a101)+(a110+a111))) * 0.125f  (fixed order)
__global__ __launch_bounds__(256) void pool_mean_f32(const float* __restrict__ src, float* __restrict__ dst,
                                                     int sx, int sy, int sz) {
    const int ox = sx >> 1, oy = sy >> 1, oz = sz >> 1;
    const size_t n = (size_t)ox * oy * oz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox);
        const size_t q = i / ox;
        const int y = (int)(q % oy), z = (int)(q / oy);
        float s[2];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
            const float2 a = *reinterpret_cast<const float2*>(src + ((size_t)(2 * z + dz) * sy + (size_t)(2 * y)) * sx + 2 * x);
            const float2 b = *reinterpret_cast<const float2*>(src + ((size_t)(2 * z + dz) * sy + (size_t)(2 * y + 1)) * sx + 2 * x);
            s[dz] = (a.x + a.y) + (b.x + b.y);
        }
        dst[i] = (s[0] + s[1]) * 0.125f;
    }
}

// u32 max
__global__ __launch_bounds__(256) void pool_max_u32(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                    int sx, int sy, int sz) {
    const int ox = sx >> 1, oy = sy >> 1, oz = sz >> 1;
    const size_t n = (size_t)ox * oy * oz;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % ox);
        const size_t q = i / ox;
        const int y = (int)(q % oy), z = (int)(q / oy);
        uint32_t m = 0;
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const uint2 v = *reinterpret_cast<const uint2*>(src + ((size_t)(2 * z + dz) * sy + (size_t)(2 * y + dy)) * sx + 2 * x);
                m = max(m, max(v.x, v.y));
            }
        dst[i] = m;
    }
}

inline int grid_for(size_t n) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace

hipError_t svr_launch_pool2x(const void* src, void* dst, const int32_t dims[3], int dtype, int mode, hipStream_t stream) {
    const int sx = dims[0], sy = dims[1], sz = dims[2];
    const size_t n = (size_t)(sx >> 1) * (sy >> 1) * (sz >> 1);
    if (n == 0) return hipSuccess;
    if (dtype == SVR_U8 && mode == 0)
        hipLaunchKernelGGL(pool_mean_u8, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), sx, sy, sz);
    else if (dtype == SVR_F32 && mode == 0)
        hipLaunchKernelGGL(pool_mean_f32, dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const float*>(src), static_cast<float*>(dst), sx, sy, sz);
    else if (dtype == SVR_U32 && mode == 1)
        hipLaunchKernelGGL(pool_max_u32, dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const uint32_t*>(src), static_cast<uint32_t*>(dst), sx, sy, sz);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
