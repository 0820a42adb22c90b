// Ring-buffer data movement kernels for gfx950: dtype-converting scatter of a
// chunk slab into the ring textures (the device half of
// `texture.data[dst] = np.array(src, f32|u32)`, _wrapping_buffer.py:325-335),
// the read-back gather, and the stripe un-tiler used after the RCCL gather.
// All three are pure HBM streaming; rows (x) are the contiguous axis.
#include <stdlib.h>

#include "svr_internal.h"

size_t svr_dtype_size(int dtype) {
    switch (dtype) {
        case SVR_U8: case SVR_I8: return 1;
        case SVR_U16: case SVR_I16: return 2;
        case SVR_U32: case SVR_I32: case SVR_F32: return 4;
        case SVR_U64: case SVR_I64: case SVR_F64: return 8;
        default: return 0;
    }
}

namespace {

// numpy cast semantics: any -> float32
__device__ __forceinline__ float load_as_f32(const char* p, int dtype) {
    switch (dtype) {
        case SVR_U8:  return (float)*reinterpret_cast<const uint8_t*>(p);
        case SVR_U16: return (float)*reinterpret_cast<const uint16_t*>(p);
        case SVR_U32: return (float)*reinterpret_cast<const uint32_t*>(p);
        case SVR_U64: return (float)*reinterpret_cast<const uint64_t*>(p);
        case SVR_I8:  return (float)*reinterpret_cast<const int8_t*>(p);
        case SVR_I16: return (float)*reinterpret_cast<const int16_t*>(p);
        case SVR_I32: return (float)*reinterpret_cast<const int32_t*>(p);
        case SVR_I64: return (float)*reinterpret_cast<const int64_t*>(p);
        case SVR_F32: return *reinterpret_cast<const float*>(p);
        default:      return (float)*reinterpret_cast<const double*>(p);
    }
}

// numpy cast semantics: integer -> uint32 wraps modulo 2^32; float -> uint32
// truncates toward zero (values outside the range are undefined in numpy too)
__device__ __forceinline__ uint32_t load_as_u32(const char* p, int dtype) {
    switch (dtype) {
        case SVR_U8:  return (uint32_t)*reinterpret_cast<const uint8_t*>(p);
        case SVR_U16: return (uint32_t)*reinterpret_cast<const uint16_t*>(p);
        case SVR_U32: return *reinterpret_cast<const uint32_t*>(p);
        case SVR_U64: return (uint32_t)*reinterpret_cast<const uint64_t*>(p);
        case SVR_I8:  return (uint32_t)(int32_t)*reinterpret_cast<const int8_t*>(p);
        case SVR_I16: return (uint32_t)(int32_t)*reinterpret_cast<const int16_t*>(p);
        case SVR_I32: return (uint32_t)*reinterpret_cast<const int32_t*>(p);
        case SVR_I64: return (uint32_t)*reinterpret_cast<const int64_t*>(p);
        case SVR_F32: return (uint32_t)(int64_t)*reinterpret_cast<const float*>(p);
        default:      return (uint32_t)(int64_t)*reinterpret_cast<const double*>(p);
    }
}

// General path: one thread per voxel, x fastest, any source dtype and strides (numpy cast semantics).
__global__ __launch_bounds__(256) void scatter_kernel(const ScatterArgs a) {
    const size_t n = (size_t)a.shape[0] * (size_t)a.shape[1] * (size_t)a.shape[2];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t x = (uint32_t)(i % (size_t)a.shape[0]);
        const size_t   q = i / (size_t)a.shape[0];
        const uint32_t y = (uint32_t)(q % (size_t)a.shape[1]);
        const uint32_t z = (uint32_t)(q / (size_t)a.shape[1]);
        const size_t dst = ((size_t)(z + a.dst_off[2]) * (size_t)a.ring[1] + (size_t)(y + a.dst_off[1])) *
                               (size_t)a.ring[0] + (size_t)(x + a.dst_off[0]);
        if (a.src_density) {
            const char* p = static_cast<const char*>(a.src_density) +
                            (int64_t)x * a.dstride[0] + (int64_t)y * a.dstride[1] + (int64_t)z * a.dstride[2];
            // (the micro-block copy of the ring, where the LOD keeps one, gets the same element)
            const int esh = a.ring_storage == SVR_U8 ? 0 : (a.ring_storage == SVR_U16 ? 1 : 2);
            const size_t tw = a.ring_twin ? svr_blocked_index(esh, (uint32_t)a.ring[0], (uint32_t)a.ring[1], x + (uint32_t)a.dst_off[0],
                                                              y + (uint32_t)a.dst_off[1], z + (uint32_t)a.dst_off[2]) : 0;
            if (a.ring_storage == SVR_U8) {
                const uint8_t v = *reinterpret_cast<const uint8_t*>(p);
                static_cast<uint8_t*>(a.ring_density)[dst] = v;
                if (a.ring_twin) static_cast<uint8_t*>(a.ring_twin)[tw] = v;
            } else if (a.ring_storage == SVR_U16) {
                const uint16_t v = *reinterpret_cast<const uint16_t*>(p);
                static_cast<uint16_t*>(a.ring_density)[dst] = v;
                if (a.ring_twin) static_cast<uint16_t*>(a.ring_twin)[tw] = v;
            } else {
                const float v = load_as_f32(p, a.density_dtype);
                static_cast<float*>(a.ring_density)[dst] = v;
                if (a.ring_twin) static_cast<float*>(a.ring_twin)[tw] = v;
            }
        }
        if (a.src_labels) {
            const char* p = static_cast<const char*>(a.src_labels) +
                            (int64_t)x * a.lstride[0] + (int64_t)y * a.lstride[1] + (int64_t)z * a.lstride[2];
            a.ring_labels[dst] = load_as_u32(p, a.labels_dtype);
        }
    }
}

// Streaming path (the staged blocks of svr_upload_region and contiguous device sources): the source
// rows are packed, already hold the ring's element types, and rows start and end on 16-voxel groups on
// both sides.  One thread moves one group of 16 voxels with 16-byte loads and stores: DES bytes per
// density element (1 / 2 / 4: 1 / 2 / 4 transfers) and 4 transfers for the u32 labels; no per-voxel
// index arithmetic, no conversion.  Pure HBM streaming: 2 * 16 * (DES + 4) bytes per thread.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DES>
__global__ __launch_bounds__(256) void scatter_rows16(const ScatterArgs a, uint32_t groups_per_row, uint32_t total) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const uint32_t g = i % groups_per_row, row = i / groups_per_row;
    const uint32_t y = row % (uint32_t)a.shape[1], z = row / (uint32_t)a.shape[1];
    const size_t dst = ((size_t)(z + a.dst_off[2]) * (size_t)a.ring[1] + (size_t)(y + a.dst_off[1])) *
                           (size_t)a.ring[0] + (size_t)(g * 16u + a.dst_off[0]);
    if (a.src_density) {
        const u32x4* s = reinterpret_cast<const u32x4*>(static_cast<const char*>(a.src_density) +
                                                        (int64_t)y * a.dstride[1] + (int64_t)z * a.dstride[2]) + (size_t)g * DES;
        u32x4* d = reinterpret_cast<u32x4*>(static_cast<char*>(a.ring_density) + dst * DES);
        u32x4 v[DES];
#pragma unroll
        for (int k = 0; k < DES; ++k) v[k] = __builtin_nontemporal_load(s + k);      // staged bytes are read once
#pragma unroll
        for (int k = 0; k < DES; ++k) d[k] = v[k];
        if (a.ring_twin) {
            // the same 16 voxels into the micro-block copy: they start a block row (x is a multiple of 16) and are the x rows
            // of 2 (1-byte voxels: 8 per row) or 4 (2- / 4-byte: 4 per row) consecutive blocks, 128 bytes apart.  The threads of
            // the 4 y rows of a block run side by side in time, so its 32- / 64-byte z slices are written whole.
            constexpr int ESH = DES == 1 ? 0 : (DES == 2 ? 1 : 2);
            char* t = static_cast<char*>(a.ring_twin) +
                      svr_blocked_index(ESH, (uint32_t)a.ring[0], (uint32_t)a.ring[1], g * 16u + (uint32_t)a.dst_off[0],
                                        y + (uint32_t)a.dst_off[1], z + (uint32_t)a.dst_off[2]) * DES;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            if constexpr (DES == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(t + k * 128) = v[k];
            } else {
#pragma unroll
                for (int k = 0; k < DES; ++k) {
                    u32x2 lo = { v[k].x, v[k].y }, hi = { v[k].z, v[k].w };
                    *reinterpret_cast<u32x2*>(t + (2 * k) * 128) = lo;
                    *reinterpret_cast<u32x2*>(t + (2 * k + 1) * 128) = hi;
                }
            }
        }
    }
    if (a.src_labels) {
        const u32x4* s = reinterpret_cast<const u32x4*>(static_cast<const char*>(a.src_labels) +
                                                        (int64_t)y * a.lstride[1] + (int64_t)z * a.lstride[2]) + (size_t)g * 4;
        u32x4* d = reinterpret_cast<u32x4*>(a.ring_labels + dst);
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(s + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = v[k];
    }
}

// ---------------------------------------------------------------------------------------------------
// Macro-cell maxima for empty-space skipping (march_kernel.hip, LMIP mode).  Per LOD, the ring is cut into
// cells of S^3 slots (S = 8 or 4, by LOD); `raw` holds the largest |value| stored in a cell, `blk` the largest
// raw value of the 2 x 2 x 2 block of cells that STARTS at the cell (on the ring torus).  Both are maintained
// here, on the upload stream, right behind every scatter: the march reads only `blk`.  Conservative by
// construction: a cell's maximum covers every slot of the cell, whether the published ROI maps it or not.
// ---------------------------------------------------------------------------------------------------
struct CellArgs {
    const void* ring; void* raw; void* blk;
    int32_t ring_dims[3];      // slots (x, y, z), each a multiple of the cell size
    int32_t cdim[3];           // cells per axis
    int32_t cshift;            // log2 of the cell size
    int32_t c0[3], cn[3];      // cell range to refresh (c0 may be negative / run past cdim: wraps on the torus)
};

template <typename T> __device__ __forceinline__ T cell_abs(T v) { return v; }
template <> __device__ __forceinline__ float cell_abs<float>(float v) { return fabsf(v); }
template <typename T> __device__ __forceinline__ T cell_max(T a, T b) { return a > b ? a : b; }
template <> __device__ __forceinline__ float cell_max<float>(float a, float b) { return fmaxf(a, b); }   // NaN never wins

__device__ __forceinline__ int torus(int c, int n) { c %= n; return c < 0 ? c + n : c; }

// One thread per cell, S * S rows of S slots each; neighbouring threads take neighbouring cells along x, so a
// wave's loads of one row index cover 64 * S contiguous slots of a ring row: coalesced, every byte read once.
template <typename T, int S>
__global__ __launch_bounds__(256) void cell_raw_kernel(const CellArgs a) {
    const int cell = (int)(blockIdx.x * 256u + threadIdx.x);
    const int total = a.cn[0] * a.cn[1] * a.cn[2];
    if (cell >= total) return;
    const int cx = torus(a.c0[0] + cell % a.cn[0], a.cdim[0]);
    const int cy = torus(a.c0[1] + (cell / a.cn[0]) % a.cn[1], a.cdim[1]);
    const int cz = torus(a.c0[2] + cell / (a.cn[0] * a.cn[1]), a.cdim[2]);
    typedef T row_t __attribute__((ext_vector_type(S)));            // one row of the cell: S * sizeof(T) bytes, aligned
    T m = 0;
    for (int z = 0; z < S; ++z) {
        const T* plane = static_cast<const T*>(a.ring) +
                         ((size_t)(cz * S + z) * (size_t)a.ring_dims[1] + (size_t)cy * S) * (size_t)a.ring_dims[0] + (size_t)cx * S;
#pragma unroll
        for (int y = 0; y < S; ++y) {
            const row_t r = *reinterpret_cast<const row_t*>(plane + (size_t)y * (size_t)a.ring_dims[0]);
#pragma unroll
            for (int k = 0; k < S; ++k) m = cell_max(m, cell_abs((T)r[k]));
        }
    }
    static_cast<T*>(a.raw)[((size_t)cz * a.cdim[1] + cy) * a.cdim[0] + cx] = m;
}

// one thread per cell of the range: maximum over the 2 x 2 x 2 raw cells starting at the cell, on the torus
template <typename T>
__global__ __launch_bounds__(256) void cell_block_kernel(const CellArgs a) {
    const int cell = (int)(blockIdx.x * 256u + threadIdx.x);
    const int total = a.cn[0] * a.cn[1] * a.cn[2];
    if (cell >= total) return;
    const int cx = torus(a.c0[0] + cell % a.cn[0], a.cdim[0]);
    const int cy = torus(a.c0[1] + (cell / a.cn[0]) % a.cn[1], a.cdim[1]);
    const int cz = torus(a.c0[2] + cell / (a.cn[0] * a.cn[1]), a.cdim[2]);
    const T* raw = static_cast<const T*>(a.raw);
    T m = raw[((size_t)cz * a.cdim[1] + cy) * a.cdim[0] + cx];
    for (int dz = 0; dz <= 1; ++dz)
        for (int dy = 0; dy <= 1; ++dy)
            for (int dx = 0; dx <= 1; ++dx) {
                const int x = torus(cx + dx, a.cdim[0]), y = torus(cy + dy, a.cdim[1]), z = torus(cz + dz, a.cdim[2]);
                m = cell_max(m, raw[((size_t)z * a.cdim[1] + y) * a.cdim[0] + x]);
            }
    static_cast<T*>(a.blk)[((size_t)cz * a.cdim[1] + cy) * a.cdim[0] + cx] = m;
}

struct GatherArgs {
    const void* ring_density; int32_t ring_storage; const uint32_t* ring_labels;
    int32_t ring[3], off[3], shape[3];
    float* out_density; uint32_t* out_labels;
};

__global__ __launch_bounds__(256) void gather_kernel(const GatherArgs a) {
    const size_t n = (size_t)a.shape[0] * (size_t)a.shape[1] * (size_t)a.shape[2];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t x = (uint32_t)(i % (size_t)a.shape[0]);
        const size_t   q = i / (size_t)a.shape[0];
        const uint32_t y = (uint32_t)(q % (size_t)a.shape[1]);
        const uint32_t z = (uint32_t)(q / (size_t)a.shape[1]);
        const size_t src = ((size_t)(z + a.off[2]) * (size_t)a.ring[1] + (size_t)(y + a.off[1])) *
                               (size_t)a.ring[0] + (size_t)(x + a.off[0]);
        if (a.out_density)
            a.out_density[i] = a.ring_storage == SVR_U8    ? (float)static_cast<const uint8_t*>(a.ring_density)[src]
                               : a.ring_storage == SVR_U16 ? (float)static_cast<const uint16_t*>(a.ring_density)[src]
                                                           : static_cast<const float*>(a.ring_density)[src];
        if (a.out_labels)  a.out_labels[i]  = a.ring_labels[src];
    }
}

// gathered: [nranks][out_h][frame_w] elements of elem_bytes (4 or 16); rank k's
// row r is frame row (r / band_h) * band_h * nranks + k * band_h + r % band_h.
template <typename T>
__global__ __launch_bounds__(256) void untile_kernel(const T* gathered, T* frame, int frame_w, int frame_h,
                                                     int band_h, int nranks, int out_h) {
    const size_t n = (size_t)frame_w * (size_t)frame_h;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int x = (int)(i % (size_t)frame_w);
        const int y = (int)(i / (size_t)frame_w);
        const int band = y / band_h, in_band = y % band_h;
        const int rank = band % nranks, k = band / nranks;
        const size_t r = (size_t)k * band_h + in_band;
        frame[i] = gathered[((size_t)rank * out_h + r) * (size_t)frame_w + x];
    }
}

// gathered: [grid_y * grid_x][tile_h][tile_w]; frame pixel (x, y) lives in tile (x / tile_w, y / tile_h)
template <typename T>
__global__ __launch_bounds__(256) void untile_grid_kernel(const T* gathered, T* frame, int frame_w, int frame_h,
                                                          int tile_w, int tile_h, int grid_x) {
    const size_t n = (size_t)frame_w * (size_t)frame_h;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int x = (int)(i % (size_t)frame_w), y = (int)(i / (size_t)frame_w);
        const int tx = x / tile_w, ty = y / tile_h;
        const size_t rank = (size_t)ty * grid_x + tx;
        frame[i] = gathered[(rank * tile_h + (size_t)(y - ty * tile_h)) * (size_t)tile_w + (size_t)(x - tx * tile_w)];
    }
}

inline int grid_for(size_t n) {
    size_t blocks = (n + 255) / 256;
    const size_t cap = 256 * 8;          // 256 CUs x 8 blocks, grid-stride the rest
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

hipError_t svr_launch_scatter(const ScatterArgs& a, hipStream_t stream) {
    const size_t n = (size_t)a.shape[0] * (size_t)a.shape[1] * (size_t)a.shape[2];
    if (n == 0) return hipSuccess;
    // streaming path: element types already those of the ring, rows packed and cut on 16-voxel groups
    const int des = (int)svr_dtype_size(a.ring_storage);
    const bool d_ok = !a.src_density ||
        (a.density_dtype == a.ring_storage && a.dstride[0] == des && aligned16(a.src_density) &&
         a.dstride[1] % 16 == 0 && a.dstride[2] % 16 == 0);
    const bool l_ok = !a.src_labels ||
        ((a.labels_dtype == SVR_U32 || a.labels_dtype == SVR_I32) && a.lstride[0] == 4 && aligned16(a.src_labels) &&
         a.lstride[1] % 16 == 0 && a.lstride[2] % 16 == 0);
    const bool geo_ok = (a.shape[0] & 15) == 0 && (a.dst_off[0] & 15) == 0 && (a.ring[0] & 15) == 0 &&
                        aligned16(a.ring_density) && aligned16(a.ring_labels) && n / 16 < 0x7fffffffu;
    static const bool force_general = svr_exp_env_set("SVR_SCATTER_GENERAL");        // A/B measurements (-DSVR_EXPERIMENTS builds)
    if (d_ok && l_ok && geo_ok && !force_general) {
        const uint32_t gpr = (uint32_t)a.shape[0] / 16u, total = (uint32_t)(n / 16);
        const dim3 grid((total + 255u) / 256u), block(256);
        if (des == 1)      hipLaunchKernelGGL((scatter_rows16<1>), grid, block, 0, stream, a, gpr, total);
        else if (des == 2) hipLaunchKernelGGL((scatter_rows16<2>), grid, block, 0, stream, a, gpr, total);
        else               hipLaunchKernelGGL((scatter_rows16<4>), grid, block, 0, stream, a, gpr, total);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t svr_launch_gather(const void* ring_density, int ring_storage, const uint32_t* ring_labels, const int32_t ring[3],
                             const int32_t off[3], const int32_t shape[3],
                             float* out_density, uint32_t* out_labels, hipStream_t stream) {
    GatherArgs a;
    a.ring_density = ring_density; a.ring_storage = ring_storage; a.ring_labels = ring_labels;
    for (int i = 0; i < 3; ++i) { a.ring[i] = ring[i]; a.off[i] = off[i]; a.shape[i] = shape[i]; }
    a.out_density = out_density; a.out_labels = out_labels;
    const size_t n = (size_t)shape[0] * (size_t)shape[1] * (size_t)shape[2];
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t svr_launch_untile(const void* gathered, void* frame_out, int frame_w, int frame_h,
                             int band_h, int nranks, int out_h, int elem_bytes, hipStream_t stream) {
    const size_t n = (size_t)frame_w * (size_t)frame_h;
    if (n == 0) return hipSuccess;
    if (elem_bytes == 16)
        hipLaunchKernelGGL((untile_kernel<float4>), dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const float4*>(gathered), static_cast<float4*>(frame_out),
                           frame_w, frame_h, band_h, nranks, out_h);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL((untile_kernel<uint32_t>), dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const uint32_t*>(gathered), static_cast<uint32_t*>(frame_out),
                           frame_w, frame_h, band_h, nranks, out_h);
    else if (elem_bytes == 1)
        hipLaunchKernelGGL((untile_kernel<uint8_t>), dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(gathered), static_cast<uint8_t*>(frame_out),
                           frame_w, frame_h, band_h, nranks, out_h);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t svr_launch_untile_grid(const void* gathered, void* frame_out, int frame_w, int frame_h,
                                  int tile_w, int tile_h, int grid_x, int grid_y, int elem_bytes, hipStream_t stream) {
    const size_t n = (size_t)frame_w * (size_t)frame_h;
    if (n == 0) return hipSuccess;
    (void)grid_y;
    if (elem_bytes == 16)
        hipLaunchKernelGGL((untile_grid_kernel<float4>), dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const float4*>(gathered), static_cast<float4*>(frame_out), frame_w, frame_h, tile_w, tile_h, grid_x);
    else if (elem_bytes == 4)
        hipLaunchKernelGGL((untile_grid_kernel<uint32_t>), dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const uint32_t*>(gathered), static_cast<uint32_t*>(frame_out), frame_w, frame_h, tile_w, tile_h, grid_x);
    else if (elem_bytes == 1)
        hipLaunchKernelGGL((untile_grid_kernel<uint8_t>), dim3(grid_for(n)), dim3(256), 0, stream,
                           static_cast<const uint8_t*>(gathered), static_cast<uint8_t*>(frame_out), frame_w, frame_h, tile_w, tile_h, grid_x);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// Refresh the macro-cell maxima of the cells a scatter into [off, off + shape) of the ring has touched (raw), and
// of the 2 x 2 x 2 blocks those cells belong to.  No-op for LODs without a cell grid.
hipError_t svr_launch_cell_update(const void* ring, int storage, const int32_t ring_dims[3], void* raw, void* blk,
                                  const int32_t cdim[3], int cshift, const int32_t off[3], const int32_t shape[3],
                                  hipStream_t stream) {
    if (!raw || !blk || shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0) return hipSuccess;
    CellArgs a;
    a.ring = ring; a.raw = raw; a.blk = blk; a.cshift = cshift;
    for (int i = 0; i < 3; ++i) {
        a.ring_dims[i] = ring_dims[i]; a.cdim[i] = cdim[i];
        a.c0[i] = off[i] >> cshift;
        a.cn[i] = ((off[i] + shape[i] - 1) >> cshift) - a.c0[i] + 1;
    }
    const int total = a.cn[0] * a.cn[1] * a.cn[2];
    const dim3 g1((unsigned)((total + 255) / 256)), block(256);
#define SVR_CELL_RAW(T)                                                                            \
    do {                                                                                           \
        if (cshift == 3) hipLaunchKernelGGL((cell_raw_kernel<T, 8>), g1, block, 0, stream, a);      \
        else             hipLaunchKernelGGL((cell_raw_kernel<T, 4>), g1, block, 0, stream, a);      \
    } while (0)
    if (storage == SVR_U8)       SVR_CELL_RAW(uint8_t);
    else if (storage == SVR_U16) SVR_CELL_RAW(uint16_t);
    else                         SVR_CELL_RAW(float);
#undef SVR_CELL_RAW
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (int i = 0; i < 3; ++i) {                  // the blocks that contain a refreshed cell start one cell earlier
        a.c0[i] -= 1;
        a.cn[i] = a.cn[i] + 1 > a.cdim[i] ? a.cdim[i] : a.cn[i] + 1;
    }
    const int total2 = a.cn[0] * a.cn[1] * a.cn[2];
    const dim3 g2((unsigned)((total2 + 255) / 256));
    if (storage == SVR_U8)       hipLaunchKernelGGL((cell_block_kernel<uint8_t>), g2, block, 0, stream, a);
    else if (storage == SVR_U16) hipLaunchKernelGGL((cell_block_kernel<uint16_t>), g2, block, 0, stream, a);
    else                         hipLaunchKernelGGL((cell_block_kernel<float>), g2, block, 0, stream, a);
    return hipGetLastError();
}
