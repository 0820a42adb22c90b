// Micro-benchmark: sustained VALU issue rate per SIMD on gfx950 for the instruction kinds the
// march loop uses.  Each wave runs N iterations of 16 independent ops of one kind; we report
// cycles per wave-instruction per SIMD at a given occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 4096

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, int ia) {
    float x[16]; int y[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { x[j] = threadIdx.x * 0.001f + j; y[j] = threadIdx.x + j; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[j]) : "v"(a));
            if (KIND == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[j]) : "v"(a));
            if (KIND == 2) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(y[j]) : "v"(x[j]));
            if (KIND == 3) asm volatile("v_min_u32 %0, %1, %0" : "+v"(y[j]) : "v"(ia));
            if (KIND == 4) asm volatile("v_mad_u32_u24 %0, %1, %0, %1" : "+v"(y[j]) : "v"(ia));
            if (KIND == 5) asm volatile("v_add_u32 %0, %1, %0" : "+v"(y[j]) : "v"(ia));
            if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(x[j]) : "v"(a));
            if (KIND == 8) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(y[j]) : "v"(ia));
            if (KIND == 9) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[j]) : "v"(ia) : "vcc");
        }
        if (KIND == 7) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {x[j], x[j + 1]}; f2 b = {a, a};
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(v) : "v"(b));
                x[j] = v.x; x[j + 1] = v.y;
            }
        }
    }
    float s = 0; int t = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s += x[j]; t += y[j]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}

template <int KIND>
void run(const char* name, int waves_per_simd, int ops_per_iter) {
    int blocks = 256 * waves_per_simd;   // 256 CUs, 4 waves per block = 1 wave per SIMD per block
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(out, 1.0001f, 3);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(out, 1.0001f, 3);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double insts_per_simd = (double)ITERS * ops_per_iter * waves_per_simd;
    double ns_per_inst = ms * 1e6 / insts_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per_inst, ns_per_inst * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mul_f32", w, 16); run<1>("v_add_f32", w, 16); run<6>("v_fma_f32", w, 16);
        run<7>("v_pk_mul_f32", w, 8);
        run<2>("v_cvt_i32_f32", w, 16); run<3>("v_min_u32", w, 16); run<4>("v_mad_u32_u24", w, 16);
        run<5>("v_add_u32", w, 16); run<8>("v_lshl_add_u32", w, 16); run<9>("v_cmp+v_cndmask", w, 32);
    }
    return 0;
}
