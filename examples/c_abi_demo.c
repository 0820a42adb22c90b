/*
 * A consumer of include/svr.h written in plain C: no Python, no torch.  Builds a small two-level volume in host
 * memory, creates a context, uploads the levels, sets material and camera, renders one frame into device buffers it
 * allocated itself with the HIP runtime and writes the planes to disk.
 *
 *   gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_abi_demo.c \
 *       -L sub_volume_renderer_amd/csrc -lsvr_hip -L /opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/sub_volume_renderer_amd/csrc -Wl,-rpath,/opt/rocm/lib -o c_abi_demo
 *   ./c_abi_demo out_prefix [camera.bin]
 *
 * camera.bin (optional): the six column-major float[16] matrices of svr_camera in struct order, as a caller that
 * already has a camera (pygfx in the reference, tests/test_gpu_c_consumer.py here) would pass them; without it the
 * demo builds a look-at camera of its own.  Writes <prefix>.rgba.f32, <prefix>.label.u32, <prefix>.flags.u8 and
 * <prefix>.ppm (the RGBA plane over black, 8 bit, no transfer curve).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "svr.h"

#define N 64          /* finest level: N^3 voxels, u8 density + u32 labels */
#define W 160
#define H 96

#define CHECK(call)                                                                   \
    do {                                                                              \
        if ((call) != SVR_OK) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #call, svr_last_error());              \
            return 1;                                                                 \
        }                                                                             \
    } while (0)
#define HIP(call)                                                                     \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));         \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

/* the volume: a few bright balls in a dark fog, labelled by ball; level 1 is the 2x mean / max pyramid of level 0 */
static void make_level0(uint8_t* d, uint32_t* l) {
    static const float balls[5][4] = { { 20, 22, 18, 9 }, { 44, 30, 40, 11 }, { 30, 48, 24, 7 }, { 12, 44, 50, 6 }, { 50, 12, 14, 8 } };
    for (int z = 0; z < N; ++z)
        for (int y = 0; y < N; ++y)
            for (int x = 0; x < N; ++x) {
                uint8_t v = (uint8_t)(10 + ((x * 7 + y * 3 + z * 5) & 15));
                uint32_t lab = 0;
                for (int b = 0; b < 5; ++b) {
                    const float dx = x - balls[b][0], dy = y - balls[b][1], dz = z - balls[b][2];
                    if (dx * dx + dy * dy + dz * dz <= balls[b][3] * balls[b][3]) { v = (uint8_t)(150 + 20 * b); lab = (uint32_t)(b + 1); }
                }
                d[((size_t)z * N + y) * N + x] = v;
                l[((size_t)z * N + y) * N + x] = lab;
            }
}
static void pool2(const uint8_t* d, const uint32_t* l, int n, uint8_t* d2, uint32_t* l2) {
    const int m = n / 2;
    for (int z = 0; z < m; ++z)
        for (int y = 0; y < m; ++y)
            for (int x = 0; x < m; ++x) {
                unsigned sum = 0; uint32_t mx = 0;
                for (int k = 0; k < 8; ++k) {
                    const size_t i = ((size_t)(2 * z + (k >> 2)) * n + (2 * y + ((k >> 1) & 1))) * n + (2 * x + (k & 1));
                    sum += d[i]; if (l[i] > mx) mx = l[i];
                }
                d2[((size_t)z * m + y) * m + x] = (uint8_t)(sum / 8);
                l2[((size_t)z * m + y) * m + x] = mx;
            }
}

/* column-major 4x4 helpers (m[c*4+r]) */
static void mat_identity(float* m) { memset(m, 0, 16 * sizeof(float)); m[0] = m[5] = m[10] = m[15] = 1.0f; }
static int mat_invert(const float* m, float* out) {          /* general inverse by cofactors, in double */
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return 0;
    for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] / det);
    return 1;
}
/* camera at `eye` looking at `at`, y up; perspective with the frustum's depth mapped to [0, 1] (the convention of
 * u_stdinfo.projection_transform in pygfx: sub_volume_renderer_amd/_transform.py) */
static void default_camera(svr_camera* c) {
    const float eye[3] = { -70.0f, 95.0f, -55.0f }, at[3] = { 31.5f, 31.5f, 31.5f };
    float f[3] = { at[0] - eye[0], at[1] - eye[1], at[2] - eye[2] };
    const float fl = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (int i = 0; i < 3; ++i) f[i] /= fl;
    const float up[3] = { 0, 1, 0 };
    float s[3] = { f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0] };
    const float sl = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    for (int i = 0; i < 3; ++i) s[i] /= sl;
    const float u[3] = { s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0] };
    float* m = c->cam_inv;                                   /* the camera's world matrix: columns right, up, -forward, eye */
    mat_identity(m);
    for (int i = 0; i < 3; ++i) { m[i] = s[i]; m[4 + i] = u[i]; m[8 + i] = -f[i]; m[12 + i] = eye[i]; }
    mat_invert(c->cam_inv, c->cam);
    const float near = 0.5f, far = 2000.0f, fov = 50.0f * 3.14159265f / 180.0f, aspect = (float)W / (float)H;
    const float size = 2.0f * near * tanf(0.5f * fov), height = 2.0f * size / (1.0f + aspect), width = height * aspect;
    memset(c->proj, 0, sizeof(c->proj));
    c->proj[0] = near / (0.5f * width);
    c->proj[5] = near / (0.5f * height);
    c->proj[10] = far / (near - far);
    c->proj[14] = near * far / (near - far);
    c->proj[11] = -1.0f;
    mat_invert(c->proj, c->proj_inv);
}

static int write_file(const char* prefix, const char* suffix, const void* data, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof(path), "%s%s", prefix, suffix);
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    const size_t n = fwrite(data, 1, bytes, f);
    fclose(f);
    return n != bytes;
}

int main(int argc, char** argv) {
    const char* prefix = argc > 1 ? argv[1] : "c_abi_demo";
    if (svr_abi_version() != SVR_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }

    /* ---- the volume, two levels */
    uint8_t* d0 = malloc((size_t)N * N * N); uint32_t* l0 = malloc((size_t)N * N * N * 4);
    uint8_t* d1 = malloc((size_t)N * N * N / 8); uint32_t* l1 = malloc((size_t)N * N * N / 2);
    make_level0(d0, l0);
    pool2(d0, l0, N, d1, l1);

    /* ---- context: level 0 keeps a 32^3 ring (a window of the volume), level 1 holds its whole 32^3 level */
    svr_lod_desc lods[2];
    memset(lods, 0, sizeof(lods));
    for (int k = 0; k < 2; ++k) { lods[k].ring_dims[0] = lods[k].ring_dims[1] = lods[k].ring_dims[2] = 32; lods[k].density_storage = SVR_U8; }
    lods[0].blocked_twin = 1;       /* the finest ring also in 128-byte micro-blocks: waves pick the copy that suits their view (svr.h) */
    svr_ctx* ctx = NULL;
    CHECK(svr_create(0, 2, lods, &ctx));

    /* level 0: the window [16, 48)^3 — its voxels land in ring slots (p mod 32), i.e. two pieces per axis */
    const int32_t win = 16;
    for (int pz = 0; pz < 2; ++pz)
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int32_t src[3] = { win + 16 * px, win + 16 * py, win + 16 * pz };          /* logical voxel (x, y, z) */
                const int32_t dst[3] = { src[0] % 32, src[1] % 32, src[2] % 32 }, shape[3] = { 16, 16, 16 };
                const size_t first = ((size_t)src[2] * N + src[1]) * N + src[0];
                const int64_t ds[3] = { 1, N, (int64_t)N * N }, ls[3] = { 4, 4 * N, 4 * (int64_t)N * N };
                CHECK(svr_upload_region(ctx, 0, dst, shape, d0 + first, SVR_U8, ds, l0 + first, SVR_U32, ls));
            }
    {
        const int32_t dst[3] = { 0, 0, 0 }, shape[3] = { 32, 32, 32 };
        const int64_t ds[3] = { 1, 32, 32 * 32 }, ls[3] = { 4, 4 * 32, 4 * 32 * 32 };
        CHECK(svr_upload_region(ctx, 1, dst, shape, d1, SVR_U8, ds, l1, SVR_U32, ls));
    }
    CHECK(svr_publish_uploads(ctx));
    svr_lod_state st0 = { { win, win, win }, { 32, 32, 32 }, { 1.0f, 1.0f, 1.0f } };
    svr_lod_state st1 = { { 0, 0, 0 }, { 32, 32, 32 }, { 0.5f, 0.5f, 0.5f } };
    CHECK(svr_set_lod_state(ctx, 0, &st0));
    CHECK(svr_set_lod_state(ctx, 1, &st1));

    /* ---- material */
    static const float colors[6 * 4] = { 0.0f, 0.0f, 1.0f, 1.0f,  0.0f, 1.0f, 1.0f, 1.0f,  0.17f, 1.0f, 1.0f, 1.0f,
                                         0.33f, 1.0f, 1.0f, 1.0f,  0.55f, 1.0f, 1.0f, 1.0f,  0.8f, 1.0f, 1.0f, 1.0f };
    svr_material mat;
    memset(&mat, 0, sizeof(mat));
    mat.clim[0] = 0.0f; mat.clim[1] = 255.0f; mat.gamma = 1.0f; mat.opacity = 1.0f;
    mat.lmip_threshold = 120.0f; mat.lmip_fall_off = 0.5f; mat.lmip_max_samples = 10;
    mat.fog_density = 0.3f; mat.fog_color[0] = mat.fog_color[1] = mat.fog_color[2] = 0.5f;
    mat.color_count = 6; mat.colors = colors; mat.colorspace_srgb = 1;
    mat.render_mode = SVR_MODE_LMIP;
    CHECK(svr_set_material(ctx, &mat));

    /* ---- camera */
    svr_camera cam;
    memset(&cam, 0, sizeof(cam));
    mat_identity(cam.world); mat_identity(cam.world_inv);
    if (argc > 2) {
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(&cam, sizeof(float), 6 * 16, f) != 6 * 16) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        fclose(f);
    } else {
        default_camera(&cam);
    }
    cam.volume_dimensions[0] = cam.volume_dimensions[1] = cam.volume_dimensions[2] = (float)N;

    /* ---- one frame */
    svr_frame frame = { W, H, 0, 0, W, H, H, H };
    svr_outputs out;
    memset(&out, 0, sizeof(out));
    HIP(hipMalloc((void**)&out.rgba, (size_t)W * H * 16));
    HIP(hipMalloc((void**)&out.depth, (size_t)W * H * 4));
    HIP(hipMalloc((void**)&out.label, (size_t)W * H * 4));
    HIP(hipMalloc((void**)&out.flags, (size_t)W * H));
    CHECK(svr_render(ctx, &cam, &frame, &out, NULL));
    CHECK(svr_sync(ctx));

    float* rgba = malloc((size_t)W * H * 16); uint32_t* label = malloc((size_t)W * H * 4); uint8_t* flags = malloc((size_t)W * H);
    HIP(hipMemcpy(rgba, out.rgba, (size_t)W * H * 16, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(label, out.label, (size_t)W * H * 4, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(flags, out.flags, (size_t)W * H, hipMemcpyDeviceToHost));
    size_t hits = 0, frags = 0;
    for (size_t i = 0; i < (size_t)W * H; ++i) { hits += flags[i] == SVR_PIX_HIT; frags += flags[i] != SVR_PIX_DISCARD; }
    printf("abi %d: %zu fragments, %zu hits of %d pixels\n", svr_abi_version(), frags, hits, W * H);

    int bad = write_file(prefix, ".rgba.f32", rgba, (size_t)W * H * 16) | write_file(prefix, ".label.u32", label, (size_t)W * H * 4) |
              write_file(prefix, ".flags.u8", flags, (size_t)W * H);
    {
        uint8_t* ppm = malloc((size_t)W * H * 3 + 32);
        const int head = sprintf((char*)ppm, "P6\n%d %d\n255\n", W, H);
        for (size_t i = 0; i < (size_t)W * H; ++i)
            for (int k = 0; k < 3; ++k) {
                const float v = rgba[4 * i + k] * rgba[4 * i + 3];
                ppm[head + 3 * i + k] = (uint8_t)(v <= 0.0f ? 0 : (v >= 1.0f ? 255 : (int)(v * 255.0f + 0.5f)));
            }
        bad |= write_file(prefix, ".ppm", ppm, (size_t)head + (size_t)W * H * 3);
        free(ppm);
    }
    (void)hipFree(out.rgba); (void)hipFree(out.depth); (void)hipFree(out.label); (void)hipFree(out.flags);
    CHECK(svr_destroy(ctx));
    free(d0); free(l0); free(d1); free(l1); free(rgba); free(label); free(flags);
    return bad || hits == 0;
}
