// Host side of the C ABI declared in include/svr.h.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "svr_internal.h"

static thread_local std::string g_err;
void svr_set_error(const std::string& msg) { g_err = msg; }

#define SVR_REQUIRE(cond, msg)                    \
    do {                                          \
        if (!(cond)) {                            \
            svr_set_error(msg);                   \
            return SVR_ERR_INVALID;               \
        }                                         \
    } while (0)

namespace {

// f32 matrix helpers in the contract's operation order (see oracle/lmip_oracle.c header)
void mat_vec4(const float* m, const float* v, float* r) {
    for (int i = 0; i < 4; ++i) r[i] = ((m[0 + i] * v[0] + m[4 + i] * v[1]) + m[8 + i] * v[2]) + m[12 + i] * v[3];
}
void mat_mul4(const float* a, const float* b, float* out) {
    for (int c = 0; c < 4; ++c) mat_vec4(a, b + 4 * c, out + 4 * c);
}

int floor_div(int a, int b) { int q = a / b, r = a % b; return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q; }

}  // namespace

extern "C" {

const char* svr_last_error(void) { return g_err.c_str(); }
int svr_abi_version(void) { return SVR_ABI_VERSION; }

int svr_create(int device, int num_lods, const svr_lod_desc* lods, svr_ctx** out_ctx) {
    SVR_REQUIRE(out_ctx && lods, "svr_create: null argument");
    SVR_REQUIRE(num_lods >= 1 && num_lods <= SVR_MAX_LODS, "svr_create: num_lods out of range");
    int ndev = 0;
    SVR_HIP_TRY(hipGetDeviceCount(&ndev));
    SVR_REQUIRE(device >= 0 && device < ndev, "svr_create: no such HIP device (is a GPU visible?)");
    for (int l = 0; l < num_lods; ++l) {
        for (int a = 0; a < 3; ++a)
            SVR_REQUIRE(lods[l].ring_dims[a] >= 1 && lods[l].ring_dims[a] < (1 << 24),
                        "svr_create: ring extent must be in [1, 2^24)");
        SVR_REQUIRE(lods[l].density_storage == SVR_F32 || lods[l].density_storage == SVR_U8 ||
                    lods[l].density_storage == SVR_U16,
                    "svr_create: density_storage must be SVR_F32, SVR_U8 or SVR_U16");
        SVR_REQUIRE(lods[l].density_storage == lods[0].density_storage,
                    "svr_create: all LODs must use the same density_storage");
        SVR_REQUIRE(lods[l].blocked_twin >= 0 && lods[l].blocked_twin <= 2, "svr_create: blocked_twin must be 0, 1 or 2");
        SVR_REQUIRE(!lods[l].blocked_twin ||
                    (lods[l].ring_dims[0] % 8 == 0 && lods[l].ring_dims[1] % 4 == 0 && lods[l].ring_dims[2] % 4 == 0),
                    "svr_create: blocked_twin needs ring extents that are multiples of (8, 4, 4)");
        SVR_REQUIRE((lods[l].no_labels != 0) == (lods[0].no_labels != 0), "svr_create: all LODs must agree on no_labels");
    }
    DeviceGuard guard(device);
    svr_ctx* c = new svr_ctx();
    c->device = device; c->num_lods = num_lods;
    c->render_stream = nullptr; c->upload_stream = nullptr; c->uploads_published = nullptr;
    c->have_published = false; c->colors_dev = nullptr; c->colors_cap = 0; c->material_set = false;
    c->variant = 0; c->slot_bytes = 0; c->next_slot = 0; c->ev_a = c->ev_b = nullptr;
    c->next_ticket = 1;
    c->comm = nullptr; c->comm_rank = 0; c->comm_size = 1;
    for (auto& t : c->tickets) t = nullptr;
    c->uploads_marker = nullptr; c->marker_set = false; c->dbg_dev = nullptr;
    for (auto& s : c->slot) { s.host = s.dev = nullptr; s.done = nullptr; s.used = false; }
    for (int l = 0; l < SVR_MAX_LODS; ++l) { c->lod[l].density = nullptr; c->lod[l].twin = nullptr; c->lod[l].labels = nullptr; c->lod[l].voxels = 0; }
    c->density_all = nullptr; c->labels_all = nullptr; c->density_all_bytes = 0; c->twin_all = nullptr; c->twin_all_bytes = 0;
    c->cells_raw_all = c->cells_dil_all = nullptr; c->cells_all_bytes = 0;
    c->density_storage = lods[0].density_storage;
    c->no_labels = lods[0].no_labels != 0 ? 1 : 0;
    c->density_u8 = lods[0].density_storage == SVR_U8 ? 1 : 0;
    c->staged_bytes = 0; c->upload_seconds = 0.0;

    auto fail = [&](int code) { svr_destroy(c); return code; };
    if (hipStreamCreateWithFlags(&c->render_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->uploads_published, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->uploads_marker, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev_a) != hipSuccess || hipEventCreate(&c->ev_b) != hipSuccess) {
        svr_set_error("svr_create: stream/event creation failed");
        return fail(SVR_ERR_HIP);
    }
    size_t total = 0;   // in VOXELS (256-voxel aligned): the same offsets serve the u8/f32 density and the u32 labels
    for (int l = 0; l < num_lods; ++l) {
        LodStorage& L = c->lod[l];
        for (int a = 0; a < 3; ++a) L.ring[a] = lods[l].ring_dims[a];
        L.voxels = (size_t)L.ring[0] * (size_t)L.ring[1] * (size_t)L.ring[2];
        memset(&L.state, 0, sizeof(L.state));
        L.state.scale[0] = L.state.scale[1] = L.state.scale[2] = 1.0f;
        c->lod_base_bytes[l] = total;                         // voxel offset of this LOD
        total += (L.voxels + 255) & ~(size_t)255;
    }
    // the micro-block copies (svr_lod_desc::blocked_twin) live in an allocation of their own, each behind a buffer resource
    // of its own: the copies must not push the RINGS over the 4 GiB below which one resource reaches every LOD (the faster
    // build of the march; config 5's byte rings are 2.4 GB, with their copies 4.8)
    const size_t ring_voxels = total;
    size_t twin_base[SVR_MAX_LODS] = {0}, twin_voxels = 0;
    for (int l = 0; l < num_lods; ++l) {
        if (!lods[l].blocked_twin) continue;
        twin_base[l] = twin_voxels;
        twin_voxels += (c->lod[l].voxels + 255) & ~(size_t)255;
    }
    const size_t des = svr_dtype_size(c->density_storage);
    c->density_all_bytes = total * des + 64;                 // + slack: 16-byte brick loads may overrun a row end
    c->twin_all_bytes = twin_voxels ? twin_voxels * des + 64 : 0;
    // one allocation per plane type; zero-initialised textures (_wrapping_buffer.py:50-59)
    if (hipMalloc((void**)&c->density_all, c->density_all_bytes) != hipSuccess ||
        (c->twin_all_bytes && hipMalloc((void**)&c->twin_all, c->twin_all_bytes) != hipSuccess) ||
        (!c->no_labels && hipMalloc((void**)&c->labels_all, ring_voxels * sizeof(uint32_t)) != hipSuccess)) {
        svr_set_error("svr_create: out of device memory for ring textures");
        return fail(SVR_ERR_NOMEM);
    }
    if (hipMemsetAsync(c->density_all, 0, c->density_all_bytes, c->upload_stream) != hipSuccess ||
        (c->twin_all_bytes && hipMemsetAsync(c->twin_all, 0, c->twin_all_bytes, c->upload_stream) != hipSuccess) ||
        (!c->no_labels && hipMemsetAsync(c->labels_all, 0, ring_voxels * sizeof(uint32_t), c->upload_stream) != hipSuccess)) {
        svr_set_error("svr_create: memset failed");
        return fail(SVR_ERR_HIP);
    }
    for (int l = 0; l < num_lods; ++l) {
        c->lod[l].density = static_cast<char*>(c->density_all) + c->lod_base_bytes[l] * des;
        c->lod[l].labels = c->no_labels ? nullptr : c->labels_all + c->lod_base_bytes[l];
        c->lod[l].twin = lods[l].blocked_twin ? static_cast<char*>(c->twin_all) + twin_base[l] * des : nullptr;
        c->lod[l].twin_policy = lods[l].blocked_twin;
    }
    {   // macro-cell maxima (empty-space skipping): one grid of cells per LOD whose extents are multiples of 8.
        // The finest level gets 8^3-slot cells, the coarser ones 4^3: their structures are half / a quarter the size
        // in slots, and a ray advances less per iteration there, so the finer grid costs no extra tests.
        size_t cells = 0;
        for (int l = 0; l < num_lods; ++l) {
            LodStorage& L = c->lod[l];
            L.cells_raw = L.cells_dil = nullptr; L.cell_base = 0;
            L.cshift = l == 0 ? 3 : 2;
            for (int a = 0; a < 3; ++a) L.cdim[a] = L.ring[a] >> L.cshift;
            if ((L.ring[0] | L.ring[1] | L.ring[2]) & 7) continue;
            L.cell_base = cells;
            cells += ((size_t)L.cdim[0] * L.cdim[1] * L.cdim[2] + 63) & ~(size_t)63;
        }
        c->cells_all_bytes = cells * des;
        if (cells && c->cells_all_bytes < ((size_t)1 << 31)) {
            if (hipMalloc(&c->cells_raw_all, c->cells_all_bytes) != hipSuccess ||
                hipMalloc(&c->cells_dil_all, c->cells_all_bytes) != hipSuccess) {
                svr_set_error("svr_create: out of device memory for the macro-cell grids");
                return fail(SVR_ERR_NOMEM);
            }
            if (hipMemsetAsync(c->cells_raw_all, 0, c->cells_all_bytes, c->upload_stream) != hipSuccess ||
                hipMemsetAsync(c->cells_dil_all, 0, c->cells_all_bytes, c->upload_stream) != hipSuccess) {
                svr_set_error("svr_create: memset failed");
                return fail(SVR_ERR_HIP);
            }
            for (int l = 0; l < num_lods; ++l) {
                LodStorage& L = c->lod[l];
                if ((L.ring[0] | L.ring[1] | L.ring[2]) & 7) continue;
                L.cells_raw = static_cast<char*>(c->cells_raw_all) + L.cell_base * des;
                L.cells_dil = static_cast<char*>(c->cells_dil_all) + L.cell_base * des;
            }
        } else {
            c->cells_all_bytes = 0;
        }
    }
    if (hipEventRecord(c->uploads_published, c->upload_stream) != hipSuccess) return fail(SVR_ERR_HIP);
    c->have_published = true;
    *out_ctx = c;
    return SVR_OK;
}

int svr_destroy(svr_ctx* c) {
    if (!c) return SVR_OK;
    DeviceGuard guard(c->device);
    if (c->comm) (void)svr_comm_destroy(c);
    if (c->render_stream) (void)hipStreamSynchronize(c->render_stream);
    if (c->upload_stream) (void)hipStreamSynchronize(c->upload_stream);
    if (c->density_all) (void)hipFree(c->density_all);
    if (c->twin_all) (void)hipFree(c->twin_all);
    if (c->labels_all) (void)hipFree(c->labels_all);
    if (c->cells_raw_all) (void)hipFree(c->cells_raw_all);
    if (c->cells_dil_all) (void)hipFree(c->cells_dil_all);
    for (auto& s : c->slot) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.dev) (void)hipFree(s.dev);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->colors_dev) (void)hipFree(c->colors_dev);
    for (float* p : c->colors_retired) (void)hipFree(p);
    if (c->dbg_dev) (void)hipFree(c->dbg_dev);
    for (auto& t : c->tile_orders) if (t.dev) (void)hipFree(t.dev);
    for (auto& t : c->cost_orders) {
        if (t.dev) (void)hipFree(t.dev);
        if (t.host) (void)hipHostFree(t.host);
        if (t.copied) (void)hipEventDestroy(t.copied);
        if (t.drawn) (void)hipEventDestroy(t.drawn);
    }
    if (c->uploads_published) (void)hipEventDestroy(c->uploads_published);
    for (auto& m : c->render_marks) if (m.done) (void)hipEventDestroy(m.done);
    for (auto& t : c->tickets) if (t) (void)hipEventDestroy(t);
    if (c->uploads_marker) (void)hipEventDestroy(c->uploads_marker);
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->render_stream) (void)hipStreamDestroy(c->render_stream);
    if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    delete c;
    return SVR_OK;
}

int svr_set_lod_state(svr_ctx* c, int lod, const svr_lod_state* st) {
    SVR_REQUIRE(c && st, "svr_set_lod_state: null argument");
    SVR_REQUIRE(lod >= 0 && lod < c->num_lods, "svr_set_lod_state: lod out of range");
    for (int a = 0; a < 3; ++a) {
        SVR_REQUIRE(st->offset[a] >= 0 && st->shape[a] >= 0, "svr_set_lod_state: negative offset/shape");
        SVR_REQUIRE(st->shape[a] <= c->lod[lod].ring[a], "svr_set_lod_state: ROI larger than the ring");
    }
    c->lod[lod].state = *st;
    return SVR_OK;
}

int svr_get_lod_state(svr_ctx* c, int lod, svr_lod_state* st) {
    SVR_REQUIRE(c && st, "svr_get_lod_state: null argument");
    SVR_REQUIRE(lod >= 0 && lod < c->num_lods, "svr_get_lod_state: lod out of range");
    *st = c->lod[lod].state;
    return SVR_OK;
}

int svr_set_material(svr_ctx* c, const svr_material* m) {
    SVR_REQUIRE(c && m, "svr_set_material: null argument");
    SVR_REQUIRE(m->color_count >= 1 && m->colors, "svr_set_material: at least one colour is required");
    SVR_REQUIRE(m->clipping_plane_count <= SVR_MAX_CLIP_PLANES, "svr_set_material: too many clipping planes");
    SVR_REQUIRE(m->clipping_plane_count == 0 || m->clipping_planes, "svr_set_material: clipping_planes is null");
    SVR_REQUIRE(m->render_mode == SVR_MODE_LMIP || m->render_mode == SVR_MODE_WEIGHTED_AVERAGE, "svr_set_material: unknown render_mode");
    SVR_REQUIRE(m->render_mode != SVR_MODE_WEIGHTED_AVERAGE || (m->weight_falloff >= 0.0f && m->weight_falloff < INFINITY),
                "svr_set_material: weight_falloff must be finite and >= 0");
    DeviceGuard guard(c->device);
    const size_t ncol = (size_t)m->color_count * 4;
    const bool same_colors = c->material_set && c->colors_host.size() == ncol &&
                             memcmp(c->colors_host.data(), m->colors, ncol * sizeof(float)) == 0;
    c->material = *m;
    c->material.colors = nullptr;
    c->material.clipping_planes = nullptr;
    c->clip_host.assign(m->clipping_planes, m->clipping_planes + (size_t)m->clipping_plane_count * 4);
    if (same_colors) return SVR_OK;
    c->colors_host.assign(m->colors, m->colors + ncol);
    // The colour table of renders still in flight must not change under them: every new table goes into a
    // fresh device buffer (the old one is freed once the device has drained; tables are tiny and change
    // rarely), so no device-wide synchronisation is needed here.
    float* fresh = nullptr;
    const uint32_t cap = std::max<uint32_t>(256u, m->color_count);
    if (hipMalloc((void**)&fresh, (size_t)cap * 4 * sizeof(float)) != hipSuccess) {
        svr_set_error("svr_set_material: out of device memory"); return SVR_ERR_NOMEM;
    }
    SVR_HIP_TRY(hipMemcpy(fresh, c->colors_host.data(), c->colors_host.size() * sizeof(float), hipMemcpyHostToDevice));
    if (c->colors_dev) c->colors_retired.push_back(c->colors_dev);
    if (c->colors_retired.size() > 32) {                 // bounded: drain once in a long while
        SVR_HIP_TRY(hipDeviceSynchronize());
        for (float* p : c->colors_retired) (void)hipFree(p);
        c->colors_retired.clear();
    }
    c->colors_dev = fresh;
    c->colors_cap = cap;
    c->material_set = true;
    return SVR_OK;
}

int svr_set_variant(svr_ctx* c, int variant) {
    SVR_REQUIRE(c, "svr_set_variant: null ctx");
    // bits 11-12 are timing experiments that render WRONG pixels: only a -DSVR_EXPERIMENTS build knows them
    SVR_REQUIRE((variant & 0x1800 & ~SVR_EXP_VARIANT_BITS) == 0,
                "svr_set_variant: bits 11-12 (timing experiments) exist only in builds made with -DSVR_EXPERIMENTS");
    c->variant = variant;
    return SVR_OK;
}

// ---------------------------------------------------------------------------
// uploads
// ---------------------------------------------------------------------------
static int check_region(svr_ctx* c, int lod, const int32_t off[3], const int32_t shape[3], const char* who) {
    if (!c || !off || !shape) { svr_set_error(std::string(who) + ": null argument"); return SVR_ERR_INVALID; }
    if (lod < 0 || lod >= c->num_lods) { svr_set_error(std::string(who) + ": lod out of range"); return SVR_ERR_INVALID; }
    for (int a = 0; a < 3; ++a) {
        if (off[a] < 0 || shape[a] < 0 || (int64_t)off[a] + shape[a] > c->lod[lod].ring[a]) {
            svr_set_error(std::string(who) + ": region outside the ring");
            return SVR_ERR_RANGE;
        }
    }
    return SVR_OK;
}

// A render enqueued earlier may still be reading the ring slots an upload is
// about to overwrite: order the upload stream behind it (the reference gets this
// ordering from running everything on one queue).
static int uploads_after_render(svr_ctx* c) {
    std::lock_guard<std::mutex> lock(c->marks_mu);
    for (auto& m : c->render_marks)
        if (m.pending) {
            SVR_HIP_TRY(hipStreamWaitEvent(c->upload_stream, m.done, 0));
            m.pending = false;
        }
    return SVR_OK;
}

// Remember that stream `s` carries a render enqueued just now (one event per stream; a stream's later
// render re-records its event, which then covers the earlier ones on that stream too).
static int mark_render(svr_ctx* c, hipStream_t s) {
    std::lock_guard<std::mutex> lock(c->marks_mu);
    svr_ctx::RenderMark* mark = nullptr;
    for (auto& m : c->render_marks) if (m.stream == s) { mark = &m; break; }
    if (!mark) {
        if (c->render_marks.size() >= 64) {
            // a caller cycling through many short-lived streams: recycle the oldest entry, after ordering the
            // upload stream behind the render it still stands for
            svr_ctx::RenderMark old = c->render_marks.front();
            if (old.pending) SVR_HIP_TRY(hipStreamWaitEvent(c->upload_stream, old.done, 0));
            c->render_marks.erase(c->render_marks.begin());
            old.stream = s; old.pending = false;
            c->render_marks.push_back(old);
        } else {
            svr_ctx::RenderMark m{ s, nullptr, false };
            SVR_HIP_TRY(hipEventCreateWithFlags(&m.done, hipEventDisableTiming));
            c->render_marks.push_back(m);
        }
        mark = &c->render_marks.back();
    }
    SVR_HIP_TRY(hipEventRecord(mark->done, s));
    mark->pending = true;
    return SVR_OK;
}

static int ensure_staging(svr_ctx* c) {
    if (c->slot_bytes) return SVR_OK;
    const size_t bytes = (size_t)48 << 20;   // per slot; 3 slots -> 144 MiB pinned
    for (auto& s : c->slot) {
        if (hipHostMalloc(&s.host, bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc(&s.dev, bytes) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
            svr_set_error("staging allocation failed");
            return SVR_ERR_NOMEM;
        }
        s.used = false;
    }
    c->slot_bytes = bytes;
    return SVR_OK;
}

// Scatter one block into a LOD's rings and bring the macro-cell maxima of the touched cells up to date
// (same stream, right behind the scatter: a render that can see the new voxels sees their maxima too).
static hipError_t scatter_into_ring(svr_ctx* c, int lod, const ScatterArgs& a, bool wrote_density) {
    hipError_t e = svr_launch_scatter(a, c->upload_stream);
    const LodStorage& L = c->lod[lod];
    if (e == hipSuccess && wrote_density && L.cells_raw)
        e = svr_launch_cell_update(L.density, c->density_storage, L.ring, L.cells_raw, L.cells_dil, L.cdim, L.cshift,
                                   a.dst_off, a.shape, c->upload_stream);
    return e;
}

// Copy rows [r0, r1) of a strided host block into a packed buffer (x fastest); a row index is z * shape[1] + y.
static void pack_row_range(const char* src, size_t es, const int64_t st[3], const int32_t shape[3],
                           int64_t r0, int64_t r1, char* dst) {
    const size_t row = (size_t)shape[0] * es;
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t z = r / shape[1], y = r % shape[1];
        const char* s = src + z * st[2] + y * st[1];
        if (st[0] == (int64_t)es) memcpy(dst, s, row);
        else for (int x = 0; x < shape[0]; ++x) memcpy(dst + (size_t)x * es, s + (int64_t)x * st[0], es);
        dst += row;
    }
}

// The host half of an upload is a gather of short rows (48 .. 528 voxels) out of a large strided array into the
// pinned staging slot: one core moves ~6-15 GB/s that way, well under the PCIe rate, so the rows are dealt to a
// few threads (contiguous row ranges).  The threads are kept: a block is a few MiB to tens of MiB, i.e. 0.1-1 ms
// of copying, and starting 7 threads per block costs a good part of that.
class PackPool {
public:
    static PackPool& get() {
        static std::mutex mu;
        static PackPool* pool = nullptr;
        static pid_t owner = 0;
        std::lock_guard<std::mutex> g(mu);
        if (!pool || owner != getpid()) { pool = new PackPool(); owner = getpid(); }   // threads do not survive fork(): the child starts its own (the parent's object is left alone)
        return *pool;
    }
    static constexpr int kMax = 16;
    // fn(t) for t in [0, nt): t = 0 on the caller, the rest on the workers; returns when all are done
    void run(int nt, const std::function<void(int)>& fn) {
        nt = std::min(nt, kMax);
        if (nt <= 1) { fn(0); return; }
        std::lock_guard<std::mutex> one(run_mu);
        {
            std::lock_guard<std::mutex> g(mu);
            while ((int)workers.size() < nt - 1) { const int id = (int)workers.size() + 1; workers.emplace_back([this, id] { loop(id); }); }
            job = &fn; want = nt; left = nt - 1; ++gen;
        }
        go.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(mu);
        done.wait(g, [this] { return left == 0; });
        job = nullptr;
    }
    ~PackPool() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        go.notify_all();
        for (auto& w : workers) w.join();
    }
private:
    void loop(int id) {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            go.wait(g, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            if (id >= want) continue;
            const std::function<void(int)>* f = job;
            g.unlock();
            (*f)(id);
            g.lock();
            if (--left == 0) done.notify_one();
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu, run_mu;
    std::condition_variable go, done;
    const std::function<void(int)>* job = nullptr;
    uint64_t gen = 0;
    int want = 0, left = 0;
    bool stop = false;
};

// Pack rows [r0, r1) of the density block and of the label block (either may be absent) in one parallel pass.
static void pack_rows(const char* dsrc, size_t des, const int64_t* dst_strides, char* ddst,
                      const char* lsrc, size_t les, const int64_t* lst_strides, char* ldst,
                      const int32_t shape[3], int64_t r0, int64_t r1) {
    const int64_t nrows = r1 - r0;
    const size_t bytes = (size_t)shape[0] * (size_t)nrows * ((dsrc ? des : 0) + (lsrc ? les : 0));
    unsigned hw = std::thread::hardware_concurrency();
    // (12 or 16 threads, and streaming stores into the slot, measured no faster)
    int nt = (int)std::min<size_t>(std::min<unsigned>(hw ? hw : 1u, 8u), bytes / ((size_t)512 << 10));
    if (const char* e = getenv("SVR_PACK_THREADS")) nt = std::max(1, atoi(e));      // deployment setting (include/svr.h), not an experiment
    nt = (int)std::min<int64_t>(std::max(nt, 1), std::max<int64_t>(1, nrows / 2));
    const int64_t per = (nrows + nt - 1) / nt;
    PackPool::get().run(nt, [&](int t) {
        const int64_t a = r0 + per * t, b = std::min(r1, a + per);
        if (a >= b) return;
        if (dsrc) pack_row_range(dsrc, des, dst_strides, shape, a, b, ddst + (size_t)(a - r0) * shape[0] * des);
        if (lsrc) pack_row_range(lsrc, les, lst_strides, shape, a, b, ldst + (size_t)(a - r0) * shape[0] * les);
    });
}

int svr_upload_region(svr_ctx* c, int lod, const int32_t dst_off[3], const int32_t shape[3],
                      const void* density, int density_dtype, const int64_t density_strides[3],
                      const void* labels, int labels_dtype, const int64_t labels_strides[3]) {
    int rc = check_region(c, lod, dst_off, shape, "svr_upload_region");
    if (rc) return rc;
    const size_t des = density ? svr_dtype_size(density_dtype) : 0;
    const size_t les = labels ? svr_dtype_size(labels_dtype) : 0;
    SVR_REQUIRE(!density || (des && density_strides), "svr_upload_region: bad density dtype/strides");
    SVR_REQUIRE(!labels || (les && labels_strides), "svr_upload_region: bad labels dtype/strides");
    if (!density && !labels) return SVR_OK;
    SVR_REQUIRE(!labels || !c->no_labels, "svr_upload_region: the context was created without label rings (no_labels)");
    SVR_REQUIRE(!density || storage_accepts(c->density_storage, density_dtype),
                "svr_upload_region: the context's integer density storage only accepts sources of that same dtype");
    if (shape[0] == 0 || shape[1] == 0 || shape[2] == 0) return SVR_OK;
    DeviceGuard guard(c->device);
    // one uploader at a time: the staging slots and the enqueue order on the upload stream are shared
    // (the render thread may load synchronously while the streaming worker is inside this call)
    std::lock_guard<std::mutex> upload_lock(c->upload_mu);
    struct Clock {                                          // time inside this call, for svr_upload_stats
        svr_ctx* c; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Clock() { c->upload_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } clock{ c };
    rc = ensure_staging(c);
    if (rc) return rc;
    rc = uploads_after_render(c);
    if (rc) return rc;
    LodStorage& L = c->lod[lod];
    // The region travels in blocks of whole rows, as many as fit a staging slot: whole z-planes when one
    // fits (the scatter kernel then sees a box), else runs of rows inside one plane.  Labels are placed
    // after the density block, 16-byte aligned.
    const size_t row_bytes = (size_t)shape[0] * (des + les);
    SVR_REQUIRE(row_bytes + 32 <= c->slot_bytes, "svr_upload_region: one row of the region exceeds the staging slot");
    // Blocks of about an eighth of the region (4 MiB .. one slot): the pack of block k+1 runs while block k is on
    // the bus and block k-1 is being scattered, so a region of one or two slots' worth still overlaps the three.
    const size_t region_bytes = row_bytes * (size_t)shape[1] * (size_t)shape[2];
    size_t block_bytes = std::min(c->slot_bytes - 32, std::max<size_t>((size_t)4 << 20, region_bytes / 8));
    if (const int mib = svr_exp_env_int("SVR_UPLOAD_BLOCK_MIB", 0)) block_bytes = std::min(c->slot_bytes - 32, std::max<size_t>(1, (size_t)mib) << 20);
    const int64_t rows_max = std::max<int64_t>(1, (int64_t)(block_bytes / row_bytes));
    const int64_t plane_rows = shape[1];
    const int64_t total_rows = plane_rows * shape[2];
    for (int64_t r0 = 0; r0 < total_rows;) {
        int64_t nrows;
        if (rows_max >= plane_rows && r0 % plane_rows == 0)
            nrows = std::min(total_rows - r0, (rows_max / plane_rows) * plane_rows);        // whole planes
        else
            nrows = std::min(rows_max, plane_rows - r0 % plane_rows);                        // rows of one plane
        const int64_t r1 = r0 + nrows;
        StagingSlot& S = c->slot[c->next_slot];
        c->next_slot = (c->next_slot + 1) % svr_ctx::kSlots;
        if (S.used) SVR_HIP_TRY(hipEventSynchronize(S.done));
        const size_t dbytes = (size_t)shape[0] * (size_t)nrows * des;
        const size_t lofs = (dbytes + 15) & ~(size_t)15;
        const size_t lbytes = (size_t)shape[0] * (size_t)nrows * les;
        pack_rows(static_cast<const char*>(density), des, density_strides, static_cast<char*>(S.host),
                  static_cast<const char*>(labels), les, labels_strides, static_cast<char*>(S.host) + lofs, shape, r0, r1);
        SVR_HIP_TRY(hipMemcpyAsync(S.dev, S.host, lofs + lbytes, hipMemcpyHostToDevice, c->upload_stream));
        // the staged block as a box of the ring
        const int32_t z0 = (int32_t)(r0 / plane_rows), y0 = (int32_t)(r0 % plane_rows);
        const int32_t bz = nrows >= plane_rows ? (int32_t)(nrows / plane_rows) : 1;
        const int32_t by = nrows >= plane_rows ? shape[1] : (int32_t)nrows;
        ScatterArgs a;
        a.src_density = density ? S.dev : nullptr; a.density_dtype = density_dtype;
        a.dstride[0] = (int64_t)des; a.dstride[1] = (int64_t)des * shape[0]; a.dstride[2] = (int64_t)des * shape[0] * by;
        a.src_labels = labels ? static_cast<char*>(S.dev) + lofs : nullptr; a.labels_dtype = labels_dtype;
        a.lstride[0] = (int64_t)les; a.lstride[1] = (int64_t)les * shape[0]; a.lstride[2] = (int64_t)les * shape[0] * by;
        a.ring_density = L.density; a.ring_labels = L.labels; a.ring_storage = c->density_storage; a.ring_twin = L.twin;
        for (int i = 0; i < 3; ++i) a.ring[i] = L.ring[i];
        a.dst_off[0] = dst_off[0]; a.dst_off[1] = dst_off[1] + y0; a.dst_off[2] = dst_off[2] + z0;
        a.shape[0] = shape[0]; a.shape[1] = by; a.shape[2] = bz;
        a.packed = 1;
        SVR_HIP_TRY(scatter_into_ring(c, lod, a, density != nullptr));
        SVR_HIP_TRY(hipEventRecord(S.done, c->upload_stream));
        S.used = true;
        c->staged_bytes += lofs + lbytes;
        r0 = r1;
    }
    return SVR_OK;
}

int svr_upload_region_device(svr_ctx* c, int lod, const int32_t dst_off[3], const int32_t shape[3],
                             const void* density, int density_dtype, const int64_t density_strides[3],
                             const void* labels, int labels_dtype, const int64_t labels_strides[3]) {
    int rc = check_region(c, lod, dst_off, shape, "svr_upload_region_device");
    if (rc) return rc;
    SVR_REQUIRE(!density || (svr_dtype_size(density_dtype) && density_strides), "svr_upload_region_device: bad density dtype/strides");
    SVR_REQUIRE(!labels || (svr_dtype_size(labels_dtype) && labels_strides), "svr_upload_region_device: bad labels dtype/strides");
    if (!density && !labels) return SVR_OK;
    SVR_REQUIRE(!labels || !c->no_labels, "svr_upload_region_device: the context was created without label rings (no_labels)");
    SVR_REQUIRE(!density || storage_accepts(c->density_storage, density_dtype),
                "svr_upload_region_device: the context's integer density storage only accepts sources of that same dtype");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> upload_lock(c->upload_mu);
    rc = uploads_after_render(c);
    if (rc) return rc;
    LodStorage& L = c->lod[lod];
    ScatterArgs a;
    a.ring_storage = c->density_storage; a.packed = 0;
    a.src_density = density; a.density_dtype = density_dtype;
    a.src_labels = labels; a.labels_dtype = labels_dtype;
    for (int i = 0; i < 3; ++i) {
        a.dstride[i] = density ? density_strides[i] : 0;
        a.lstride[i] = labels ? labels_strides[i] : 0;
        a.ring[i] = L.ring[i]; a.dst_off[i] = dst_off[i]; a.shape[i] = shape[i];
    }
    a.ring_density = L.density; a.ring_labels = L.labels; a.ring_twin = L.twin;
    SVR_HIP_TRY(scatter_into_ring(c, lod, a, density != nullptr));
    return SVR_OK;
}

int svr_publish_uploads(svr_ctx* c) {
    SVR_REQUIRE(c, "svr_publish_uploads: null ctx");
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(hipEventRecord(c->uploads_published, c->upload_stream));
    c->have_published = true;
    return SVR_OK;
}

int svr_mark_uploads(svr_ctx* c) {
    SVR_REQUIRE(c, "svr_mark_uploads: null ctx");
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(hipEventRecord(c->uploads_marker, c->upload_stream));
    c->marker_set = true;
    return SVR_OK;
}

int svr_uploads_pending(svr_ctx* c, int* pending) {
    SVR_REQUIRE(c && pending, "svr_uploads_pending: null argument");
    *pending = 0;
    if (!c->marker_set) return SVR_OK;
    DeviceGuard guard(c->device);
    const hipError_t e = hipEventQuery(c->uploads_marker);
    if (e == hipErrorNotReady) { *pending = 1; return SVR_OK; }
    SVR_HIP_TRY(e);
    return SVR_OK;
}

int svr_upload_stats(svr_ctx* c, uint64_t* staged_bytes, double* seconds_in_calls, int reset) {
    SVR_REQUIRE(c, "svr_upload_stats: null ctx");
    std::lock_guard<std::mutex> upload_lock(c->upload_mu);
    if (staged_bytes) *staged_bytes = c->staged_bytes;
    if (seconds_in_calls) *seconds_in_calls = c->upload_seconds;
    if (reset) { c->staged_bytes = 0; c->upload_seconds = 0.0; }
    return SVR_OK;
}

int svr_upload_ticket(svr_ctx* c, uint64_t* ticket) {
    SVR_REQUIRE(c && ticket, "svr_upload_ticket: null argument");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> upload_lock(c->upload_mu);      // behind whatever an uploader is enqueueing right now
    std::lock_guard<std::mutex> lock(c->ticket_mu);
    hipEvent_t& ev = c->tickets[c->next_ticket % svr_ctx::kTickets];
    if (!ev) SVR_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    SVR_HIP_TRY(hipEventRecord(ev, c->upload_stream));
    *ticket = c->next_ticket++;
    return SVR_OK;
}

int svr_ticket_pending(svr_ctx* c, uint64_t ticket, int* pending) {
    SVR_REQUIRE(c && pending, "svr_ticket_pending: null argument");
    *pending = 0;
    std::lock_guard<std::mutex> lock(c->ticket_mu);
    SVR_REQUIRE(ticket >= 1 && ticket < c->next_ticket, "svr_ticket_pending: no such ticket");
    // a recycled slot holds a LATER event of the same stream: done there implies done here
    DeviceGuard guard(c->device);
    const hipError_t e = hipEventQuery(c->tickets[ticket % svr_ctx::kTickets]);
    if (e == hipErrorNotReady) { *pending = 1; return SVR_OK; }
    SVR_HIP_TRY(e);
    return SVR_OK;
}

int svr_clear_lod(svr_ctx* c, int lod) {
    SVR_REQUIRE(c, "svr_clear_lod: null ctx");
    SVR_REQUIRE(lod >= 0 && lod < c->num_lods, "svr_clear_lod: lod out of range");
    DeviceGuard guard(c->device);
    int rc = uploads_after_render(c);
    if (rc) return rc;
    LodStorage& L = c->lod[lod];
    std::lock_guard<std::mutex> upload_lock(c->upload_mu);
    SVR_HIP_TRY(hipMemsetAsync(L.density, 0, L.voxels * svr_dtype_size(c->density_storage), c->upload_stream));
    if (L.twin) SVR_HIP_TRY(hipMemsetAsync(L.twin, 0, L.voxels * svr_dtype_size(c->density_storage), c->upload_stream));
    if (L.labels) SVR_HIP_TRY(hipMemsetAsync(L.labels, 0, L.voxels * sizeof(uint32_t), c->upload_stream));
    if (L.cells_raw) {
        const size_t cb = (size_t)L.cdim[0] * L.cdim[1] * L.cdim[2] * svr_dtype_size(c->density_storage);
        SVR_HIP_TRY(hipMemsetAsync(L.cells_raw, 0, cb, c->upload_stream));
        SVR_HIP_TRY(hipMemsetAsync(L.cells_dil, 0, cb, c->upload_stream));
    }
    return SVR_OK;
}

int svr_read_region(svr_ctx* c, int lod, const int32_t off[3], const int32_t shape[3],
                    float* density_out, uint32_t* labels_out) {
    int rc = check_region(c, lod, off, shape, "svr_read_region");
    if (rc) return rc;
    const size_t n = (size_t)shape[0] * (size_t)shape[1] * (size_t)shape[2];
    if (n == 0 || (!density_out && !labels_out)) return SVR_OK;
    DeviceGuard guard(c->device);
    LodStorage& L = c->lod[lod];
    float* dtmp = nullptr; uint32_t* ltmp = nullptr;
    if (labels_out && !L.labels) {                     // no label rings: every label reads as 0
        memset(labels_out, 0, n * sizeof(uint32_t));
        labels_out = nullptr;
        if (!density_out) return SVR_OK;
    }
    SVR_HIP_TRY(hipStreamSynchronize(c->upload_stream));
    if (density_out) SVR_HIP_TRY(hipMalloc((void**)&dtmp, n * sizeof(float)));
    if (labels_out && hipMalloc((void**)&ltmp, n * sizeof(uint32_t)) != hipSuccess) {
        if (dtmp) (void)hipFree(dtmp);
        svr_set_error("svr_read_region: out of device memory"); return SVR_ERR_NOMEM;
    }
    hipError_t e = svr_launch_gather(L.density, c->density_storage, L.labels, L.ring, off, shape, dtmp, ltmp, c->upload_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->upload_stream);
    if (e == hipSuccess && dtmp) e = hipMemcpy(density_out, dtmp, n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && ltmp) e = hipMemcpy(labels_out, ltmp, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (dtmp) (void)hipFree(dtmp);
    if (ltmp) (void)hipFree(ltmp);
    if (e != hipSuccess) { svr_set_error(std::string("svr_read_region: ") + hipGetErrorString(e)); return SVR_ERR_HIP; }
    return SVR_OK;
}

// ---------------------------------------------------------------------------
// the draw
// ---------------------------------------------------------------------------
// Block -> tile table.  Workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own
// L2, so (1) tiles that sample neighbouring voxels should share an XCD and (2) every XCD should get the
// same amount of work.  A contiguous run of tiles per XCD (mode 1) is best for (1) but gives each XCD
// one horizontal band of the frame, and the bands differ in cost; instead deal CHUNKS of cw x ch tiles
// round-robin to the XCDs, so each XCD samples the whole frame.  Placement never affects results.
static void build_tile_order(int tx, int ty, int cw, int ch, std::vector<uint32_t>& order) {
    const int n = tx * ty;
    std::vector<uint32_t> seq;
    seq.reserve((size_t)n);
    for (int cy = 0; cy < ty; cy += ch)
        for (int cx = 0; cx < tx; cx += cw)
            for (int y = cy; y < std::min(cy + ch, ty); ++y)
                for (int x = cx; x < std::min(cx + cw, tx); ++x) seq.push_back((uint32_t)(y * tx + x));
    std::vector<uint32_t> lists[8];
    const int run = cw * ch;
    for (int i = 0; i < n; ++i) lists[(i / run) % 8].push_back(seq[(size_t)i]);
    // XCD k runs blocks k, k + 8, ...: exactly ceil((n - k) / 8) of them
    std::vector<uint32_t> spare;
    for (int k = 0; k < 8; ++k) {
        const size_t need = (size_t)((n - k + 7) / 8);
        while (lists[k].size() > need) { spare.push_back(lists[k].back()); lists[k].pop_back(); }
    }
    for (int k = 0; k < 8; ++k) {
        const size_t need = n > k ? (size_t)((n - k + 7) / 8) : 0;
        while (lists[k].size() < need) { lists[k].push_back(spare.back()); spare.pop_back(); }
    }
    order.assign((size_t)n, 0u);
    for (int k = 0; k < 8; ++k)
        for (size_t j = 0; j < lists[k].size(); ++j) order[8 * j + (size_t)k] = lists[k][j];
}

// Length (in finest-level voxels) of the part of the pixel's ray inside the proxy box: what a full march of that
// pixel costs, up to the constant step.  Same geometry as setup_ray, evaluated loosely (placement only).
static float ray_cost(const MarchParams& P, float px_frame, float py_frame) {
    const float ndcx = 2.0f * px_frame / (float)P.frame.frame_w - 1.0f, ndcy = 1.0f - 2.0f * py_frame / (float)P.frame.frame_h;
    const float nv[4] = { ndcx, ndcy, -1.0f, 1.0f }, fv[4] = { ndcx, ndcy, 1.0f, 1.0f };
    float n4[4], f4[4];
    mat_vec4(P.ndc_to_data, nv, n4);
    mat_vec4(P.ndc_to_data, fv, f4);
    float o[3], d[3], len = 0.0f;
    for (int a = 0; a < 3; ++a) { o[a] = n4[a] / n4[3]; d[a] = f4[a] / f4[3] - o[a]; len += d[a] * d[a]; }
    len = sqrtf(len);
    if (!(len > 0.0f)) return 0.0f;
    float t0 = 0.0f, t1 = len;
    for (int a = 0; a < 3; ++a) {
        const float r = d[a] / len, lo = -0.5f, hi = P.size[a] - 0.5f;
        if (fabsf(r) < 1e-12f) { if (o[a] < lo || o[a] > hi) return 0.0f; continue; }
        const float ta = (lo - o[a]) / r, tb = (hi - o[a]) / r;
        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
    }
    return t1 > t0 ? t1 - t0 : 0.0f;
}

// Cost-sorted placement: chunks of tiles ordered by the length of their rays, longest first, and dealt to the XCDs
// in snake order (0..7, 7..0, ...): every XCD gets the same share of the work, the expensive tiles start first and
// the cheap ones fill the end of the frame (longest-processing-time-first scheduling of the frame's tail).
static void build_tile_order_by_cost(const MarchParams& P, int tx, int ty, int tile_w, int tile_h, int cw, int ch,
                                     std::vector<uint32_t>& order) {
    struct Chunk { float cost; int cx, cy; };
    std::vector<Chunk> chunks;
    for (int cy = 0; cy < ty; cy += ch)
        for (int cx = 0; cx < tx; cx += cw) {
            // output pixel of the chunk's centre -> frame pixel (svr_frame mapping)
            const int ox = std::min(P.frame.out_w - 1, (cx + cw / 2) * tile_w + tile_w / 2);
            const int oy = std::min(P.frame.out_h - 1, (cy + ch / 2) * tile_h + tile_h / 2);
            const float fx = (float)(P.frame.x0 + ox) + 0.5f;
            const float fy = (float)(P.frame.y0 + (oy / P.frame.band_h) * P.frame.band_pitch + oy % P.frame.band_h) + 0.5f;
            chunks.push_back({ ray_cost(P, fx, fy), cx, cy });
        }
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& a, const Chunk& b) { return a.cost > b.cost; });
    const int n = tx * ty;
    std::vector<uint32_t> lists[8];
    for (size_t i = 0; i < chunks.size(); ++i) {
        const int round = (int)(i / 8), pos = (int)(i % 8), k = (round & 1) ? 7 - pos : pos;
        for (int y = chunks[i].cy; y < std::min(chunks[i].cy + ch, ty); ++y)
            for (int x = chunks[i].cx; x < std::min(chunks[i].cx + cw, tx); ++x) lists[k].push_back((uint32_t)(y * tx + x));
    }
    // XCD k runs blocks k, k + 8, ...: exactly ceil((n - k) / 8) of them; surplus tiles (the cheapest: list tails) move over
    std::vector<uint32_t> spare;
    for (int k = 0; k < 8; ++k) {
        const size_t need = n > k ? (size_t)((n - k + 7) / 8) : 0;
        while (lists[k].size() > need) { spare.push_back(lists[k].back()); lists[k].pop_back(); }
    }
    for (int k = 0; k < 8; ++k) {
        const size_t need = n > k ? (size_t)((n - k + 7) / 8) : 0;
        while (lists[k].size() < need) { lists[k].push_back(spare.back()); spare.pop_back(); }
    }
    order.assign((size_t)n, 0u);
    for (int k = 0; k < 8; ++k)
        for (size_t j = 0; j < lists[k].size(); ++j) order[8 * j + (size_t)k] = lists[k][j];
}

static int tile_order_for(svr_ctx* c, int tx, int ty, int tile_w, int tile_h, int mode, const uint32_t** out) {
    *out = nullptr;
    if (mode == 1 || tx * ty <= 0) return SVR_OK;             // in-kernel contiguous mapping
    // chunk size in pixels per policy
    // (policy 7: the camera-independent 64x64 table of round 1 — what policy 0 falls back to with SVR_STATIC_PLACEMENT)
    static const int kChunk[8][2] = { { 64, 64 }, { 0, 0 }, { 1, 1 }, { 64, 32 }, { 32, 32 }, { 128, 64 }, { 32, 16 }, { 64, 64 } };
    const int cw = std::max(1, kChunk[mode][0] / tile_w), ch = std::max(1, kChunk[mode][1] / tile_h);
    const int key = mode | cw << 8 | ch << 16;
    for (const auto& t : c->tile_orders)
        if (t.tiles_x == tx && t.tiles_y == ty && t.mode == key) { *out = t.dev; return SVR_OK; }
    std::vector<uint32_t> order;
    build_tile_order(tx, ty, cw, ch, order);
    DeviceGuard guard(c->device);
    uint32_t* dev = nullptr;
    SVR_HIP_TRY(hipMalloc((void**)&dev, order.size() * sizeof(uint32_t)));
    SVR_HIP_TRY(hipMemcpy(dev, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (c->tile_orders.size() >= 64) {                        // bounded cache: drop the oldest table once idle
        (void)hipDeviceSynchronize();
        (void)hipFree(c->tile_orders.front().dev);
        c->tile_orders.erase(c->tile_orders.begin());
    }
    c->tile_orders.push_back({ tx, ty, key, dev });
    *out = dev;
    return SVR_OK;
}

// The cost-sorted table of this draw: kept per stream, recomputed only when the camera / frame region changed.
// The new table is written into the stream's pinned buffer and copied on that same stream, i.e. behind the
// previous draw of that stream (which may still read the old table) and before this one.
static int tile_order_by_cost_for(svr_ctx* c, const MarchParams& P, int tile_w, int tile_h, hipStream_t stream,
                                  const uint32_t** out) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
    mix(P.ndc_to_data, sizeof(P.ndc_to_data)); mix(P.size, sizeof(P.size)); mix(&P.frame, sizeof(P.frame));
    mix(&tile_w, sizeof(tile_w)); mix(&tile_h, sizeof(tile_h)); mix(&P.tiles_x, sizeof(P.tiles_x)); mix(&P.tiles_y, sizeof(P.tiles_y));
    svr_ctx::CostOrder* slot = nullptr;
    for (auto& t : c->cost_orders) if (t.stream == stream) { slot = &t; break; }
    DeviceGuard guard(c->device);
    if (!slot) {
        if (c->cost_orders.size() >= 64) {                   // a caller cycling through many streams: recycle the oldest slot
            // (wait for the last draw that read the old table, through the slot's own event: the stream it ran on may
            // be gone by now)
            svr_ctx::CostOrder old = c->cost_orders.front();
            if (old.drawn_set) SVR_HIP_TRY(hipEventSynchronize(old.drawn));
            else if (old.valid) SVR_HIP_TRY(hipEventSynchronize(old.copied));
            c->cost_orders.erase(c->cost_orders.begin());
            old.stream = stream; old.valid = false; old.drawn_set = false;
            c->cost_orders.push_back(old);
        } else {
            svr_ctx::CostOrder t{ stream, nullptr, nullptr, 0, nullptr, 0, false, nullptr, false };
            SVR_HIP_TRY(hipEventCreateWithFlags(&t.copied, hipEventDisableTiming));
            SVR_HIP_TRY(hipEventCreateWithFlags(&t.drawn, hipEventDisableTiming));
            c->cost_orders.push_back(t);
        }
        slot = &c->cost_orders.back();
    }
    const size_t n = (size_t)P.tiles_x * (size_t)P.tiles_y;
    if (slot->valid && slot->key == h && slot->cap >= n) { *out = slot->dev; return SVR_OK; }
    if (slot->cap < n) {
        if (slot->dev) { SVR_HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(slot->dev); (void)hipHostFree(slot->host); slot->dev = slot->host = nullptr; slot->cap = 0; }
        const size_t cap = n + n / 4 + 64;
        if (hipMalloc((void**)&slot->dev, cap * sizeof(uint32_t)) != hipSuccess ||
            hipHostMalloc((void**)&slot->host, cap * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
            svr_set_error("svr_render: out of memory for the block -> tile table"); return SVR_ERR_NOMEM;
        }
        slot->cap = cap;
    } else if (slot->valid) {
        SVR_HIP_TRY(hipEventSynchronize(slot->copied));      // the pinned buffer's previous contents have left for the device
    }
    std::vector<uint32_t> order;
    build_tile_order_by_cost(P, P.tiles_x, P.tiles_y, tile_w, tile_h, std::max(1, 64 / tile_w), std::max(1, 64 / tile_h), order);
    memcpy(slot->host, order.data(), n * sizeof(uint32_t));
    SVR_HIP_TRY(hipMemcpyAsync(slot->dev, slot->host, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    SVR_HIP_TRY(hipEventRecord(slot->copied, stream));
    slot->key = h; slot->valid = true;
    *out = slot->dev;
    return SVR_OK;
}

// Can the span kernel address this context's rings?  Every LOD's ring must stay below 4 GiB (32-bit byte offsets
// into its buffer resource) and within the 24-bit row index and row pitch of the exact general path; otherwise
// every render takes the straightforward kernel's 64-bit addressing.
static bool span_addressable(const svr_ctx* c) {
    const uint64_t des = svr_dtype_size(c->density_storage);
    for (int l = 0; l < c->num_lods; ++l) {
        // (a ring of 4 GiB or more is reached through up to 8 resources of whole z planes, each below 4 GiB: 32 GiB)
        const uint64_t plane = (uint64_t)c->lod[l].ring[0] * (uint64_t)c->lod[l].ring[1] * des;
        uint64_t zsplit = plane ? std::min<uint64_t>((uint64_t)c->lod[l].ring[2], (((uint64_t)1 << 32) - 128) / plane) : 0;
        if (zsplit < (uint64_t)c->lod[l].ring[2] && zsplit >= 4) zsplit &= ~(uint64_t)3;      // (parts of whole micro-blocks: fill_params)
        if (zsplit == 0 || ((uint64_t)c->lod[l].ring[2] + zsplit - 1) / zsplit > 8 ||
            (uint64_t)c->lod[l].ring[1] * (uint64_t)c->lod[l].ring[2] >= (1u << 24) ||
            (uint64_t)c->lod[l].ring[0] * des >= (1u << 24)) return false;
    }
    return true;
}

// Rings of 4 GiB or more in total take the BIG build of the march (one buffer resource per LOD, several for a ring of
// 4 GiB or more).  -DSVR_EXPERIMENTS builds can force that path onto small rings (SVR_FORCE_BIG=1) and cut them into
// parts of SVR_FORCE_ZSPLIT planes, so that the fuzzer reaches the multi-resource addressing with part boundaries everywhere.
static bool rings_need_big(const svr_ctx* c) {
    static const bool force_big = svr_exp_env_int("SVR_FORCE_BIG", 0) != 0;
    return c->density_all_bytes >= ((size_t)1 << 32) || force_big;
}

// Behind a draw that may have read a cost-sorted table: mark the stream's slot (see CostOrder::drawn).
static int cost_order_drawn(svr_ctx* c, const MarchParams& P, hipStream_t stream) {
    for (auto& t : c->cost_orders)
        if (t.stream == stream && t.dev == P.tile_order) {
            SVR_HIP_TRY(hipEventRecord(t.drawn, stream));
            t.drawn_set = true;
            break;
        }
    return SVR_OK;
}

static int fill_params(svr_ctx* c, const svr_camera* cam, const svr_frame* fr, const svr_outputs* out, hipStream_t stream,
                       MarchParams& P) {
    SVR_REQUIRE(c && cam && fr && out && out->rgba, "svr_render: null argument");
    SVR_REQUIRE(c->material_set, "svr_render: svr_set_material has not been called");
    SVR_REQUIRE(fr->frame_w > 0 && fr->frame_h > 0 && fr->out_w > 0 && fr->out_h > 0, "svr_render: empty frame");
    SVR_REQUIRE(fr->x0 >= 0 && fr->y0 >= 0, "svr_render: negative tile origin");
    memset(&P, 0, sizeof(P));
    float tmp[16];
    mat_mul4(cam->world_inv, cam->cam_inv, tmp);            // vs_main.wgsl:22 (left-assoc)
    mat_mul4(tmp, cam->proj_inv, P.ndc_to_data);
    mat_mul4(cam->proj, cam->cam, P.pc);                    // vs_main.wgsl:19
    memcpy(P.world, cam->world, sizeof(P.world));
    for (int a = 0; a < 3; ++a) {
        SVR_REQUIRE(cam->volume_dimensions[a] >= 1.0f, "svr_render: volume_dimensions must be >= 1");
        P.size[a] = cam->volume_dimensions[a];
    }
    const float mx = fmaxf(P.size[0], fmaxf(P.size[1], P.size[2]));
    P.rel_step = fminf(fmaxf(sqrtf(mx) / 20.0f, 0.1f), 0.8f);   // fs_main.wgsl:20
    P.frame = *fr;
    if (P.frame.band_h <= 0) { P.frame.band_h = fr->out_h; P.frame.band_pitch = fr->out_h; }
    const svr_material& m = c->material;
    P.clim0 = m.clim[0]; P.clim1 = m.clim[1]; P.gamma = m.gamma; P.opacity = m.opacity;
    P.lmip_threshold = m.lmip_threshold; P.lmip_fall_off = m.lmip_fall_off;
    {   // integer rings: "(f32)v >= threshold" as an integer compare; NaN or beyond the largest value: never
        const float vmax = c->density_storage == SVR_U16 ? 65535.0f : 255.0f;
        P.lmip_threshold_raw = (uint32_t)vmax + 1u;
        if (m.lmip_threshold <= 0.0f) P.lmip_threshold_raw = 0u;
        else if (m.lmip_threshold <= vmax) P.lmip_threshold_raw = (uint32_t)ceilf(m.lmip_threshold);
    }
    P.clip_count = m.clipping_plane_count; P.clip_all = m.clipping_mode_all;
    for (uint32_t k = 0; k < m.clipping_plane_count; ++k)
        for (int a = 0; a < 4; ++a) P.clip[k][a] = c->clip_host[4 * k + a];
    P.lmip_max_samples = m.lmip_max_samples; P.fog_density = m.fog_density;
    P.render_mode = m.render_mode; P.weight_falloff = m.weight_falloff;
    for (int a = 0; a < 3; ++a) P.fog_color[a] = m.fog_color[a];
    P.color_count = m.color_count; P.colors = c->colors_dev; P.colorspace_srgb = m.colorspace_srgb;
    P.num_lods = c->num_lods;
    if (out->steps && !c->dbg_dev && hipMalloc((void**)&c->dbg_dev, 40 * sizeof(uint32_t)) == hipSuccess)
        (void)hipMemset(c->dbg_dev, 0, 40 * sizeof(uint32_t));      // 8 census counters + 16 u64 cycle sums
    P.dbg = out->steps ? c->dbg_dev : nullptr;
    P.rgba = out->rgba; P.depth = out->depth; P.label = out->label; P.flags = out->flags; P.steps = out->steps;
    P.pick = reinterpret_cast<unsigned long long*>(out->pick); P.pick_id = out->pick_id;
    // variant: bits 0-3 kernel kind (0 batched U=8, 1 simple, 2 batched U=4), bits 4-7 = 1 + log2 of the
    // wave tile width (0 = default 8x8), bit 8 = disable the LDS brick path
    int lw = ((c->variant >> 4) & 15) ? ((c->variant >> 4) & 15) - 1 : 3;
    if (lw > 6) lw = 6;
    // the weighted-average mode rides the span march (another reducer over the same batches); rings the span march
    // cannot address, and variant 1, take march_wavg, which is laid out like the simple kernel
    const bool wavg_simple = m.render_mode == SVR_MODE_WEIGHTED_AVERAGE &&
                             ((c->variant & 3) == 1 || !span_addressable(c) || rings_need_big(c));
    if ((c->variant & 3) == 1 || wavg_simple) lw = 3;         // the simple kernels are 8x8 only
    P.tile_log2w = lw;
    P.brick = (c->variant & 256) ? 0 : ((c->variant & 512) ? 2 : 1);     // never / always / auto (per wave)
    P.brick_lines = ((c->variant >> 16) & 0xFF) ? ((c->variant >> 16) & 0xFF) : 32;
    P.orient = (c->variant & 1024) ? 0 : 1;
    P.dbg_nowait = ((c->variant & SVR_EXP_VARIANT_BITS) >> 11) & 3;      // -DSVR_EXPERIMENTS builds: bit 11 no brick wait, bit 12 skip the march loop
    P.brick_lod_mask = ((c->variant >> 24) & 0xFF) ? ((c->variant >> 24) & 0xFF) : 0xFF;
    const int brick_mask = P.brick_lod_mask;
    // Every wave starts with brick slabs of twice the plain length (the staged bytes per sample fall with the
    // slab length) and drops to the plain length at its first box that does not fit; variant bit 2: never long.
    P.slab_long = (c->variant & 4) ? 0 : 1;
    static const bool brick_pow2 = svr_exp_env_set("SVR_BRICK_POW2");         // A/B measurements (-DSVR_EXPERIMENTS builds)
    P.brick_pow2 = brick_pow2 ? 1 : 0;
    // LDS per wave: 8 KiB holds the box of a 16..64-iteration slab of byte voxels and lets 20 one-wave blocks share
    // a CU's 160 KiB.  2- and 4-byte voxels fit shorter slabs in the same 8 KiB; 16 KiB for them (10 waves per CU)
    // measured slower: K1 full, f32 rings 1.60 ms against 1.31 ms (SVR_BRICK_BYTES: experiments)
    static const int brick_bytes_env = svr_exp_env_int("SVR_BRICK_BYTES", 0);
    P.brick_bytes = brick_bytes_env >= 1024 && brick_bytes_env <= 65536 ? (brick_bytes_env & ~15) : 8192;
    {   // vanishing point of the volume's x axis: (proj*cam) * (world * (1,0,0,0))
        const float ex[4] = { 1.0f, 0.0f, 0.0f, 0.0f };
        float wx[4];
        mat_vec4(cam->world, ex, wx);
        mat_vec4(P.pc, wx, P.xdir);
    }
    // kernel kind 0: span march, one wave per block; 2: span march, 2 x 2 waves per block; 1: simple (2 x 2)
    // (rings of 4 GiB or more fall back to the simple kernel's 64-bit addressing, see launch_nl)
    P.block_waves_log2 = ((c->variant & 3) == 0 && span_addressable(c) && !wavg_simple) ? 0 : 1;
    const int bw = (1 << P.block_waves_log2) << lw, bh = (1 << P.block_waves_log2) * (64 >> lw);
    P.tiles_x = (fr->out_w + bw - 1) / bw; P.tiles_y = (fr->out_h + bh - 1) / bh;
    {   // variant bits 13-15: block -> tile policy (0 default = 64x64-pixel chunks dealt to the XCDs, 1 contiguous, 2.. other chunks)
        const int rc_order = tile_order_for(c, P.tiles_x, P.tiles_y, bw, bh, (c->variant >> 13) & 7, &P.tile_order);
        if (rc_order) return rc_order;
        // default placement: the same 64x64-pixel chunks, but sorted by the length of their rays for THIS camera
        // (longest first, snake-dealt to the XCDs); SVR_STATIC_PLACEMENT keeps the camera-independent table (A/B)
        static const bool static_placement = svr_exp_env_set("SVR_STATIC_PLACEMENT");
        if (!static_placement && ((c->variant >> 13) & 7) == 0 && P.tiles_x * P.tiles_y > 0) {
            const int rc_lpt = tile_order_by_cost_for(c, P, bw, bh, stream, &P.tile_order);
            if (rc_lpt) return rc_lpt;
        }
    }
    for (int l = 0; l < c->num_lods; ++l) {
        const LodStorage& L = c->lod[l];
        LodParams& Q = P.lod[l];
        Q.density = L.density; Q.labels = L.labels;
        for (int a = 0; a < 3; ++a) {
            Q.off[a] = L.state.offset[a];
            Q.shape[a] = (uint32_t)L.state.shape[a];
            Q.ring[a] = (uint32_t)L.ring[a];
            Q.wrap0[a] = (uint32_t)(L.state.offset[a] - floor_div(L.state.offset[a], L.ring[a]) * L.ring[a]);
            Q.scale[a] = L.state.scale[a];
            Q.addw[a] = (int32_t)Q.wrap0[a] - Q.off[a];
        }
        {
            bool ok = true;
            for (int a = 0; a < 3; ++a) {
                int e = 0;
                const float m = frexpf(Q.scale[a], &e);
                ok = ok && m == 0.5f && Q.scale[a] > 0.0f && Q.scale[a] <= 1.0f;    // exactly 2^-k
                ok = ok && P.size[a] * Q.scale[a] < 8388608.0f;                      // indices < 2^23
                ok = ok && Q.ring[a] < (1u << 15);                                   // packed i16 brick boxes
            }
            ok = ok && (double)P.size[1] * Q.scale[1] * Q.ring[1] < 16777216.0;      // row index < 2^24
            P.lod_pow2[l] = ok ? 1 : 0;
        }
        const uint32_t des = (uint32_t)svr_dtype_size(c->density_storage);
        Q.rx4 = Q.ring[0] * des;
        Q.base_bytes = (uint32_t)(c->lod_base_bytes[l] * des);
        for (int a = 0; a < 3; ++a) Q.ss[a] = P.size[a] * Q.scale[a];          // one IEEE multiply, as the kernel did
        {   // brick slabs (u8 rings): about 12 ring voxels of travel per slab (coarser LODs advance less per
            // iteration); rows in 16-byte groups, indices below 2^15 for the packed (y, z) brick address
            const float smax = fmaxf(Q.scale[0], fmaxf(Q.scale[1], Q.scale[2]));
            static const int slab_shift = svr_exp_env_int("SVR_SLAB_SHIFT", 0);     // A/B measurements (-DSVR_EXPERIMENTS builds)
            const int slab = std::max(8, (smax > 0.75f ? 16 : (smax > 0.375f ? 32 : 64)) >> slab_shift);
            // (ring rows must be whole 16-byte groups: 16 / 8 / 4 voxels for u8 / u16 / f32 rings)
            bool ok = ((Q.ring[0] * des) & 15u) == 0u && (brick_mask >> l & 1);
            for (int a = 0; a < 3; ++a) ok = ok && (long long)Q.off[a] + (long long)Q.shape[a] < 32768;
            Q.slab = ok ? slab : 0;
            // empty-space skipping: cell tests every `skip_batches` batches of 8 iterations, such that a ray
            // (0.8 voxels of the finest level per iteration, fs_main.wgsl:20) travels at most one cell between two
            // tests on every axis (checked again per lane in the kernel): 8^3 cells at scale 1 and 4^3 cells at scale
            // 1/2 -> every batch; 4^3 cells at scale 1/4 -> every other batch
            Q.cell_base = (uint32_t)(L.cell_base * des);
            for (int a = 0; a < 3; ++a) Q.cdim[a] = (uint32_t)L.cdim[a];
            Q.cshift = L.cshift;
            Q.skip_batches = 0;
            if (L.cells_dil && P.lod_pow2[l]) {
                const float room = (float)(1 << L.cshift) - 0.5f, per_batch = smax * 8.0f * P.rel_step;
                Q.skip_batches = per_batch * 4.0f <= room ? 4 : (per_batch * 2.0f <= room ? 2 : (per_batch <= room ? 1 : 0));
            }
        }
    }
    // skip only when some texel value can reach the threshold and not every one does: threshold = +inf (or NaN)
    // is the "full" march, whose point is the traversal itself; variant bit 3 switches skipping off (A/B)
    // A state machine that can never stop (no fall-off, no sample limit: MIP, _material.py lmip_uniforms) is the running
    // maximum of the ray; a lane that follows one can then pass every block that cannot beat it, like an empty one.
    const bool mip_like = m.lmip_fall_off == 0.0f && m.lmip_max_samples == INT32_MAX;
    const bool skip = c->cells_dil_all && !(c->variant & 8) && m.render_mode == SVR_MODE_LMIP &&       // (a mean needs every sample)
                      (mip_like || (P.lmip_threshold_raw > 0u &&
                      (c->density_storage == SVR_F32 ? (m.lmip_threshold > 0.0f && m.lmip_threshold < INFINITY)
                                                     : P.lmip_threshold_raw <= (c->density_storage == SVR_U16 ? 65535u : 255u))));
    static const int skip_flags_env = svr_exp_env_int("SVR_SKIP_FLAGS", 1);      // A/B measurements (-DSVR_EXPERIMENTS builds)
    P.skip_flags = (skip_flags_env & 1) | (mip_like ? 2 : 0);
    P.cells_all = skip ? c->cells_dil_all : nullptr;
    P.cells_all_bytes = skip ? (uint32_t)c->cells_all_bytes : 0u;
    if (!brick_bytes_env && !skip) {
        // When a pixel is wider than about 1.5 finest-level voxels at the volume's centre (a 2048^3 volume seen
        // whole at 1080p: 1.8), the 64 rays of a wave are that far apart and the box of even a short slab outgrows
        // 8 KiB; 16 KiB per wave (10 waves per CU) then wins: config 5, K1, full mode 5.28 -> 4.70 ms.  Not while
        // empty-space skipping is active: its cell tests want the occupancy (config 5 LMIP: 0.80 ms at 8 KiB, 1.08 at 16)
        // (32 KiB: 5.61, 64 KiB: 9.35; at config 2's 0.9 voxels per pixel 16 KiB loses 20 %).
        const float centre[4] = { 0.5f * P.size[0] - 0.5f, 0.5f * P.size[1] - 0.5f, 0.5f * P.size[2] - 0.5f, 1.0f };
        float w[4], q[4];
        mat_vec4(cam->world, centre, w);
        mat_vec4(P.pc, w, q);
        if (q[3] > 0.0f) {
            const float a[4] = { q[0] / q[3], q[1] / q[3], q[2] / q[3], 1.0f };
            const float b[4] = { a[0] + 2.0f / (float)fr->frame_w, a[1], a[2], 1.0f };
            float da[4], db[4];
            mat_vec4(P.ndc_to_data, a, da);
            mat_vec4(P.ndc_to_data, b, db);
            float d2 = 0.0f;
            for (int k = 0; k < 3; ++k) { const float d = db[k] / db[3] - da[k] / da[3]; d2 += d * d; }
            if (d2 >= 2.25f && d2 < 1e12f) P.brick_bytes = 16384;
        }
    }
    P.density_all = c->density_all;
    P.span_ok = span_addressable(c) ? 1 : 0;
    P.per_lod_rsrc = rings_need_big(c) ? 1 : 0;
    P.density_all_bytes = P.per_lod_rsrc ? 0u : (uint32_t)c->density_all_bytes;
    for (int l = 0; l < c->num_lods; ++l) {
        LodParams& Q = P.lod[l];
        Q.nparts = 1u; Q.zsplit = Q.ring[2]; Q.part_bytes = 0u; Q.rbytes_last = 0u;
        // the micro-block copy of the ring (svr_lod_desc::blocked_twin): always behind a resource of its own, which is cut
        // into parts exactly like the ring's where that is (parts of whole blocks: zsplit is a multiple of 4 planes)
        // (the march addresses it from the packed voxel index y | z << 16: both below 2^16, whatever the ray)
        Q.twin = (c->lod[l].twin && P.size[1] * Q.scale[1] < 65536.0f && P.size[2] * Q.scale[2] < 65536.0f) ? (uint32_t)c->lod[l].twin_policy : 0u;
        Q.twin_rbase = c->lod[l].twin;
        Q.twin_rbytes = (uint32_t)std::min<uint64_t>((uint64_t)c->lod[l].voxels * svr_dtype_size(c->density_storage) + 64, 0xFFFFFFFFull);
        if (P.per_lod_rsrc) {
            const uint64_t des64 = svr_dtype_size(c->density_storage);
            const uint64_t plane = (uint64_t)Q.ring[0] * (uint64_t)Q.ring[1] * des64, bytes = (uint64_t)c->lod[l].voxels * des64;
            Q.rbase = c->lod[l].density; Q.base_bytes = 0u;
            static const int force_zsplit = svr_exp_env_int("SVR_FORCE_ZSPLIT", 0);      // (see rings_need_big)
            if (force_zsplit > 0 && Q.ring[2] > 1u) {
                Q.zsplit = std::max<uint32_t>((uint32_t)force_zsplit, (Q.ring[2] + 7u) / 8u);
                if (Q.twin) Q.zsplit = (Q.zsplit + 3u) & ~3u;
                Q.zsplit = std::min<uint32_t>(Q.zsplit, Q.ring[2]);
                Q.nparts = (Q.ring[2] + Q.zsplit - 1u) / Q.zsplit;
                Q.part_bytes = Q.rbytes = (uint32_t)((uint64_t)Q.zsplit * plane);
                Q.rbytes_last = (uint32_t)(bytes - (uint64_t)(Q.nparts - 1) * Q.part_bytes + 64);
                if (Q.nparts == 1u) { Q.rbytes = Q.rbytes_last; Q.part_bytes = 0u; }
            } else if (bytes + 64 < ((uint64_t)1 << 32)) {
                Q.rbytes = Q.rbytes_last = (uint32_t)(bytes + 64);
            } else {                                         // parts of whole ring z planes (span_addressable checked the count)
                Q.zsplit = (uint32_t)std::min<uint64_t>((uint64_t)Q.ring[2], (((uint64_t)1 << 32) - 128) / plane);
                if (Q.zsplit < Q.ring[2] && Q.zsplit >= 4u) Q.zsplit &= ~3u;
                Q.nparts = (uint32_t)(((uint64_t)Q.ring[2] + Q.zsplit - 1) / Q.zsplit);
                Q.part_bytes = Q.rbytes = (uint32_t)((uint64_t)Q.zsplit * plane);
                Q.rbytes_last = (uint32_t)(bytes - (uint64_t)(Q.nparts - 1) * Q.part_bytes + 64);
            }
            Q.twin_rbytes = Q.rbytes;                       // (a full part's size where the ring is cut; part_rsrc sizes the last one)
        } else {
            Q.rbase = c->density_all; Q.rbytes = P.density_all_bytes;
        }
    }
    P.density_esh = c->density_storage == SVR_U8 ? 0 : (c->density_storage == SVR_U16 ? 1 : 2);
    return SVR_OK;
}

int svr_render(svr_ctx* c, const svr_camera* cam, const svr_frame* fr, const svr_outputs* out, void* stream) {
    MarchParams P;
    int rc = fill_params(c, cam, fr, out, static_cast<hipStream_t>(stream), P);
    if (rc) return rc;
    DeviceGuard guard(c->device);
    // the caller's stream; NULL is the device's default stream (what torch.cuda.current_stream() is
    // unless the caller switched streams), so the draw is ordered with the caller's other work on it
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c->have_published) SVR_HIP_TRY(hipStreamWaitEvent(s, c->uploads_published, 0));
    SVR_HIP_TRY(svr_launch_march(P, (c->variant & 3) == 1 ? 1 : 0, s));
    { const int rco = cost_order_drawn(c, P, s); if (rco) return rco; }
    return mark_render(c, s);
}

int svr_time_render(svr_ctx* c, const svr_camera* cam, const svr_frame* fr, const svr_outputs* out,
                    int iters, float* avg_ms) {
    SVR_REQUIRE(avg_ms && iters >= 1, "svr_time_render: bad arguments");
    MarchParams P;
    int rc = fill_params(c, cam, fr, out, c->render_stream, P);
    if (rc) return rc;
    DeviceGuard guard(c->device);
    hipStream_t s = c->render_stream;
    if (c->have_published) SVR_HIP_TRY(hipStreamWaitEvent(s, c->uploads_published, 0));
    SVR_HIP_TRY(hipEventRecord(c->ev_a, s));
    for (int i = 0; i < iters; ++i) SVR_HIP_TRY(svr_launch_march(P, (c->variant & 3) == 1 ? 1 : 0, s));
    SVR_HIP_TRY(hipEventRecord(c->ev_b, s));
    { const int rco = cost_order_drawn(c, P, s); if (rco) return rco; }
    { const int rcm = mark_render(c, s); if (rcm) return rcm; }
    SVR_HIP_TRY(hipEventSynchronize(c->ev_b));
    float ms = 0.f;
    SVR_HIP_TRY(hipEventElapsedTime(&ms, c->ev_a, c->ev_b));
    *avg_ms = ms / (float)iters;
    return SVR_OK;
}

int svr_untile_stripes(svr_ctx* c, const void* gathered, void* frame_out, int frame_w, int frame_h,
                       int band_h, int nranks, int out_h, int elem_bytes, void* stream) {
    SVR_REQUIRE(c && gathered && frame_out, "svr_untile_stripes: null argument");
    SVR_REQUIRE(frame_w > 0 && frame_h > 0 && band_h > 0 && nranks > 0, "svr_untile_stripes: bad geometry");
    const int nbands = (frame_h + band_h - 1) / band_h;
    SVR_REQUIRE(out_h >= ((nbands + nranks - 1) / nranks) * band_h, "svr_untile_stripes: out_h too small for the frame");
    DeviceGuard guard(c->device);
    hipStream_t s = static_cast<hipStream_t>(stream);       // NULL = default stream, as in svr_render
    SVR_HIP_TRY(svr_launch_untile(gathered, frame_out, frame_w, frame_h, band_h, nranks, out_h, elem_bytes, s));
    return SVR_OK;
}

int svr_untile_grid(svr_ctx* c, const void* gathered, void* frame_out, int frame_w, int frame_h,
                    int tile_w, int tile_h, int grid_x, int grid_y, int elem_bytes, void* stream) {
    SVR_REQUIRE(c && gathered && frame_out, "svr_untile_grid: null argument");
    SVR_REQUIRE(frame_w > 0 && frame_h > 0 && tile_w > 0 && tile_h > 0 && grid_x > 0 && grid_y > 0, "svr_untile_grid: bad geometry");
    SVR_REQUIRE((int64_t)tile_w * grid_x >= frame_w && (int64_t)tile_h * grid_y >= frame_h, "svr_untile_grid: the tiles do not cover the frame");
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(svr_launch_untile_grid(gathered, frame_out, frame_w, frame_h, tile_w, tile_h, grid_x, grid_y, elem_bytes,
                                       static_cast<hipStream_t>(stream)));
    return SVR_OK;
}

int svr_pool2x(int device, const void* src, void* dst, const int32_t src_dims[3], int dtype, int mode, void* stream) {
    SVR_REQUIRE(src && dst && src_dims, "svr_pool2x: null argument");
    for (int a = 0; a < 3; ++a)
        SVR_REQUIRE(src_dims[a] >= 2 && (src_dims[a] & 1) == 0, "svr_pool2x: every source extent must be even and >= 2");
    SVR_REQUIRE((dtype == SVR_U8 && mode == 0) || (dtype == SVR_F32 && mode == 0) || (dtype == SVR_U32 && mode == 1),
                "svr_pool2x: supported: mean of u8 / f32 (mode 0), max of u32 (mode 1)");
    DeviceGuard guard(device);
    SVR_HIP_TRY(svr_launch_pool2x(src, dst, src_dims, dtype, mode, static_cast<hipStream_t>(stream)));
    return SVR_OK;
}

int svr_compose(svr_ctx* c, const float* rgba, const float* depth, const uint8_t* flags, int width, int height,
                const svr_compose_params* params, uint8_t* out_rgba8, float* inout_depth, void* stream) {
    SVR_REQUIRE(c && rgba && params && out_rgba8, "svr_compose: null argument");
    SVR_REQUIRE(width >= 0 && height >= 0, "svr_compose: negative size");
    SVR_REQUIRE(!inout_depth || depth, "svr_compose: a depth test needs the render's depth plane");
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(svr_launch_compose(rgba, depth, flags, width, height, *params, out_rgba8, inout_depth,
                                   static_cast<hipStream_t>(stream)));
    return SVR_OK;
}

int svr_sync(svr_ctx* c) {
    SVR_REQUIRE(c, "svr_sync: null ctx");
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(hipStreamSynchronize(c->upload_stream));
    SVR_HIP_TRY(hipStreamSynchronize(c->render_stream));
    return SVR_OK;
}

int svr_sync_uploads(svr_ctx* c) {
    SVR_REQUIRE(c, "svr_sync_uploads: null ctx");
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(hipStreamSynchronize(c->upload_stream));
    return SVR_OK;
}

int svr_debug_counters(svr_ctx* c, uint32_t out[8], int reset) {
    SVR_REQUIRE(c && out, "svr_debug_counters: null argument");
    memset(out, 0, 8 * sizeof(uint32_t));
    if (!c->dbg_dev) return SVR_OK;
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(hipDeviceSynchronize());
    SVR_HIP_TRY(hipMemcpy(out, c->dbg_dev, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (reset) SVR_HIP_TRY(hipMemset(c->dbg_dev, 0, 8 * sizeof(uint32_t)));
    return SVR_OK;
}

int svr_debug_timers(svr_ctx* c, uint64_t out[16], int reset) {
    SVR_REQUIRE(c && out, "svr_debug_timers: null argument");
    memset(out, 0, 16 * sizeof(uint64_t));
    if (!c->dbg_dev) return SVR_OK;
    DeviceGuard guard(c->device);
    SVR_HIP_TRY(hipDeviceSynchronize());
    SVR_HIP_TRY(hipMemcpy(out, c->dbg_dev + 8, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) SVR_HIP_TRY(hipMemset(c->dbg_dev + 8, 0, 16 * sizeof(uint64_t)));
    return SVR_OK;
}

int svr_lod_twin_ptr(svr_ctx* c, int lod, void** twin) {
    SVR_REQUIRE(c && twin, "svr_lod_twin_ptr: null argument");
    SVR_REQUIRE(lod >= 0 && lod < c->num_lods, "svr_lod_twin_ptr: lod out of range");
    *twin = c->lod[lod].twin;
    return SVR_OK;
}

int svr_lod_device_ptrs(svr_ctx* c, int lod, void** density, void** labels) {
    SVR_REQUIRE(c, "svr_lod_device_ptrs: null ctx");
    SVR_REQUIRE(lod >= 0 && lod < c->num_lods, "svr_lod_device_ptrs: lod out of range");
    if (density) *density = c->lod[lod].density;
    if (labels) *labels = c->lod[lod].labels;
    return SVR_OK;
}

}  // extern "C"
