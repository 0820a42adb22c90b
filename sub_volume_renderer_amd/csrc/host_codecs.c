/*
 * host_codecs.c — byte-level helpers of the chunk-store readers (sub_volume_renderer_amd/zarr3.py) that are too slow
 * in Python at chunk sizes: CRC-32C (Castagnoli), the checksum the zarr v3 `crc32c` codec appends to a chunk or a
 * shard index.  Slicing-by-8, table driven, no ISA extensions (the same bytes on any host).
 *
 * Build: gcc -O3 -fPIC -shared (see __graft_entry__.build_host_codecs).  zarr3.py falls back to its Python loop when
 * the library has not been built.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "../../include/svr_host_codecs.h"

static uint32_t T[8][256];
static int ready;

static void init(void) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int k = 1; k < 8; ++k) T[k][i] = (T[k - 1][i] >> 8) ^ T[0][T[k - 1][i] & 0xFFu];
    ready = 1;
}

/* CRC-32C of n bytes, continuing from `crc` (0 for a fresh checksum): init and final xor with 0xFFFFFFFF. */
uint32_t svr_crc32c(const void* data, size_t n, uint32_t crc) {
    if (!ready) init();
    const uint8_t* p = (const uint8_t*)data;
    crc = ~crc;
    while (n && ((uintptr_t)p & 7u)) { crc = T[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8); --n; }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);                       /* little-endian hosts only (x86-64, the GPU box) */
        w ^= crc;
        crc = T[7][w & 0xFF] ^ T[6][(w >> 8) & 0xFF] ^ T[5][(w >> 16) & 0xFF] ^ T[4][(w >> 24) & 0xFF] ^
              T[3][(w >> 32) & 0xFF] ^ T[2][(w >> 40) & 0xFF] ^ T[1][(w >> 48) & 0xFF] ^ T[0][w >> 56];
        p += 8; n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Batch decode / encode of zarr v3 inner chunks (3-D, `bytes` little endian -> [zstd] -> [crc32c]), many chunks per call,
 * a thread pool over chunks: the native half of zarr3.py's reader — what zarr-python / tensorstore do in C++ for the
 * reference (README.md:18, _wrapping_buffer.py:307-322).  libzstd is bound at run time (no headers on the target).
 * --------------------------------------------------------------------------------------------------------------- */
#include <dlfcn.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <stdlib.h>

/* A small persistent thread pool: workers SLEEP on a condition variable between jobs (an OpenMP team spins after
 * every parallel region, and a spinning team beside the render thread burns the cgroup CPU quota of a one-GPU job:
 * 30-60 ms stalls per read were measured that way).  One job at a time (callers are serialised by a mutex); items
 * are claimed one by one from an atomic counter, the calling thread works too. */
#define POOL_MAX 64
typedef void (*pool_fn)(int item, int worker, void* arg);
static struct {
    pthread_mutex_t mu, job_mu;
    pthread_cond_t wake, done;
    pthread_t th[POOL_MAX];
    int nthreads, generation, active, want;      /* want: workers allowed on the current job */
    pool_fn fn; void* arg; int nitems;
    atomic_int next;
    int init;
} pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER };

static void pool_drain(int worker) {
    for (;;) {
        const int k = atomic_fetch_add(&pool.next, 1);
        if (k >= pool.nitems) return;
        pool.fn(k, worker, pool.arg);
    }
}

static int pool_start_gen[POOL_MAX + 1];     /* the generation a worker was created in (set under pool.mu by its creator) */

/* Pin worker `id` to the id-th CPU this process may run on.  A thread woken for a job of a few milliseconds otherwise
 * starts on its waker's CPU and the scheduler does not spread the team before the job is over (measured: 8 threads, one
 * CPU's worth of progress on 60 us items). */
static void pool_pin(int id) {
    const char* e = getenv("SVR_ZARR_PIN");              /* A/B: 0 leaves the workers to the scheduler */
    if (e && e[0] == '0') return;
    cpu_set_t allowed, one;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const int count = CPU_COUNT(&allowed);
    if (count < 2) return;
    int want = id % count, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &allowed) && seen++ == want) {
            CPU_ZERO(&one); CPU_SET(c, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
            return;
        }
}

static void* pool_worker(void* p) {
    const int id = (int)(intptr_t)p;
    pool_pin(id);
    pthread_mutex_lock(&pool.mu);
    int seen = pool_start_gen[id];
    for (;;) {
        while (pool.generation == seen) pthread_cond_wait(&pool.wake, &pool.mu);
        seen = pool.generation;
        if (id > pool.want) continue;                     /* this job runs on fewer threads */
        pthread_mutex_unlock(&pool.mu);
        pool_drain(id);
        pthread_mutex_lock(&pool.mu);
        if (--pool.active == 0) pthread_cond_signal(&pool.done);
    }
    return NULL;
}

/* A forked child inherits the pool's bookkeeping but none of its threads: start over there. */
static void pool_after_fork_in_child(void) {
    pthread_mutex_init(&pool.mu, NULL); pthread_mutex_init(&pool.job_mu, NULL);
    pthread_cond_init(&pool.wake, NULL); pthread_cond_init(&pool.done, NULL);
    pool.nthreads = 0; pool.active = 0; pool.want = 0;
}

/* run fn(item, worker, arg) for item in [0, n) on up to nthreads threads (worker ids 0 .. nthreads - 1; 0 = the caller) */
static void pool_run(int n, int nthreads, pool_fn fn, void* arg) {
    if (nthreads > POOL_MAX) nthreads = POOL_MAX;
    if (nthreads > n) nthreads = n;
    if (nthreads <= 1) { for (int k = 0; k < n; ++k) fn(k, 0, arg); return; }
    pthread_mutex_lock(&pool.job_mu);
    if (!pool.init) { pool.init = 1; pthread_atfork(NULL, NULL, pool_after_fork_in_child); }
    pthread_mutex_lock(&pool.mu);
    while (pool.nthreads < nthreads - 1) {
        const int id = pool.nthreads + 1;
        pool_start_gen[id] = pool.generation;          /* the job about to be posted is new to it */
        if (pthread_create(&pool.th[pool.nthreads], NULL, pool_worker, (void*)(intptr_t)id) != 0) break;
        pthread_detach(pool.th[pool.nthreads]);
        ++pool.nthreads;
    }
    pool.fn = fn; pool.arg = arg; pool.nitems = n;
    atomic_store(&pool.next, 0);
    pool.want = nthreads - 1 < pool.nthreads ? nthreads - 1 : pool.nthreads;
    pool.active = pool.want;
    ++pool.generation;
    pthread_cond_broadcast(&pool.wake);
    pthread_mutex_unlock(&pool.mu);
    pool_drain(0);
    pthread_mutex_lock(&pool.mu);
    while (pool.active > 0) pthread_cond_wait(&pool.done, &pool.mu);
    pthread_mutex_unlock(&pool.mu);
    pthread_mutex_unlock(&pool.job_mu);
}

typedef size_t (*zstd_decompress_t)(void*, size_t, const void*, size_t);
typedef size_t (*zstd_compress_t)(void*, size_t, const void*, size_t, int);
typedef size_t (*zstd_bound_t)(size_t);
typedef unsigned (*zstd_iserror_t)(size_t);
typedef unsigned long long (*zstd_framesize_t)(const void*, size_t);
typedef void* (*zstd_create_t)(void);
typedef size_t (*zstd_free_t)(void*);
typedef size_t (*zstd_decompress_ctx_t)(void*, void*, size_t, const void*, size_t);
typedef size_t (*zstd_compress_ctx_t)(void*, void*, size_t, const void*, size_t, int);
static zstd_create_t z_create_dctx, z_create_cctx;            /* one context per worker and call: ZSTD_decompress() builds */
static zstd_free_t z_free_dctx, z_free_cctx;                  /* and frees a context per chunk (20 us for a 4 KiB chunk)     */
static zstd_decompress_ctx_t z_decompress_ctx;
static zstd_compress_ctx_t z_compress_ctx;
static zstd_decompress_t z_decompress;
static zstd_compress_t z_compress;
static zstd_bound_t z_bound;
static zstd_iserror_t z_iserror;
static zstd_framesize_t z_framesize;
static int z_state;                        /* 0 not tried, 1 bound, -1 missing */

static int zstd_bind(void) {
    if (z_state) return z_state;
    void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_GLOBAL);
    if (h) {
        z_decompress = (zstd_decompress_t)dlsym(h, "ZSTD_decompress");
        z_compress = (zstd_compress_t)dlsym(h, "ZSTD_compress");
        z_bound = (zstd_bound_t)dlsym(h, "ZSTD_compressBound");
        z_iserror = (zstd_iserror_t)dlsym(h, "ZSTD_isError");
        z_framesize = (zstd_framesize_t)dlsym(h, "ZSTD_getFrameContentSize");
        z_create_dctx = (zstd_create_t)dlsym(h, "ZSTD_createDCtx"); z_free_dctx = (zstd_free_t)dlsym(h, "ZSTD_freeDCtx");
        z_create_cctx = (zstd_create_t)dlsym(h, "ZSTD_createCCtx"); z_free_cctx = (zstd_free_t)dlsym(h, "ZSTD_freeCCtx");
        z_decompress_ctx = (zstd_decompress_ctx_t)dlsym(h, "ZSTD_decompressDCtx");
        z_compress_ctx = (zstd_compress_ctx_t)dlsym(h, "ZSTD_compressCCtx");
    }
    z_state = (h && z_decompress && z_compress && z_bound && z_iserror && z_framesize && z_create_dctx && z_free_dctx &&
               z_create_cctx && z_free_cctx && z_decompress_ctx && z_compress_ctx) ? 1 : -1;
    return z_state;
}

/* Copy the part of one decoded chunk (C order, chunk[0] x chunk[1] x chunk[2] elements of `elem` bytes; NULL: the
 * fill value) that falls inside the destination box.  origin = destination coordinates of the chunk's first element. */
static void place_chunk(const uint8_t* block, const void* fill, int elem, const int32_t chunk[3], uint8_t* dst,
                        const int64_t strides[3], const int32_t shape[3], const int32_t origin[3]) {
    int32_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = origin[a] < 0 ? -origin[a] : 0;
        hi[a] = origin[a] + chunk[a] > shape[a] ? shape[a] - origin[a] : chunk[a];
        if (hi[a] <= lo[a]) return;
    }
    const size_t run = (size_t)(hi[2] - lo[2]) * (size_t)elem;
    for (int32_t i = lo[0]; i < hi[0]; ++i)
        for (int32_t j = lo[1]; j < hi[1]; ++j) {
            uint8_t* d = dst + (int64_t)(origin[0] + i) * strides[0] + (int64_t)(origin[1] + j) * strides[1] +
                         (int64_t)(origin[2] + lo[2]) * strides[2];
            if (block) {
                const uint8_t* s = block + (((size_t)i * chunk[1] + j) * chunk[2] + lo[2]) * (size_t)elem;
                if (strides[2] == elem) memcpy(d, s, run);
                else for (int32_t k = 0; k < hi[2] - lo[2]; ++k) memcpy(d + (int64_t)k * strides[2], s + (size_t)k * elem, elem);
            } else {
                for (int32_t k = 0; k < hi[2] - lo[2]; ++k) memcpy(d + (int64_t)k * strides[2], fill, elem);
            }
        }
}

/* Decode n chunks into a strided destination box.
 *   base + off[k], nbytes[k]: the stored bytes of chunk k (off[k] == UINT64_MAX: not stored -> fill value;
 *                             base == NULL: off[k] is the address itself)
 *   zstd / crc: the chunk's bytes->bytes codecs in ENCODE order zstd, then crc32c (either may be absent)
 *   origin[3 k ..]: destination coordinates of chunk k's first element (may lie outside the box: clipped)
 * Returns 0, -1 when libzstd is needed and missing, or 1 + k for the first chunk that fails (checksum mismatch,
 * corrupt frame, wrong decoded size). */
typedef struct {
    const uint8_t* base; const uint64_t* off; const uint64_t* nbytes; int zstd, crc, elem; const int32_t* chunk;
    uint8_t* dst; const int64_t* strides; const int32_t* shape; const int32_t* origin; const void* fill;
    size_t raw; uint8_t* tmp[POOL_MAX]; void* ctx[POOL_MAX]; atomic_int bad;
} decode_job;

static void decode_one(int k, int worker, void* arg) {
    decode_job* j = (decode_job*)arg;
    if (j->off[k] == UINT64_MAX) { place_chunk(NULL, j->fill, j->elem, j->chunk, j->dst, j->strides, j->shape, j->origin + 3 * k); return; }
    const uint8_t* p = j->base ? j->base + j->off[k] : (const uint8_t*)(uintptr_t)j->off[k];     /* base NULL: off[] holds addresses */
    size_t len = (size_t)j->nbytes[k];
    int ok = 1;
    if (j->crc) {
        uint32_t want;
        ok = len >= 4;
        if (ok) { memcpy(&want, p + len - 4, 4); len -= 4; ok = svr_crc32c(p, len, 0) == want; }
    }
    const uint8_t* block = p;
    if (ok && j->zstd) {
        if (!j->tmp[worker]) { j->tmp[worker] = (uint8_t*)malloc(j->raw ? j->raw : 1); j->ctx[worker] = z_create_dctx(); }
        uint8_t* tmp = j->tmp[worker];
        const unsigned long long claimed = z_framesize(p, len);
        ok = tmp && j->ctx[worker] && (claimed == j->raw || claimed >= 0xFFFFFFFFFFFFFFFEull);     /* unknown size: decode and see */
        if (ok) { const size_t got = z_decompress_ctx(j->ctx[worker], tmp, j->raw, p, len); ok = !z_iserror(got) && got == j->raw; }
        block = tmp;
    } else if (ok) {
        ok = len == j->raw;
    }
    if (ok) { place_chunk(block, j->fill, j->elem, j->chunk, j->dst, j->strides, j->shape, j->origin + 3 * k); return; }
    int cur = atomic_load(&j->bad);                      /* keep the FIRST failing chunk */
    while ((cur == 0 || k + 1 < cur) && !atomic_compare_exchange_weak(&j->bad, &cur, k + 1)) {}
}

int svr_zarr_decode_chunks(int n, const uint8_t* base, const uint64_t* off, const uint64_t* nbytes, int zstd, int crc,
                           int elem, const int32_t chunk[3], uint8_t* dst, const int64_t dst_strides[3],
                           const int32_t dst_shape[3], const int32_t* origin, const void* fill, int nthreads) {
    if (zstd && zstd_bind() != 1) return -1;
    decode_job j = { base, off, nbytes, zstd, crc, elem, chunk, dst, dst_strides, dst_shape, origin, fill,
                     (size_t)chunk[0] * chunk[1] * chunk[2] * (size_t)elem, { 0 }, { 0 }, 0 };
    pool_run(n, nthreads < 1 ? 1 : nthreads, decode_one, &j);
    for (int w = 0; w < POOL_MAX; ++w) { free(j.tmp[w]); if (j.ctx[w]) z_free_dctx(j.ctx[w]); }
    return atomic_load(&j.bad);
}

/* Encode n chunks cut from a C-contiguous source block of shape src_shape (elements of `elem` bytes): chunk k starts at
 * corner[3 k ..] and is padded with `fill` beyond the block.  out: n slots of `slot` bytes each (slot >= bound, see
 * svr_zarr_encode_bound); out_bytes[k] = encoded size, or 0 when skip_fill is set and the chunk holds only the fill value.
 * Returns 0, -1 without libzstd, -2 when a slot is too small. */
size_t svr_zarr_encode_bound(size_t raw, int zstd) {
    if (zstd && zstd_bind() != 1) return 0;
    return (zstd ? z_bound(raw) : raw) + 4;
}

typedef struct {
    const uint8_t* src; const int32_t* src_shape; const int32_t* corner; int elem; const int32_t* chunk; const void* fill;
    int zstd, level, crc, skip_fill; uint8_t* out; size_t slot; uint64_t* out_bytes; size_t raw; uint8_t* blk[POOL_MAX]; void* ctx[POOL_MAX]; atomic_int rc;
} encode_job;

static void encode_one(int k, int worker, void* arg) {
    encode_job* j = (encode_job*)arg;
    if (!j->blk[worker]) { j->blk[worker] = (uint8_t*)malloc(j->raw ? j->raw : 1); if (j->zstd) j->ctx[worker] = z_create_cctx(); }
    uint8_t* blk = j->blk[worker];
    const int32_t* c = j->corner + 3 * k;
    const int32_t* chunk = j->chunk; const int32_t* src_shape = j->src_shape;
    const int elem = j->elem;
    if (!blk || (j->zstd && !j->ctx[worker])) { atomic_store(&j->rc, -2); j->out_bytes[k] = 0; return; }
    for (int32_t i = 0; i < chunk[0]; ++i)
        for (int32_t jj = 0; jj < chunk[1]; ++jj) {
            uint8_t* d = blk + ((size_t)i * chunk[1] + jj) * chunk[2] * (size_t)elem;
            const int inside = c[0] + i < src_shape[0] && c[1] + jj < src_shape[1];
            const int32_t have = inside ? (c[2] + chunk[2] <= src_shape[2] ? chunk[2] : (src_shape[2] > c[2] ? src_shape[2] - c[2] : 0)) : 0;
            if (have > 0)
                memcpy(d, j->src + (((size_t)(c[0] + i) * src_shape[1] + (c[1] + jj)) * src_shape[2] + c[2]) * (size_t)elem,
                       (size_t)have * elem);
            for (int32_t q = have; q < chunk[2]; ++q) memcpy(d + (size_t)q * elem, j->fill, elem);
        }
    if (j->skip_fill) {
        int only_fill = 1;
        for (size_t q = 0; q < j->raw && only_fill; q += elem) only_fill = memcmp(blk + q, j->fill, elem) == 0;
        if (only_fill) { j->out_bytes[k] = 0; return; }
    }
    uint8_t* o = j->out + (size_t)k * j->slot;
    size_t len = j->raw;
    if (j->zstd) {
        len = z_compress_ctx(j->ctx[worker], o, j->slot - 4, blk, j->raw, j->level);
        if (z_iserror(len)) { atomic_store(&j->rc, -2); j->out_bytes[k] = 0; return; }
    } else {
        memcpy(o, blk, j->raw);
    }
    if (j->crc) { const uint32_t sum = svr_crc32c(o, len, 0); memcpy(o + len, &sum, 4); len += 4; }
    j->out_bytes[k] = len;
}

int svr_zarr_encode_chunks(int n, const uint8_t* src, const int32_t src_shape[3], const int32_t* corner, int elem,
                           const int32_t chunk[3], const void* fill, int zstd, int level, int crc, int skip_fill,
                           uint8_t* out, size_t slot, uint64_t* out_bytes, int nthreads) {
    if (zstd && zstd_bind() != 1) return -1;
    const size_t raw = (size_t)chunk[0] * chunk[1] * chunk[2] * (size_t)elem;
    if (slot < (zstd ? z_bound(raw) : raw) + 4) return -2;
    encode_job j = { src, src_shape, corner, elem, chunk, fill, zstd, level, crc, skip_fill, out, slot, out_bytes, raw, { 0 }, { 0 }, 0 };
    pool_run(n, nthreads < 1 ? 1 : nthreads, encode_one, &j);
    for (int w = 0; w < POOL_MAX; ++w) { free(j.blk[w]); if (j.ctx[w]) z_free_cctx(j.ctx[w]); }
    return atomic_load(&j.rc);
}

/* ---------------------------------------------------------------------------------------------------------------
 * A whole read request in one call: the inner chunks are handed over in GROUPS, each group = the chunks of the request that
 * live in one file (a shard, or a plain chunk file when chunks_per_shard == 0), at most a few dozen per group so that
 * the groups spread over the pool.  Per group: open the file, read and verify the shard index (little-endian u64
 * (offset, nbytes) pairs [+ crc32c]), read the span that holds the group's chunks with one pread, then check / decompress /
 * place every chunk.  A missing file or an index entry of 2^64 - 1 is the fill value.
 * Returns 0, -1 without libzstd, -2 for an I/O or index error (bad_out = the group), or 1 + k for the first corrupt
 * chunk k (position in the request's chunk list).
 * --------------------------------------------------------------------------------------------------------------- */
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

typedef struct {
    const char* const* paths; const int32_t* grp_first; const int32_t* within; int chunks_per_shard, index_at_end, index_crc;
    decode_job dec;                       /* base = NULL: off[] / nbytes[] are filled here per chunk */
    uint64_t* off; uint64_t* nbytes;
    uint8_t* span[POOL_MAX]; size_t span_cap[POOL_MAX];
    atomic_int io_error; atomic_ullong stored;
} read_job;

static int pread_all(int fd, void* buf, size_t n, off_t at) {
    uint8_t* p = (uint8_t*)buf;
    while (n) {
        const ssize_t got = pread(fd, p, n, at);
        if (got < 0 && errno == EINTR) continue;
        if (got <= 0) return -1;
        p += got; n -= (size_t)got; at += got;
    }
    return 0;
}

static void read_group(int g, int worker, void* arg) {
    read_job* j = (read_job*)arg;
    const int k0 = j->grp_first[g], k1 = j->grp_first[g + 1];
    if (k1 <= k0) return;
    const int fd = open(j->paths[g], O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        if (errno != ENOENT) { atomic_store(&j->io_error, g + 1); return; }
        for (int k = k0; k < k1; ++k) { j->off[k] = UINT64_MAX; decode_one(k, worker, &j->dec); }      /* not stored */
        return;
    }
    struct stat st;
    int ok = fstat(fd, &st) == 0;
    uint64_t lo = UINT64_MAX, hi = 0;
    if (ok && j->chunks_per_shard == 0) {                                     /* a plain chunk file: the file is the chunk */
        j->off[k0] = 0; j->nbytes[k0] = (uint64_t)st.st_size; lo = 0; hi = (uint64_t)st.st_size;
        ok = k1 == k0 + 1;
    } else if (ok) {
        const size_t ilen = (size_t)16 * j->chunks_per_shard + (j->index_crc ? 4 : 0);
        uint64_t stack_index[2 * 512];
        uint64_t* index = ilen <= sizeof(stack_index) ? stack_index : (uint64_t*)malloc(ilen);
        ok = index && (size_t)st.st_size >= ilen && pread_all(fd, index, ilen, j->index_at_end ? st.st_size - (off_t)ilen : 0) == 0;
        if (ok && j->index_crc) {
            uint32_t want; memcpy(&want, (uint8_t*)index + ilen - 4, 4);
            ok = svr_crc32c(index, ilen - 4, 0) == want;
        }
        for (int k = k0; ok && k < k1; ++k) {
            const uint64_t o = index[2 * j->within[k]], nb = index[2 * j->within[k] + 1];
            j->off[k] = o; j->nbytes[k] = nb;
            if (o == UINT64_MAX && nb == UINT64_MAX) continue;
            if (o > (uint64_t)st.st_size || nb > (uint64_t)st.st_size - o) { ok = 0; break; }
            if (o < lo) lo = o;
            if (o + nb > hi) hi = o + nb;
        }
        if (index != stack_index) free(index);
    }
    if (ok && hi > lo) {
        const size_t need = (size_t)(hi - lo);
        if (j->span_cap[worker] < need) {
            free(j->span[worker]);
            j->span_cap[worker] = need + need / 2 + 4096;
            j->span[worker] = (uint8_t*)malloc(j->span_cap[worker]);
            if (!j->span[worker]) { j->span_cap[worker] = 0; ok = 0; }
        }
        ok = ok && pread_all(fd, j->span[worker], need, (off_t)lo) == 0;
        if (ok) atomic_fetch_add(&j->stored, (unsigned long long)need);
    }
    close(fd);
    if (!ok) { atomic_store(&j->io_error, g + 1); return; }
    for (int k = k0; k < k1; ++k) {
        if (j->off[k] != UINT64_MAX) j->off[k] = (uint64_t)(uintptr_t)(j->span[worker] + (j->off[k] - lo));     /* address (base = NULL) */
        decode_one(k, worker, &j->dec);
    }
}

int svr_zarr_read_groups(int ngroups, const char* const* paths, const int32_t* grp_first, const int32_t* within,
                         int chunks_per_shard, int index_at_end, int index_crc, int zstd, int crc, int elem,
                         const int32_t chunk[3], uint8_t* dst, const int64_t dst_strides[3], const int32_t dst_shape[3],
                         const int32_t* origin, const void* fill, int nthreads, uint64_t* stored_bytes, int* bad_group) {
    if (zstd && zstd_bind() != 1) return -1;
    const int n = grp_first[ngroups];
    read_job j;
    memset(&j, 0, sizeof(j));
    j.paths = paths; j.grp_first = grp_first; j.within = within;
    j.chunks_per_shard = chunks_per_shard; j.index_at_end = index_at_end; j.index_crc = index_crc;
    j.off = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n > 0 ? n : 1));
    j.nbytes = (uint64_t*)calloc((size_t)(n > 0 ? n : 1), sizeof(uint64_t));
    if (!j.off || !j.nbytes) { free(j.off); free(j.nbytes); return -2; }
    decode_job d = { NULL, j.off, j.nbytes, zstd, crc, elem, chunk, dst, dst_strides, dst_shape, origin, fill,
                     (size_t)chunk[0] * chunk[1] * chunk[2] * (size_t)elem, { 0 }, { 0 }, 0 };
    j.dec = d;
    pool_run(ngroups, nthreads < 1 ? 1 : nthreads, read_group, &j);
    for (int w = 0; w < POOL_MAX; ++w) { free(j.dec.tmp[w]); if (j.dec.ctx[w]) z_free_dctx(j.dec.ctx[w]); free(j.span[w]); }
    free(j.off); free(j.nbytes);
    if (stored_bytes) *stored_bytes = (uint64_t)atomic_load(&j.stored);
    const int io = atomic_load(&j.io_error);
    if (io) { if (bad_group) *bad_group = io - 1; return -2; }
    return atomic_load(&j.dec.bad);
}
