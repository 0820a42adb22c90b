/*
 * synth_host.c — host generator of the deterministic synthetic volumes (sub_volume_renderer_amd/synth.py),
 * bit for bit the same closed form as its numpy / torch versions, fused and multi-threaded.
 *
 * Config C4 streams a 4096^3 volume that is never resident: every ring reload asks the lazy backing array
 * for a block, and a block of LOD k costs 8^k LOD-0 evaluations.  numpy evaluates the closed form one
 * whole-array operation at a time (tens of passes over int64 temporaries); this file does one pass.
 * It stands in for the chunk reader of a real store (zarr / tensorstore), it is not part of the render path.
 *
 * Build: gcc -O3 -fopenmp -fPIC -shared (see __graft_entry__.build_synth).
 */
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Exact division / remainder of 32-bit operands by a run-time constant through one 64 x 64 -> 128 bit multiply
 * (Lemire, Kaser, Kurz: "Faster remainder by direct computation", 2019): M = floor((2^64 - 1) / d) + 1. */
typedef struct { uint64_t M; uint32_t d; } fastdiv_t;
static fastdiv_t fastdiv_make(uint32_t d) { fastdiv_t f; f.M = UINT64_MAX / d + 1; f.d = d; return f; }
static inline uint32_t fast_div(uint32_t a, fastdiv_t f) { return (uint32_t)(((__uint128_t)f.M * a) >> 64); }
static inline uint32_t fast_mod(uint32_t a, fastdiv_t f) { return (uint32_t)(((__uint128_t)(f.M * a) * f.d) >> 64); }

typedef struct {
    int64_t n;            /* LOD-0 edge */
    fastdiv_t p[3];       /* triangle-wave periods (synth._periods) */
    int64_t nb;           /* 32^3 blocks per edge */
    fastdiv_t n_labels;
} synth_params;

static inline uint32_t tri(uint32_t u, fastdiv_t p) {      /* synth._tri: min(u % p, p - u % p) * 160 // p */
    const uint32_t m = fast_mod(u, p);
    const uint32_t t = m < p.d - m ? m : p.d - m;
    return fast_div(t * 160u, p);
}

static inline uint32_t hash3(int64_t a0, int64_t a1, int64_t a2) {     /* synth._hash3 (all arithmetic mod 2^32 after the xor) */
    uint64_t h = ((uint64_t)(a0 * 73856093)) ^ ((uint64_t)(a1 * 19349663)) ^ ((uint64_t)(a2 * 83492791));
    h &= 0xFFFFFFFFull;
    h ^= h >> 13;
    h = (h * 0x5BD1E995ull) & 0xFFFFFFFFull;
    h ^= h >> 15;
    return (uint32_t)h;
}

/* one LOD-0 voxel: synth.lod0_block */
static inline void lod0_voxel(const synth_params* q, int64_t a0, int64_t a1, int64_t a2, uint32_t* dens, uint32_t* lab) {
    /* coordinates are < 2^24 (checked by the caller), so every operand below fits 32 bits */
    const uint32_t t0 = tri((uint32_t)(3 * a0 + a1 + 2 * a2), q->p[0]);
    const uint32_t t1 = tri((uint32_t)(a0 + 4 * a1 + 2 * a2), q->p[1]);
    const uint32_t t2 = tri((uint32_t)(2 * a0 + a1 + 5 * a2), q->p[2]);
    const uint32_t d = (t0 * t1 * t2 * 3u) / 6400u + (hash3(a0, a1, a2) & 15u);       /* t <= 80: product < 2^21 */
    *dens = d;
    *lab = 0u;
    if (d >= 32u) {
        const int64_t bid = ((a0 >> 5) * q->nb + (a1 >> 5)) * q->nb + (a2 >> 5);
        *lab = fast_mod((uint32_t)(((uint64_t)bid * 2654435761ull) & 0xFFFFFFFFull), q->n_labels);
    }
}

/* voxel (a0, a1, a2) of LOD `lod`: 2x mean-pool (floor, level by level) of the density, 2x max-pool of the labels */
static void lod_voxel(const synth_params* q, int lod, int64_t a0, int64_t a1, int64_t a2, uint32_t* dens, uint32_t* lab) {
    if (lod == 0) { lod0_voxel(q, a0, a1, a2, dens, lab); return; }
    uint32_t sum = 0, mx = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t d, l;
        lod_voxel(q, lod - 1, 2 * a0 + (i >> 2), 2 * a1 + ((i >> 1) & 1), 2 * a2 + (i & 1), &d, &l);
        sum += d;
        if (l > mx) mx = l;
    }
    *dens = sum / 8;
    *lab = mx;
}

/* Block [off, off + shape) of LOD `lod` of an n^3 volume into packed arrays (a2 fastest).
 * Either output may be NULL.  Returns 0. */
int svr_synth_block(int64_t n, int lod, const int64_t off[3], const int64_t shape[3], int64_t n_labels,
                    const int64_t periods[3], uint8_t* density, uint32_t* labels, int nthreads) {
    if (n < 1 || n >= (1 << 24) || lod < 0 || lod > 8 || n_labels < 1 || n_labels > 0xFFFFFFFFll) return -1;
    for (int a = 0; a < 3; ++a)
        if (periods[a] < 1 || periods[a] > 0x7FFFFFFF || off[a] < 0 || shape[a] < 0 || ((off[a] + shape[a]) << lod) > (1 << 24)) return -1;
    synth_params q;
    q.n = n;
    for (int a = 0; a < 3; ++a) q.p[a] = fastdiv_make((uint32_t)periods[a]);
    q.nb = (n + 31) / 32; q.n_labels = fastdiv_make((uint32_t)n_labels);
    const int64_t rows = shape[0] * shape[1];
#ifdef _OPENMP
    const int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nt) if (rows * shape[2] * (1ll << (3 * lod)) > 200000)
#else
    (void)nthreads;
#endif
    for (int64_t r = 0; r < rows; ++r) {
        const int64_t a0 = off[0] + r / shape[1], a1 = off[1] + r % shape[1];
        uint8_t* drow = density ? density + r * shape[2] : NULL;
        uint32_t* lrow = labels ? labels + r * shape[2] : NULL;
        for (int64_t x = 0; x < shape[2]; ++x) {
            uint32_t d, l;
            lod_voxel(&q, lod, a0, a1, off[2] + x, &d, &l);
            if (drow) drow[x] = (uint8_t)d;
            if (lrow) lrow[x] = l;
        }
    }
    return 0;
}
