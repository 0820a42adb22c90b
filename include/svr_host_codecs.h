/*
 * svr_host_codecs.h — C ABI of libsvr_hostcodec.so (sub_volume_renderer_amd/csrc/host_codecs.c): the native half of the
 * chunk-store reader that feeds the ring buffers.
 *
 * In the reference the backing arrays of a WrappingBuffer are zarr.Array / tensorstore objects whose reads run in those
 * libraries' own C++ (README.md:18; src/sub_volume/_wrapping_buffer.py:307-322 `backing_data[...]`, `.read().result()`),
 * over the stores written by scripts/create_mouse_multiscale.py:98-131 (zarr v3, 16^3 chunks in 64^3 shards).  Neither
 * library exists on the target; sub_volume_renderer_amd/zarr3.py reads that format and hands whole read requests to
 * these entry points.  Host code only (gcc; libzstd bound at run time); plain pointers and sizes; thread-safe
 * (requests are serialised on an internal pool of sleeping worker threads).
 *
 * 3-D arrays, inner-chunk codec chain `bytes` (little endian) [-> zstd] [-> crc32c]; anything else stays on zarr3.py's
 * Python path.  Return values: 0 = ok; -1 = the zstd codec is needed and libzstd is not installed; other codes below.
 */
#ifndef SVR_HOST_CODECS_H
#define SVR_HOST_CODECS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32C (Castagnoli) of n bytes, continuing from `crc` (0 for a fresh checksum): what the zarr v3 `crc32c` codec
 * appends to a chunk or a shard index (little endian). */
uint32_t svr_crc32c(const void* data, size_t n, uint32_t crc);

/* Decode n inner chunks whose stored bytes are already in memory into the strided destination box `dst`
 * (dst_shape elements of `elem` bytes, dst_strides in bytes).  Chunk k: base + off[k] (base NULL: off[k] is the address
 * itself), nbytes[k] bytes; off[k] == UINT64_MAX: not stored -> `fill`.  origin[3k..]: destination coordinates of the
 * chunk's first element (clipped to the box).  zstd / crc: the chunk codecs present, in encode order zstd then crc32c.
 * Returns 1 + k for the first chunk that fails its checksum, is a corrupt frame or has the wrong decoded size. */
int svr_zarr_decode_chunks(int n, const uint8_t* base, const uint64_t* off, const uint64_t* nbytes, int zstd, int crc,
                           int elem, const int32_t chunk[3], uint8_t* dst, const int64_t dst_strides[3],
                           const int32_t dst_shape[3], const int32_t* origin, const void* fill, int nthreads);

/* A whole read request: the chunks come in `ngroups` groups (grp_first[g] .. grp_first[g + 1] in the chunk list), each
 * group = chunks of ONE file `paths[g]` — a shard whose index holds `chunks_per_shard` little-endian (offset, nbytes)
 * u64 pairs [+ crc32c when index_crc] at its end (index_at_end) or start, `within[k]` = the chunk's position in that
 * index; or a plain chunk file (chunks_per_shard == 0).  The pool's threads open the files, read and verify the
 * indexes, read the byte ranges, then check / decompress / place every chunk as svr_zarr_decode_chunks does.  A missing
 * file or an empty index entry is the fill value.  stored_bytes: bytes read from the files.
 * Returns -2 for an unreadable / too short file or an index that fails its checksum (bad_group = the group), or
 * 1 + k for the first corrupt chunk. */
int svr_zarr_read_groups(int ngroups, const char* const* paths, const int32_t* grp_first, const int32_t* within,
                         int chunks_per_shard, int index_at_end, int index_crc, int zstd, int crc, int elem,
                         const int32_t chunk[3], uint8_t* dst, const int64_t dst_strides[3], const int32_t dst_shape[3],
                         const int32_t* origin, const void* fill, int nthreads, uint64_t* stored_bytes, int* bad_group);

/* Encode n chunks cut from a C-contiguous source block (src_shape elements; chunk k starts at corner[3k..] and is padded
 * with `fill` beyond the block) into n slots of `slot` bytes (>= svr_zarr_encode_bound(raw chunk bytes, zstd));
 * out_bytes[k] = encoded size, 0 for a chunk of nothing but fill values when skip_fill is set.  -2: a slot is too small. */
size_t svr_zarr_encode_bound(size_t raw, int zstd);
int svr_zarr_encode_chunks(int n, const uint8_t* src, const int32_t src_shape[3], const int32_t* corner, int elem,
                           const int32_t chunk[3], const void* fill, int zstd, int level, int crc, int skip_fill,
                           uint8_t* out, size_t slot, uint64_t* out_bytes, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
