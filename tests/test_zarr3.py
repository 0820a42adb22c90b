"""The zarr v3 reader (sub_volume_renderer_amd/zarr3.py): the on-disk format of the stores the reference's pyramid
builders write (scripts/create_mouse_multiscale.py:102-131).  No zarr library and no zarr-written fixture exist
offline, so the reader is pinned against byte strings assembled by hand from the zarr v3 specification's layout,
and against the module's own writer."""
import json
import os
import struct

import numpy as np
import pytest

from sub_volume_renderer_amd import SubVolume, SubVolumeMaterial, zarr3


def _write(path, rel, data):
    full = os.path.join(path, *rel.split("/"))
    os.makedirs(os.path.dirname(full), exist_ok=True)
    with open(full, "wb") as f:
        f.write(data)


def test_crc32c_check_value():
    assert zarr3.crc32c(b"123456789") == 0xE3069283          # the CRC-32C (Castagnoli) check value
    assert zarr3.crc32c(b"") == 0


def test_crc32c_is_verified_at_every_size_and_agrees_with_the_python_loop():
    """Round 2 verified checksums only up to 64 KiB; a corrupt large chunk must be refused like a small one."""
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 9, 4096, 65536, 65537, 300001):
        body = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        crc = zarr3.crc32c(body)
        if n <= 65537:
            assert crc == zarr3._crc32c_python(body)
        framed = body + struct.pack("<I", crc)
        codecs = [{"name": "crc32c"}]
        assert zarr3._decode_bytes_codecs(codecs, framed, None) == body
        bad = bytearray(framed)
        bad[n // 2] ^= 0x10
        with pytest.raises(ValueError, match="crc32c"):
            zarr3._decode_bytes_codecs(codecs, bytes(bad), None)
    with pytest.raises(ValueError, match="crc32c"):
        zarr3._decode_bytes_codecs([{"name": "crc32c"}], b"\x01\x02", None)        # shorter than a checksum


def test_zstd_frame_larger_than_its_chunk_is_refused():
    big = zarr3._zstd_compress(bytes(1 << 20))
    with pytest.raises(ValueError, match="zstd frame claims"):
        zarr3._zstd_decompress(big, expected=4096)
    assert zarr3._zstd_decompress(big, expected=1 << 20) == bytes(1 << 20)


def test_hand_assembled_plain_chunks_big_endian_and_missing_chunk(tmp_path):
    root = str(tmp_path / "a.zarr")
    meta = {"zarr_format": 3, "node_type": "array", "shape": [4, 5], "data_type": "uint16",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [2, 3]}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}},
            "fill_value": 7, "codecs": [{"name": "bytes", "configuration": {"endian": "big"}}], "attributes": {"k": 1}}
    _write(root, "zarr.json", json.dumps(meta).encode())
    want = np.full((4, 5), 7, np.uint16)
    # chunk (0,0): rows 0-1, cols 0-2; chunk (1,1): rows 2-3, cols 3-5 (col 5 is padding); chunks (0,1), (1,0) missing
    c00 = np.arange(6, dtype=np.uint16).reshape(2, 3) + 100
    c11 = np.arange(6, dtype=np.uint16).reshape(2, 3) + 200
    _write(root, "c/0/0", c00.astype(">u2").tobytes())     # stored big-endian, as the bytes codec says
    _write(root, "c/1/1", c11.astype(">u2").tobytes())
    want[0:2, 0:3] = c00
    want[2:4, 3:5] = c11[:, :2]
    a = zarr3.open_zarr(root)
    assert (a.shape, a.dtype, a.chunks, a.shards, a.ndim) == ((4, 5), np.dtype("uint16"), (2, 3), None, 2)
    assert a.attrs == {"k": 1}
    np.testing.assert_array_equal(a[:, :], want)
    np.testing.assert_array_equal(a[1:3, 2:5], want[1:3, 2:5])
    np.testing.assert_array_equal(a[-1], want[-1])
    np.testing.assert_array_equal(a[..., 4], want[:, 4])
    np.testing.assert_array_equal(np.asarray(a), want)
    assert a[3, 4] == want[3, 4]


def test_hand_assembled_shard_with_index_at_end_and_v2_keys(tmp_path):
    """One 4x4 shard of 2x2 inner chunks, two of them stored: body = chunk(0,0) | chunk(1,1); index = C-order
    (offset, nbytes) pairs, 2^64-1 for the missing ones, little-endian u64, + crc32c."""
    root = str(tmp_path / "s.zarr")
    meta = {"zarr_format": 3, "node_type": "array", "shape": [4, 4], "data_type": "uint8",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [4, 4]}},
            "chunk_key_encoding": {"name": "v2", "configuration": {"separator": "."}}, "fill_value": 0,
            "codecs": [{"name": "sharding_indexed", "configuration": {
                "chunk_shape": [2, 2], "codecs": [{"name": "bytes"}],
                "index_codecs": [{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "crc32c"}],
                "index_location": "end"}}]}
    _write(root, "zarr.json", json.dumps(meta).encode())
    E = 2 ** 64 - 1
    body = bytes([1, 2, 3, 4]) + bytes([9, 8, 7, 6])
    index = struct.pack("<8Q", 0, 4, E, E, E, E, 4, 4)
    _write(root, "0.0", body + index + struct.pack("<I", zarr3.crc32c(index)))
    a = zarr3.ZarrV3Array(root)
    assert a.chunks == (2, 2) and a.shards == (4, 4)
    np.testing.assert_array_equal(a[:, :], np.array([[1, 2, 0, 0], [3, 4, 0, 0], [0, 0, 9, 8], [0, 0, 7, 6]], np.uint8))
    # a corrupted index is refused
    _write(root, "0.0", body + index + struct.pack("<I", 12345))
    with pytest.raises(ValueError):
        zarr3.ZarrV3Array(root)[:, :]


@pytest.mark.parametrize("compressor", [None, "gzip", "zstd"])
@pytest.mark.parametrize("shards,index_location", [(None, "end"), ((32, 32, 32), "end"), ((32, 16, 32), "start")])
def test_round_trip_of_the_reference_builders_layout(tmp_path, compressor, shards, index_location):
    rng = np.random.default_rng(3)
    data = rng.integers(0, 255, (40, 33, 70), dtype=np.uint8)
    data[8:24, :, :] = 0                                   # whole chunks of fill value: not stored
    root = str(tmp_path / "r.zarr")
    a = zarr3.write_array(root, data, chunks=(8, 8, 16), shards=shards, compressor=compressor, index_location=index_location)
    assert a.chunks == (8, 8, 16) and a.shape == data.shape and a.dtype == np.uint8
    np.testing.assert_array_equal(a[:, :, :], data)
    for _ in range(20):
        lo = [int(rng.integers(0, n)) for n in data.shape]
        hi = [int(rng.integers(l, n + 1)) for l, n in zip(lo, data.shape)]
        sl = tuple(slice(l, h) for l, h in zip(lo, hi))
        np.testing.assert_array_equal(a[sl], data[sl])
    stored = sum(len(files) for _, _, files in os.walk(root)) - 1
    full = int(np.prod([-(-n // c) for n, c in zip(data.shape, shards or (8, 8, 16))]))
    assert 0 < stored < full or shards is not None         # all-fill chunks have no file (plain layout)


def test_labels_uint32_float32_and_group_of_scales(tmp_path):
    rng = np.random.default_rng(4)
    g = zarr3.create_group(str(tmp_path / "labels.zarr"), {"multiscales": "scale0..2"})
    arrays = {}
    for k in range(3):
        n = 32 >> k
        arrays[k] = rng.integers(0, 2 ** 32, (n, n, n), dtype=np.uint32)
        zarr3.write_array(os.path.join(g.path, f"scale{k}"), arrays[k], chunks=(16 >> k,) * 3 if k < 2 else (8, 8, 8),
                          shards=(16, 16, 16) if k == 0 else None)
    grp = zarr3.open_group(str(tmp_path / "labels.zarr"))
    assert grp.keys() == ["scale0", "scale1", "scale2"] and "scale1" in grp and "scale9" not in grp
    for k in range(3):
        np.testing.assert_array_equal(grp[f"scale{k}"][:, :, :], arrays[k])
    f = rng.random((10, 12), dtype=np.float32)
    fa = zarr3.write_array(str(tmp_path / "f.zarr"), f, chunks=(4, 5), compressor="zstd", fill_value=float("nan"))
    np.testing.assert_array_equal(fa[:, :], f)


def test_unsupported_codecs_fail_loudly(tmp_path):
    root = str(tmp_path / "b.zarr")
    meta = {"zarr_format": 3, "node_type": "array", "shape": [4], "data_type": "uint8",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [4]}},
            "chunk_key_encoding": {"name": "default"}, "fill_value": 0,
            "codecs": [{"name": "bytes"}, {"name": "blosc", "configuration": {"cname": "zstd"}}]}
    _write(root, "zarr.json", json.dumps(meta).encode())
    _write(root, "c/0", b"\x00" * 20)
    with pytest.raises(NotImplementedError, match="blosc"):
        zarr3.ZarrV3Array(root)[:]
    with pytest.raises(FileNotFoundError):
        zarr3.ZarrV3Array(str(tmp_path))                   # not a store at all
    _write(str(tmp_path / "g"), "zarr.json", json.dumps({"zarr_format": 3, "node_type": "group"}).encode())
    with pytest.raises(ValueError):
        zarr3.ZarrV3Array(str(tmp_path / "g"))             # a group is not an array


def test_subvolume_takes_chunk_shapes_from_zarr_arrays(tmp_path):
    """``chunk_shape_in_pixels=None``: the chunking comes from the base array's ``.chunks`` (_wobject.py:46-53),
    which for a sharded array is the inner chunk shape."""
    data = np.zeros((32, 32, 32), np.uint8)
    seg = np.zeros((32, 32, 32), np.uint32)
    d = zarr3.write_array(str(tmp_path / "raw.zarr"), data + 1, chunks=(8, 8, 16), shards=(16, 16, 32))
    s = zarr3.write_array(str(tmp_path / "seg.zarr"), seg + 1, chunks=(8, 8, 16), shards=(16, 16, 32))
    vol = SubVolume(SubVolumeMaterial(0.5), [(d, s)], (3, 3, 2))
    assert tuple(vol.wrapping_buffers[0].chunk_shape_in_pixels) == (8, 8, 16)
    assert tuple(vol.wrapping_buffers[0].shape_in_pixels) == (24, 24, 32)


# ---------------------------------------------------------------------------------------------------
# the native reader / writer (csrc/host_codecs.c) against the Python path
# ---------------------------------------------------------------------------------------------------
def _native_built():
    """(evaluated at collection time, before any fixture: build the library here if gcc is around)"""
    try:
        import __graft_entry__ as g

        g.build_host_codecs()
    except Exception:  # noqa: BLE001 - no compiler: the native tests skip, the Python path is still tested
        pass
    return zarr3._host_codecs() is not None


@pytest.mark.skipif(not _native_built(), reason="libsvr_hostcodec.so not built")
@pytest.mark.parametrize("dtype", [np.uint8, np.uint32, np.float32])
@pytest.mark.parametrize("compressor,shards", [("zstd", (64, 64, 64)), ("zstd", None), (None, (32, 64, 16)), ("gzip", (64, 64, 64))])
def test_native_reader_equals_python_reader(tmp_path, dtype, compressor, shards):
    """Ragged array (extents that are no chunk multiples), fill-value chunks left out, boxes that cut chunks and shards:
    `svr_zarr_decode_chunks` and the per-chunk Python path must hand out the same arrays (gzip: no native path, the
    request silently takes the Python one)."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 200, (150, 70, 97)).astype(dtype)
    a[40:100] = 9                                           # whole chunks of the fill value: not stored
    z = zarr3.write_array(str(tmp_path / "a"), a, (16, 16, 16), shards, compressor=compressor, fill_value=9)
    assert (z._native_codecs is not None) == (compressor != "gzip")
    for box in [((0, 0, 0), (150, 70, 97)), ((3, 17, 5), (77, 64, 96)), ((149, 69, 96), (150, 70, 97)), ((64, 0, 64), (128, 64, 97)),
                ((45, 3, 3), (99, 60, 60))]:
        sl = tuple(slice(l, h) for l, h in zip(*box))
        z.native = True
        fast = z[sl]
        z.native = False
        z._cache.clear()
        slow = z[sl]
        np.testing.assert_array_equal(fast, a[sl])
        np.testing.assert_array_equal(slow, a[sl])
        assert fast.dtype == slow.dtype == a.dtype and fast.flags.c_contiguous
    assert z.read_bytes > 0 and z.read_seconds > 0


@pytest.mark.skipif(not _native_built(), reason="libsvr_hostcodec.so not built")
def test_native_reader_refuses_corrupt_chunks(tmp_path):
    """A flipped bit inside one inner chunk of a shard: zstd notices (corrupt frame), and with a crc32c codec behind it
    the checksum does — the request raises and names the chunk; other boxes still read."""
    rng = np.random.default_rng(6)
    a = rng.integers(0, 255, (64, 64, 64)).astype(np.uint8)
    for tail in (["zstd"], ["zstd", "crc32c"], ["crc32c"]):
        root = str(tmp_path / ("c_" + "_".join(tail)))
        z = zarr3.write_array(root, a, (16, 16, 16), (64, 64, 64), compressor="zstd" if "zstd" in tail else None)
        if "crc32c" in tail:                                  # re-write with a checksummed inner chain
            meta = json.load(open(os.path.join(root, "zarr.json")))
            inner = meta["codecs"][0]["configuration"]["codecs"]
            inner.append({"name": "crc32c"})
            json.dump(meta, open(os.path.join(root, "zarr.json"), "w"))
            z = zarr3.ZarrV3Array(root)
            zarr3.write_block(z, (0, 0, 0), a)
            z = zarr3.ZarrV3Array(root)
        np.testing.assert_array_equal(z[:, :, :], a)
        file = os.path.join(root, "c", "0", "0", "0")
        raw = bytearray(open(file, "rb").read())
        index = np.frombuffer(bytes(raw[-(16 * 64 + 4):-4]), "<u8").reshape(4, 4, 4, 2)
        off, n = (int(v) for v in index[1, 2, 3])
        # (zstd keeps incompressible data as a raw block without a checksum: a flipped payload bit goes unnoticed unless a
        # crc32c codec follows; a damaged frame header does not)
        raw[off + (1 if tail == ["zstd"] else n // 2)] ^= 0x40
        open(file, "wb").write(bytes(raw))
        z = zarr3.ZarrV3Array(root)
        np.testing.assert_array_equal(z[0:16, 0:16, 0:16], a[0:16, 0:16, 0:16])          # an intact chunk
        with pytest.raises(ValueError, match=r"chunk \(1, 2, 3\).*corrupt"):
            z[:, :, :]
        if tail != ["zstd"]:                                  # (a flipped bit in raw zstd payload is not always detected by zstd alone)
            z.native = False
            with pytest.raises(ValueError, match="crc32c"):
                z[16:32, 32:48, 48:64]


def test_create_array_and_write_block_make_a_sparse_store(tmp_path):
    """Only the shards that are written exist on disk; everything else reads as the fill value (how bench.py lays down
    the corridor of config 4's fly-through inside a 4096^3 array)."""
    z = zarr3.create_array(str(tmp_path / "big"), (4096, 4096, 4096), np.uint8, (16, 16, 16), (64, 64, 64), fill_value=0)
    rng = np.random.default_rng(7)
    block = rng.integers(1, 255, (64, 64, 64)).astype(np.uint8)
    n = zarr3.write_block(z, (31, 2, 63), block)
    assert n > 0 and zarr3.write_block(z, (5, 5, 5), np.zeros((64, 64, 64), np.uint8)) == 0      # nothing but fill: no file
    files = [os.path.join(r, f) for r, _, fs in os.walk(str(tmp_path / "big")) for f in fs]
    assert len(files) == 2                                     # zarr.json + one shard
    got = z[31 * 64 - 3:32 * 64 + 3, 2 * 64:3 * 64, 4090:4096]
    want = np.zeros((70, 64, 6), np.uint8)
    want[3:67] = block[:, :, 58:64]
    np.testing.assert_array_equal(got, want)


@pytest.mark.skipif(not _native_built(), reason="libsvr_hostcodec.so not built")
def test_native_reader_survives_a_fork(tmp_path, monkeypatch):
    """The decoder's thread pool does not exist in a forked child (threads are not inherited): the child must build its
    own instead of waiting for workers that are not there."""
    monkeypatch.setenv("SVR_ZARR_THREADS", "4")
    a = np.random.default_rng(8).integers(0, 255, (96, 80, 64)).astype(np.uint8)
    z = zarr3.write_array(str(tmp_path / "a"), a, (16, 16, 16), (32, 32, 32))
    np.testing.assert_array_equal(z[:, :, :], a)               # the parent's pool exists now
    pid = os.fork()
    if pid == 0:
        try:
            ok = np.array_equal(z[:, :, :], a)
        except BaseException:
            ok = False
        os._exit(0 if ok else 1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    np.testing.assert_array_equal(z[3:, :, 5:], a[3:, :, 5:])


@pytest.mark.skipif(not _native_built(), reason="libsvr_hostcodec.so not built")
def test_native_reader_reports_damaged_shards(tmp_path):
    """A truncated shard (shorter than its index says) and a shard whose index fails its crc32c: the request raises and
    names the file; a missing shard is not an error (fill value)."""
    a = np.random.default_rng(9).integers(1, 255, (128, 64, 64)).astype(np.uint8)
    root = str(tmp_path / "a")
    z = zarr3.write_array(root, a, (16, 16, 16), (64, 64, 64), fill_value=0)
    np.testing.assert_array_equal(z[:, :, :], a)
    first, second = os.path.join(root, "c", "0", "0", "0"), os.path.join(root, "c", "1", "0", "0")
    raw = open(first, "rb").read()
    open(first, "wb").write(raw[:len(raw) // 2] + raw[-(16 * 64 + 4):])          # body cut short, index intact
    with pytest.raises(ValueError, match="c/0/0/0"):
        zarr3.ZarrV3Array(root)[0:64, :, :]
    np.testing.assert_array_equal(zarr3.ZarrV3Array(root)[64:128, :, :], a[64:128])     # the other shard still reads
    raw2 = bytearray(open(second, "rb").read())
    raw2[-10] ^= 1                                                                 # inside the index
    open(second, "wb").write(bytes(raw2))
    with pytest.raises(ValueError, match="c/1/0/0"):
        zarr3.ZarrV3Array(root)[64:128, :, :]
    os.remove(second)
    assert not zarr3.ZarrV3Array(root)[64:128, :, :].any()


# ---------------------------------------------------------------------------------------------------
# property test (the reference's own wrapping-buffer tests use hypothesis the same way, FUTURE.md:210-216)
# ---------------------------------------------------------------------------------------------------
from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402


@pytest.mark.skipif(not _native_built(), reason="libsvr_hostcodec.so not built")
@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(data=st.data())
def test_any_box_of_any_store_reads_back_what_was_written(tmp_path_factory, data):
    """Random array shapes, chunk / shard shapes, dtypes, codecs, index locations, fill values and request boxes: the
    native reader, the Python reader and numpy slicing of the source agree."""
    shape = tuple(data.draw(st.integers(1, 40)) for _ in range(3))
    chunks = tuple(data.draw(st.sampled_from([1, 2, 3, 4, 8])) for _ in range(3))
    sharded = data.draw(st.booleans())
    shards = tuple(c * data.draw(st.integers(1, 3)) for c in chunks) if sharded else None
    dtype = data.draw(st.sampled_from([np.uint8, np.uint16, np.uint32, np.float32]))
    compressor = data.draw(st.sampled_from(["zstd", None]))
    fill = data.draw(st.sampled_from([0, 3]))
    loc = data.draw(st.sampled_from(["end", "start"]))
    seed = data.draw(st.integers(0, 2 ** 16))
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 6, shape).astype(dtype)                  # few distinct values: some chunks are all fill
    a[rng.random(shape) < 0.5] = fill
    root = str(tmp_path_factory.mktemp("z") / "a")
    z = zarr3.write_array(root, a, chunks, shards, compressor=compressor, fill_value=fill, index_location=loc)
    lo = [data.draw(st.integers(0, n - 1)) for n in shape]
    hi = [data.draw(st.integers(l + 1, n)) for l, n in zip(lo, shape)]
    sl = tuple(slice(l, h) for l, h in zip(lo, hi))
    z.native = True
    fast = z[sl]
    z.native = False
    z._cache.clear()
    slow = z[sl]
    np.testing.assert_array_equal(fast, a[sl])
    np.testing.assert_array_equal(slow, a[sl])
