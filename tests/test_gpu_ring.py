"""GPU: ring-buffer contents in HBM after loads through the C ABI, against the reference's
known answers and the oracle's textures (bit-exact)."""
import numpy as np
import pytest

from oracle import ring_oracle as R
from sub_volume_renderer_amd import Coordinate, Roi, WrappingBuffer

from helpers import KNOWN, as_pair, fixture_arrays, slices
from test_oracle_ring import check_boundary_case

pytestmark = pytest.mark.gpu


def make(fixture):
    data, seg, ring, chunk = fixture_arrays(fixture)
    return WrappingBuffer(data, seg, Coordinate(ring), Coordinate(chunk)), R.OracleWrappingBuffer(data, seg, ring, chunk)


def roi_of(r):
    return Roi(tuple(r[0]), tuple(r[1]))


@pytest.mark.parametrize("case", KNOWN["load"], ids=lambda c: c["name"])
def test_load_known_answers(case):
    buf, orac = make(case["fixture"])
    before = None
    for r in case["loads"]:
        before = buf.texture.data
        buf.load_logical_roi(roi_of(r))
        orac.load_logical_roi(as_pair(r))
    tex = buf.texture.data
    assert tex.dtype == np.float32 and buf.segmentations_texture.data.dtype == np.uint32
    for eq in case.get("equal", []):
        np.testing.assert_array_equal(tex[slices(eq["ring"])], buf.backing_data[slices(eq["data"])])
    for z in case.get("zero", []):
        assert np.all(tex[slices(z)] == 0)
    if case.get("all_zero"):
        assert np.all(tex == 0)
    if case.get("idempotent"):
        np.testing.assert_array_equal(before, tex)
    np.testing.assert_array_equal(tex, orac.texture)
    np.testing.assert_array_equal(buf.segmentations_texture.data, orac.segmentations_texture)


@pytest.mark.parametrize("case", KNOWN["load_into_buffer"], ids=lambda c: c["name"])
def test_load_into_buffer_known_answers(case):
    buf, _ = make(case["fixture"])
    buf.load_into_buffer(roi_of(case["buffer_roi"]), roi_of(case["logical_roi"]))
    eq = case["equal"]
    np.testing.assert_array_equal(buf.texture.data[slices(eq["ring"])], buf.backing_data[slices(eq["data"])])


@pytest.mark.parametrize("case", KNOWN["boundary"], ids=lambda c: c["name"])
def test_boundary_known_answers(case):
    buf, orac = make(case["fixture"])
    for r in case["loads"]:
        buf.load_logical_roi(roi_of(r))
        orac.load_logical_roi(as_pair(r))
        if case.get("not_none_after_each_load"):
            assert buf._current_logical_roi_in_pixels is not None
    got = buf._current_logical_roi_in_pixels
    got = None if got is None else (tuple(got.offset), tuple(got.shape))
    assert got == orac.current_logical_roi_in_pixels
    tex = buf.texture.data
    check_boundary_case(case, got, tex, buf.backing_data, tuple(buf.shape_in_pixels))
    np.testing.assert_array_equal(tex, orac.texture)


@pytest.mark.parametrize("seed", range(4))
def test_random_walk_matches_oracle(seed):
    rng = np.random.default_rng(100 + seed)
    shape = tuple(int(v) for v in rng.integers(24, 60, 3))
    chunk = tuple(int(v) for v in rng.integers(2, 9, 3))
    ring = tuple(int(v) for v in rng.integers(2, 6, 3))
    data = rng.integers(0, 65535, shape, dtype=np.uint16)
    seg = rng.integers(0, 2 ** 32 - 1, shape, dtype=np.uint32)
    buf = WrappingBuffer(data, seg, ring, chunk)
    orac = R.OracleWrappingBuffer(data, seg, ring, chunk)
    pos = np.array([s // 2 for s in shape])
    cap = (np.array(ring) - 1) * np.array(chunk)
    for _ in range(20):
        pos = pos + rng.integers(-9, 10, 3)
        size = tuple(int(v) for v in rng.integers(0, cap + 1))
        off = tuple(int(p - s // 2) for p, s in zip(pos, size))
        buf.load_logical_roi(Roi(off, size))
        orac.load_logical_roi((off, size))
    np.testing.assert_array_equal(buf.texture.data, orac.texture)
    np.testing.assert_array_equal(buf.segmentations_texture.data, orac.segmentations_texture)
    st = orac.uniform()
    u = buf.uniform_buffer.data
    assert tuple(u["current_logical_offset_in_pixels"]) == st["offset"]


@pytest.mark.parametrize("ddt,ldt", [(np.uint8, np.uint8), (np.uint16, np.uint64), (np.float32, np.int32),
                                     (np.float64, np.uint32), (np.int16, np.int64), (np.uint32, np.uint16)])
def test_upload_dtype_conversion_is_numpy_cast(ddt, ldt):
    rng = np.random.default_rng(7)
    shape = (12, 10, 20)
    if np.issubdtype(ddt, np.floating):
        data = (rng.random(shape) * 300 - 20).astype(ddt)
    else:
        data = rng.integers(0, min(np.iinfo(ddt).max, 2 ** 31), shape).astype(ddt)
    seg = rng.integers(0, min(np.iinfo(ldt).max, 2 ** 40), shape).astype(ldt)
    buf = WrappingBuffer(data, seg, (3, 5, 2), (4, 2, 10))
    buf.load_logical_roi(Roi((0, 0, 0), (8, 8, 10)))
    np.testing.assert_array_equal(buf.texture.data[:8, :8, :10], np.array(data[:8, :8, :10], dtype=np.float32))
    np.testing.assert_array_equal(buf.segmentations_texture.data[:8, :8, :10],
                                  np.array(seg[:8, :8, :10], dtype=np.uint32))


def test_non_contiguous_and_device_sources():
    import torch

    rng = np.random.default_rng(3)
    big = rng.integers(0, 255, (20, 24, 64), dtype=np.uint8)
    data = big[:, ::2, ::2]                    # strided view, x stride 2 bytes
    seg = rng.integers(0, 2 ** 31, data.shape, dtype=np.int64).astype(np.uint32)
    buf = WrappingBuffer(data, seg, (3, 3, 2), (4, 4, 16))
    buf.load_logical_roi(Roi((4, 0, 0), (8, 8, 32)))
    np.testing.assert_array_equal(buf.texture.data[4:12, :8, :32], data[4:12, :8, :32].astype(np.float32))
    # device-resident backing volume (torch tensors on the GPU)
    d_dev = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    s_dev = torch.from_numpy(seg.view(np.int32)).cuda()
    buf2 = WrappingBuffer(d_dev, s_dev, (3, 3, 2), (4, 4, 16))
    buf2.load_logical_roi(Roi((4, 0, 0), (8, 8, 32)))
    np.testing.assert_array_equal(buf2.texture.data, buf.texture.data)
    np.testing.assert_array_equal(buf2.segmentations_texture.data, buf.segmentations_texture.data)


def test_large_region_is_split_over_staging_slots():
    rng = np.random.default_rng(5)
    data = rng.integers(0, 255, (300, 256, 256), dtype=np.uint8)       # 19.7 M voxels, u8 + u32 = 98 MB > one slot
    seg = (np.arange(data.size, dtype=np.uint32).reshape(data.shape) * 2654435761).astype(np.uint32)
    buf = WrappingBuffer(data, seg, (5, 4, 4), (64, 64, 64))
    buf.load_logical_roi(Roi((0, 0, 0), (256, 256, 256)))
    np.testing.assert_array_equal(buf.texture.data[:256], data[:256].astype(np.float32))
    np.testing.assert_array_equal(buf.segmentations_texture.data[:256], seg[:256])


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32], ids=lambda d: np.dtype(d).name)
def test_micro_block_copy_follows_the_ring_through_wrapped_loads(dtype):
    """svr_lod_desc::blocked_twin: after every load — chunk-aligned ones through the 16-voxel streaming scatter, odd ones
    through the per-voxel scatter, windows that wrap around the ring — the micro-block copy holds exactly the ring's
    elements at svr_blocked_index's places; and a cleared ring clears its copy."""
    from sub_volume_renderer_amd.testing import micro_blocks_of, read_micro_block_copy

    rng = np.random.default_rng(7)
    shape = (96, 80, 160)
    data = (rng.integers(1, 255, shape).astype(dtype) if dtype != np.float32 else rng.standard_normal(shape).astype(np.float32))
    seg = rng.integers(0, 2 ** 31, shape, dtype=np.uint32)
    for chunk, ring in (((4, 4, 16), (6, 5, 4)), ((2, 2, 4), (10, 6, 6))):          # rings 24x20x64 and 20x12x24 voxels (z, y, x)
        buf = WrappingBuffer(data, seg, Coordinate(ring), Coordinate(chunk))
        assert buf.rings.blocked_twin == [True]
        ring_px = tuple(int(v) for v in buf.shape_in_pixels)
        for off in ((0, 0, 0), (8, 4, 16), (40, 36, 80), (70, 60, 130), (33, 27, 51), (0, 0, 0)):
            # (one chunk short of the ring: an offset off the chunk grid snaps outwards by up to a chunk)
            want = tuple(min(r - c, s - o) for r, c, s, o in zip(ring_px, chunk, shape, off))
            buf.load_logical_roi(Roi(off, want))
            tex = buf.texture.data                                       # float32 view of the ring (values exact)
            twin = read_micro_block_copy(buf.rings, 0)
            np.testing.assert_array_equal(twin, micro_blocks_of(tex.astype(twin.dtype)))
            assert np.count_nonzero(twin) > 0
    import ctypes as C

    from sub_volume_renderer_amd import _native as N
    N.check(N.lib().svr_clear_lod(buf.rings.handle, 0), "svr_clear_lod")
    assert np.count_nonzero(read_micro_block_copy(buf.rings, 0)) == 0 and np.count_nonzero(buf.texture.data) == 0
