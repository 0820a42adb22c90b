"""The raw chunk-directory array (stand-in for the zarr / tensorstore sources the reference accepts,
README.md:18): round trip, slicing across chunk borders, and use as backing data of a ring buffer."""
import numpy as np
import pytest

from oracle import ring_oracle as R
from sub_volume_renderer_amd import Roi, SubVolume, SubVolumeMaterial, WrappingBuffer
from sub_volume_renderer_amd.chunkstore import ChunkDirArray, write_chunk_dir


@pytest.fixture
def store(tmp_path):
    rng = np.random.default_rng(1)
    data = rng.integers(0, 255, (37, 22, 50), dtype=np.uint8)      # not chunk multiples: padded edge chunks
    return data, write_chunk_dir(data, str(tmp_path / "vol"), (16, 8, 16))


def test_roundtrip_and_cross_chunk_slices(store):
    data, arr = store
    assert arr.shape == data.shape and arr.dtype == data.dtype and arr.chunks == (16, 8, 16) and arr.ndim == 3
    np.testing.assert_array_equal(arr[:, :, :], data)
    rng = np.random.default_rng(2)
    for _ in range(50):
        lo = [int(rng.integers(0, s)) for s in data.shape]
        hi = [int(rng.integers(l, s + 1)) for l, s in zip(lo, data.shape)]
        sl = tuple(slice(l, h) for l, h in zip(lo, hi))
        np.testing.assert_array_equal(arr[sl], data[sl])
    assert arr[5:5, :, :].shape == (0, 22, 50)
    with pytest.raises(IndexError):
        arr[::2, :, :]
    reopened = ChunkDirArray(arr.root)
    np.testing.assert_array_equal(reopened[3:30, 2:20, 10:40], data[3:30, 2:20, 10:40])


def test_chunks_attribute_is_picked_up_like_zarr(store, tmp_path):
    """_wobject.py:46-53: chunk shape inferred from the first array's ``.chunks``."""
    data, arr = store
    seg = write_chunk_dir(np.zeros(data.shape, np.uint32), str(tmp_path / "seg"), (16, 8, 16))
    vol = SubVolume(SubVolumeMaterial(0.5), [(arr, seg)], (2, 2, 2))
    assert tuple(vol.wrapping_buffers[0].chunk_shape_in_pixels) == (16, 8, 16)
    buf = WrappingBuffer(arr, seg, (2, 3, 2))
    assert tuple(buf.shape_in_pixels) == (32, 24, 32)


def test_ring_plan_from_a_chunk_store_matches_oracle(store, tmp_path):
    data, arr = store
    seg_np = (np.arange(data.size, dtype=np.uint32).reshape(data.shape) * 7919) % 1000
    seg = write_chunk_dir(seg_np, str(tmp_path / "seg"), (16, 8, 16))
    prod = WrappingBuffer(arr, seg, (2, 3, 2))
    orac = R.OracleWrappingBuffer(arr, seg, (2, 3, 2), (16, 8, 16))
    for off in [(0, 0, 0), (9, 4, 20), (21, 6, 18), (5, -3, 30)]:
        roi = Roi(off, (16, 16, 16))
        plan = prod.plan_logical_roi(roi)
        orac.load_logical_roi((off, (16, 16, 16)))
        snapped, in_chunks, pieces = plan
        prod._current_logical_roi_in_pixels, prod._current_logical_roi_in_chunks = snapped, in_chunks
        assert (tuple(snapped.offset), tuple(snapped.shape)) == orac.current_logical_roi_in_pixels
    # the oracle consumed the chunk store through __getitem__: its texture equals the numpy data
    roi = orac.current_logical_roi_in_pixels
    want = R.brute_force_ring(data, orac.shape_in_pixels, roi)
    sel = R.brute_force_ring(np.ones_like(data), orac.shape_in_pixels, roi) > 0
    np.testing.assert_array_equal(orac.texture[sel], want[sel])


@pytest.mark.gpu
def test_upload_from_a_chunk_store(store, tmp_path):
    data, arr = store
    seg_np = (np.arange(data.size, dtype=np.uint32).reshape(data.shape) * 7919) % 1000
    seg = write_chunk_dir(seg_np, str(tmp_path / "seg"), (16, 8, 16))
    buf = WrappingBuffer(arr, seg, (2, 3, 2))
    orac = R.OracleWrappingBuffer(data, seg_np, (2, 3, 2), (16, 8, 16))
    for off in [(0, 0, 0), (9, 4, 20), (21, 6, 18)]:
        buf.load_logical_roi(Roi(off, (16, 16, 16)))
        orac.load_logical_roi((off, (16, 16, 16)))
    np.testing.assert_array_equal(buf.texture.data, orac.texture)
    np.testing.assert_array_equal(buf.segmentations_texture.data, orac.segmentations_texture)
