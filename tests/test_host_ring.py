"""Host logic of the product's ring buffer (no GPU): the reference's known answers,
its hypothesis property, and agreement of the upload plan with the oracle."""
from itertools import combinations

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import ring_oracle as R
from sub_volume_renderer_amd import Coordinate, Roi, WrappingBuffer, subtract_rois
from sub_volume_renderer_amd._wrapping_buffer import set_dim

from helpers import KNOWN, as_pair, fixture_arrays


def make(fixture):
    data, seg, ring, chunk = fixture_arrays(fixture)
    return WrappingBuffer(data, seg, Coordinate(ring), Coordinate(chunk))


def roi_of(r):
    return Roi(tuple(r[0]), tuple(r[1]))


@pytest.mark.parametrize("case", KNOWN["wrap"], ids=lambda c: c["name"])
def test_wrap_known_answers(case):
    buf = make(case["fixture"])
    got = buf.wrap_logical_roi_into_buffer_rois(roi_of(case["roi"]))
    assert len(got) == len(case["pairs"])
    for b, l in case["pairs"]:
        assert (roi_of(b), roi_of(l)) in got


def test_wrap_asserts_like_reference():
    buf = make("F1")
    with pytest.raises(AssertionError):
        buf.wrap_logical_roi_into_buffer_rois(Roi((0, 0, 0), (6, 1, 1)))     # larger than the ring
    with pytest.raises(AssertionError):
        buf.wrap_logical_roi_into_buffer_rois(Roi((0, 0), (1, 1)))           # wrong rank
    assert buf.wrap_logical_roi_into_buffer_rois(Roi((3, 3, 3), (0, 2, 2))) == []


@pytest.mark.parametrize("case", KNOWN["can_load"], ids=lambda c: str(c["roi"]))
def test_can_load_known_answers(case):
    assert make(case["fixture"]).can_load_logical_roi(roi_of(case["roi"])) is case["expect"]


@pytest.mark.parametrize("case", KNOWN["subtract"], ids=lambda c: c["name"])
def test_subtract_known_answers(case):
    a, b = roi_of(case["a"]), roi_of(case["b"])
    got = subtract_rois(a, b)
    if "exact" in case:
        assert got == [roi_of(r) for r in case["exact"]]
    if "set" in case:
        assert set(got) == {roi_of(r) for r in case["set"]}
    assert len(got) <= case.get("max_slabs", 6)
    for r in got:
        assert not r.intersects(b)
        assert a.contains(r)
    assert sum(x.size for x in got) == a.size - a.intersect(b).size
    assert all(not x.intersects(y) for x, y in combinations(got, 2))
    # and the decomposition itself equals the oracle's (same slabs, same order)
    assert [(tuple(r.offset), tuple(r.shape)) for r in got] == R.subtract_rois(as_pair(case["a"]), as_pair(case["b"]))


@st.composite
def two_random_rois(draw):
    # the reference's strategy (tests/wrapping_buffer/test_subtract_rois.py:105-121): 1..10 dims, unbounded ints
    length = draw(st.integers(min_value=1, max_value=10))
    lists = st.lists(st.integers(min_value=0), min_size=length, max_size=length)
    return (Roi(tuple(draw(lists)), tuple(draw(lists))), Roi(tuple(draw(lists)), tuple(draw(lists))), length)


@given(two_random_rois())
@settings(deadline=None, max_examples=200)
def test_property_subtract(data):
    a, b, length = data
    result = subtract_rois(a, b)
    assert len(result) <= 2 * length
    for r in result:
        assert not r.intersects(b)
        assert a.contains(r)
    assert sum(x.size for x in result) == a.size - a.intersect(b).size
    assert all(not x.intersects(y) for x, y in combinations(result, 2))


def test_set_dim():
    assert set_dim(Coordinate(1, 2, 3), 1, 9) == (1, 9, 3)
    assert set_dim(Coordinate((7,)), 0, 4) == (4,)


def test_roi_algebra_matches_funlib_semantics():
    a = Roi((0, 0, 0), (4, 4, 4))
    assert a.intersect(Roi((8, 8, 8), (1, 1, 1))).empty
    assert not a.intersects(Roi((4, 0, 0), (2, 2, 2)))              # touching is not intersecting
    assert not a.intersects(Roi((1, 1, 1), (0, 2, 2)))              # empty intersects nothing
    assert Roi((3, 5, -3), (6, 2, 4)).snap_to_grid(Coordinate(4, 4, 4), mode="grow") == Roi((0, 4, -4), (12, 4, 8))
    assert (Roi((8, 4, 12), (4, 8, 4)) / Coordinate(4, 4, 4)) == Roi((2, 1, 3), (1, 2, 1))
    assert (Roi((2, 1, 3), (1, 2, 1)) * Coordinate(4, 4, 4)) == Roi((8, 4, 12), (4, 8, 4))
    assert Roi((1, 1, 1), (2, 2, 2)) + Coordinate(1, 0, 2) == Roi((2, 1, 3), (2, 2, 2))
    assert a.contains(Roi((1, 1, 1), (3, 3, 3))) and not a.contains(Roi((1, 1, 1), (4, 1, 1)))
    assert a.size == 64 and a.dims == 3 and a.end == (4, 4, 4) and a.begin == (0, 0, 0)


def test_snapped_roi_clips_then_grows():
    buf = make("F2")
    assert buf.get_snapped_roi_in_pixels(Roi((12, 12, 12), (8, 8, 8))) == Roi((12, 12, 12), (4, 4, 4))
    assert buf.get_snapped_roi_in_pixels(Roi((-4, -4, -4), (8, 8, 8))) == Roi((0, 0, 0), (4, 4, 4))
    assert buf.get_snapped_roi_in_pixels(Roi((20, 20, 20), (4, 4, 4))).empty
    assert buf.get_snapped_roi_in_pixels(Roi((1, 1, 1), (3, 5, 7))) == Roi((0, 0, 0), (4, 8, 8))
    # data extent not a chunk multiple: the grown ROI may pass the data end
    data = np.zeros((10, 10, 10), np.uint8)
    b2 = WrappingBuffer(data, data, (3, 3, 3), (4, 4, 4))
    assert b2.get_snapped_roi_in_pixels(Roi((5, 5, 5), (5, 5, 5))) == Roi((4, 4, 4), (8, 8, 8))


def apply_plan(buf, pieces, ring_d, ring_l):
    """Replay an upload plan on numpy arrays the way load_into_buffer would (clip to the data)."""
    c = buf.chunk_shape_in_pixels
    for b, l in pieces:
        dst, src = b * c, l * c
        src = Roi((0, 0, 0), buf.backing_data.shape).intersect(src)
        if src.empty:
            continue
        dst = Roi(dst.offset, src.shape)
        ring_d[dst.to_slices()] = np.array(buf.backing_data[src.to_slices()], np.float32)
        ring_l[dst.to_slices()] = np.array(buf.segmentations[src.to_slices()], np.uint32)


@pytest.mark.parametrize("seed", range(6))
def test_plan_sequence_matches_oracle(seed):
    """Random walks of load_logical_roi: same ROI state, same pieces, same ring contents as the oracle."""
    rng = np.random.default_rng(seed)
    shape = tuple(int(v) for v in rng.integers(20, 45, 3))
    chunk = tuple(int(v) for v in rng.integers(2, 7, 3))
    ring = tuple(int(v) for v in rng.integers(2, 6, 3))
    data = rng.integers(0, 255, shape, dtype=np.uint8)
    seg = rng.integers(0, 2 ** 32 - 1, shape, dtype=np.uint32)
    prod = WrappingBuffer(data, seg, ring, chunk)
    orac = R.OracleWrappingBuffer(data, seg, ring, chunk)
    ring_d = np.zeros(prod.shape_in_pixels, np.float32)
    ring_l = np.zeros(prod.shape_in_pixels, np.uint32)
    pos = np.array([s // 2 for s in shape])
    for _ in range(25):
        pos = pos + rng.integers(-7, 8, 3)
        # up to (N-1)*C per axis: the domain in which growing to the chunk grid always fits
        # (_wobject.py:151-177); occasionally far too large (silent no-op)
        cap = (np.array(ring) - 1) * np.array(chunk)
        size = tuple(int(v) for v in rng.integers(0, cap + 1))
        if rng.random() < 0.1:
            size = tuple(int(v) for v in np.array(prod.shape_in_pixels) + 1)
        off = tuple(int(p - s // 2) for p, s in zip(pos, size))
        plan = prod.plan_logical_roi(Roi(off, size))
        n_before = len(orac.uploads)
        orac.load_logical_roi((off, size))
        if plan is None:
            assert len(orac.uploads) == n_before
        else:
            snapped, in_chunks, pieces = plan
            prod._current_logical_roi_in_pixels = snapped
            prod._current_logical_roi_in_chunks = in_chunks
            apply_plan(prod, pieces, ring_d, ring_l)
        want = orac.current_logical_roi_in_pixels
        got = prod._current_logical_roi_in_pixels
        assert (got is None) == (want is None)
        if got is not None:
            assert (tuple(got.offset), tuple(got.shape)) == want
        np.testing.assert_array_equal(ring_d, orac.texture)
        np.testing.assert_array_equal(ring_l, orac.segmentations_texture)
        u, w = prod.uniform_buffer.data, orac.uniform()
        assert tuple(u["current_logical_offset_in_pixels"]) == w["offset"]
        assert tuple(u["current_logical_shape_in_pixels"]) == w["shape"]


def test_a_window_that_stays_on_its_chunk_grid_plans_nothing():
    """Requests that snap to the resident window (a camera moving inside one chunk) return the resident state and no
    pieces — without any ROI subtraction — and never disagree with the full computation."""
    rng = np.random.default_rng(11)
    data = rng.integers(0, 255, (40, 40, 40), dtype=np.uint8)
    seg = np.zeros((40, 40, 40), np.uint32)
    prod = WrappingBuffer(data, seg, (4, 4, 4), (4, 4, 4))
    snapped, in_chunks, pieces = prod.plan_logical_roi(Roi((9, 9, 9), (10, 10, 10)))
    assert pieces and (tuple(snapped.offset), tuple(snapped.shape)) == ((8, 8, 8), (12, 12, 12))
    prod._current_logical_roi_in_pixels, prod._current_logical_roi_in_chunks = snapped, in_chunks
    for off in ((8, 8, 8), (9, 9, 10), (10, 10, 10)):                  # all snap to [8, 20)^3
        again = prod.plan_logical_roi(Roi(off, (10, 10, 10)))
        assert again == (snapped, in_chunks, [])
        assert subtract_rois(again[1], in_chunks) == []
    moved = prod.plan_logical_roi(Roi((11, 10, 10), (10, 10, 10)))   # crosses into the next chunk along axis 0
    assert moved[2] and moved[0] != snapped


def test_unaligned_ring_sized_roi_asserts_like_reference():
    """can_load only checks the unsnapped shape (_wrapping_buffer.py:145-158), so a ring-sized
    ROI that is not chunk-aligned grows past the ring and trips the wrap assertion (:215-221)
    — in the reference, in the oracle and here alike."""
    data = np.zeros((40, 40, 40), np.uint8)
    prod = WrappingBuffer(data, data, (2, 2, 2), (4, 4, 4))
    orac = R.OracleWrappingBuffer(data, data, (2, 2, 2), (4, 4, 4))
    with pytest.raises(AssertionError):
        orac.load_logical_roi(((5, 5, 5), (8, 8, 8)))
    with pytest.raises(AssertionError):
        prod.plan_logical_roi(Roi((5, 5, 5), (8, 8, 8)))
