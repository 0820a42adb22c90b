"""Pin the oracle's ring-buffer restatement against every known answer of the
reference's own tests (tests/wrapping_buffer/*, transcribed in golden/ring_known_answers.json)."""
from itertools import combinations

import numpy as np
import pytest

from oracle import ring_oracle as R

from helpers import KNOWN, as_pair, fixture_arrays, slices


def make(fixture):
    data, seg, ring, chunk = fixture_arrays(fixture)
    return R.OracleWrappingBuffer(data, seg, ring, chunk)


@pytest.mark.parametrize("case", KNOWN["wrap"], ids=lambda c: c["name"])
def test_wrap_known_answers(case):
    buf = make(case["fixture"])
    got = buf.wrap_logical_roi_into_buffer_rois(as_pair(case["roi"]))
    want = [(as_pair(b), as_pair(l)) for b, l in case["pairs"]]
    assert len(got) == len(want)
    for pair in want:
        assert pair in got


@pytest.mark.parametrize("case", KNOWN["can_load"], ids=lambda c: str(c["roi"]))
def test_can_load_known_answers(case):
    assert make(case["fixture"]).can_load_logical_roi(as_pair(case["roi"])) is case["expect"]


def check_subtract_invariants(a, b, result, max_slabs):
    assert len(result) <= max_slabs
    for r in result:
        assert not R.roi_intersects(r, b)
        assert R.roi_contains(a, r)
    assert sum(R.roi_size(x) for x in result) == R.roi_size(a) - R.roi_size(R.roi_intersect(a, b))
    assert all(not R.roi_intersects(x, y) for x, y in combinations(result, 2))


@pytest.mark.parametrize("case", KNOWN["subtract"], ids=lambda c: c["name"])
def test_subtract_known_answers(case):
    a, b = as_pair(case["a"]), as_pair(case["b"])
    got = R.subtract_rois(a, b)
    if "exact" in case:
        assert got == [as_pair(r) for r in case["exact"]]
    if "set" in case:
        assert sorted(got) == sorted(as_pair(r) for r in case["set"])
    check_subtract_invariants(a, b, got, case.get("max_slabs", 2 * len(a[0])))


@pytest.mark.parametrize("case", KNOWN["load"], ids=lambda c: c["name"])
def test_load_known_answers(case):
    buf = make(case["fixture"])
    before = None
    for r in case["loads"]:
        before = buf.texture.copy()
        buf.load_logical_roi(as_pair(r))
    for eq in case.get("equal", []):
        np.testing.assert_array_equal(buf.texture[slices(eq["ring"])], buf.backing_data[slices(eq["data"])])
    for z in case.get("zero", []):
        assert np.all(buf.texture[slices(z)] == 0)
    if case.get("all_zero"):
        assert np.all(buf.texture == 0)
    if case.get("idempotent"):
        np.testing.assert_array_equal(before, buf.texture)


@pytest.mark.parametrize("case", KNOWN["load_into_buffer"], ids=lambda c: c["name"])
def test_load_into_buffer_known_answers(case):
    buf = make(case["fixture"])
    buf.load_into_buffer(as_pair(case["buffer_roi"]), as_pair(case["logical_roi"]))
    eq = case["equal"]
    np.testing.assert_array_equal(buf.texture[slices(eq["ring"])], buf.backing_data[slices(eq["data"])])


def check_boundary_case(case, roi_px, ring_texture, data, ring_shape):
    if case["roi_is_none"] is None:                 # the reference test leaves it open whether a ROI is published;
        if roi_px is None:                          # the remaining bounds apply if one is
            return
    elif case["roi_is_none"]:
        assert roi_px is None
        return
    assert roi_px is not None
    off, shp = roi_px
    if "offset_ge" in case:
        assert all(o >= g for o, g in zip(off, case["offset_ge"]))
    if "end_le" in case:
        assert all(o + s <= e for o, s, e in zip(off, shp, case["end_le"]))
    if "shape_ge" in case:
        assert all(s >= g for s, g in zip(shp, case["shape_ge"]))
    if "aligned" in case:
        assert all(o % case["aligned"] == 0 and s % case["aligned"] == 0 for o, s in zip(off, shp))
    if case.get("ring_matches_data"):
        # test_boundary_loading.py:133-160: buf[pos % ring] == data[pos] for every voxel of the ROI
        want = R.brute_force_ring(data, ring_shape, roi_px)
        sel = R.brute_force_ring(np.ones_like(data), ring_shape, roi_px) > 0
        np.testing.assert_array_equal(ring_texture[sel], want[sel])


@pytest.mark.parametrize("case", KNOWN["boundary"], ids=lambda c: c["name"])
def test_boundary_known_answers(case):
    buf = make(case["fixture"])
    for r in case["loads"]:
        buf.load_logical_roi(as_pair(r))
        if case.get("not_none_after_each_load"):
            assert buf.current_logical_roi_in_pixels is not None
    check_boundary_case(case, buf.current_logical_roi_in_pixels, buf.texture, buf.backing_data, buf.shape_in_pixels)
    if case["name"] == "empty_roi_handling":
        # what the reference's code does with it (_wrapping_buffer.py:170-172: the snapped ROI is empty -> return): nothing
        assert buf.current_logical_roi_in_pixels is None and not buf.texture.any()


def test_uniform_is_reversed_and_none_is_zero():
    buf = make("F2")
    assert buf.uniform() == {"offset": (0, 0, 0), "shape": (0, 0, 0), "scale": (1.0, 1.0, 1.0)}
    buf.load_logical_roi(((4, 0, 8), (4, 8, 4)))
    u = buf.uniform()
    assert u["offset"] == (8, 0, 4) and u["shape"] == (4, 8, 4)     # reversed (_wrapping_buffer.py:84-89)
