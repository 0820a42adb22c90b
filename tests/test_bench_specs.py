"""bench.py's workload descriptions (no GPU): the geometry of BASELINE configs 2, 4 and 5 as the bench and the
full-size GPU tests build them, the fly-through path, and the stamp that ties `roofline.traffic` to a kernel build."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import synth  # noqa: E402


class _Shape:
    def __init__(self, n, dtype):
        self.shape, self.ndim, self.dtype = (n, n, n), 3, np.dtype(dtype)


def _pairs(n):
    return [(_Shape(n >> k, np.uint8), _Shape(n >> k, np.uint32)) for k in range(3)]


def test_config2_geometry_is_surveys_8d():
    spec = bench.config2_spec(1024, 1920, 1080, "K1", _pairs(1024))
    assert spec.chunk_shapes == [(16, 16, 48), (8, 8, 48), (4, 4, 48)]
    assert spec.ring_shapes == [(32, 32, 11), (64, 64, 11), (64, 64, 6)]
    rings = [tuple(a * b for a, b in zip(r, c)) for r, c in zip(spec.ring_shapes, spec.chunk_shapes)]
    assert rings == [(512, 512, 528), (512, 512, 528), (256, 256, 288)]
    (centre, sizes), = spec.centers
    assert sizes == [(496, 496, 480), (512, 512, 512), (256, 256, 256)]        # LOD0 default (N-1)*C, LOD1/2 whole level
    assert centre == (511.5, 511.5, 511.5)
    m = spec.material
    assert (m["lmip_threshold"], m["fog_density"], len(m["colors"])) == (0.5 * 255.0, 0.01, 4)
    # camera K1: 1.6 N from the centre along normalize(-0.80, 0.36, 0.48)
    d = np.array(spec.cam_position) - 511.5
    np.testing.assert_allclose(d / np.linalg.norm(d), np.array([-0.80, 0.36, 0.48]) / np.linalg.norm([-0.80, 0.36, 0.48]), atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(d), 1.6 * 1024)


def test_config5_is_config2_scaled_by_two_with_its_material():
    spec = bench.config5_spec(2048, 1920, 1080, "K1", _pairs(2048))
    assert spec.ring_shapes == [(64, 64, 22), (128, 128, 22), (128, 128, 12)]
    m = spec.material
    assert m["lmip_threshold"] == 0.3 * 255.0 and m["fog_density"] == 0.05 and len(m["colors"]) == 256
    assert [c[0] for c in m["colors"][:3]] == [0.0, 1 / 256, 2 / 256]
    assert spec.centers[0][1][1:] == [(1024,) * 3, (512,) * 3]


def test_config4_is_lazy_and_its_path_moves_two_voxels_per_frame():
    spec = bench.config4_spec(4096, 1920, 1080)
    assert [type(d).__name__ for d, _ in spec.pairs] == ["LazyLod"] * 3
    assert [d.shape for d, _ in spec.pairs] == [(4096,) * 3, (2048,) * 3, (1024,) * 3]
    assert spec.pairs[0][0].dtype == np.uint8 and spec.pairs[0][1].dtype == np.uint32
    assert spec.ring_shapes == [(32, 32, 11), (64, 64, 11), (64, 64, 6)] and spec.centers == [(spec.cam_position, None)]
    poses = bench.flythrough_poses(spec, 240)
    assert len(poses) == 240 and poses[0][0] == tuple(np.array(spec.cam_position, float))
    step = np.array(poses[1][0]) - np.array(poses[0][0])
    np.testing.assert_allclose(np.linalg.norm(step), 2.0)
    np.testing.assert_allclose(np.array(poses[239][0]) - np.array(poses[0][0]), 239 * step, atol=1e-9)
    view = np.array(poses[0][1]) - np.array(poses[0][0])
    np.testing.assert_allclose(np.cross(view, step), 0, atol=1e-9)           # along the view direction
    # a block of the never-resident volume is what the closed form says
    d = spec.pairs[1][0][(slice(100, 104), slice(8, 12), slice(48, 96))]
    np.testing.assert_array_equal(d, synth.block(4096, 1, (100, 8, 48), (4, 4, 48))[0])


def test_kernel_source_hash_covers_the_march_sources_and_traffic_is_stamped():
    import json

    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    path = os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "profiles", bench.PROFILE_ROUND, "traffic.json")
    assert os.path.exists(path), "tools/profile_bench.sh writes it; copy it into profiles/<round>/"
    doc = json.load(open(path))
    storages = set()
    for e in doc["entries"]:                               # one entry per ring storage of the default workload
        assert {"config", "n", "width", "height", "camera", "variant", "ring_storage", "kernel_source_sha16"} <= set(e["workload"])
        assert e["traffic_bytes_per_launch"] == int(e["FETCH_SIZE_KB"] * 1024 * 2 + e["WRITE_SIZE_KB"] * 1024)
        b = e["binding"]                                   # what the counters of the same passes show busiest
        assert b["resource"] and 0.0 < b["frac"] <= 1.0 and b["insts"] > 0
        storages.add(e["workload"]["ring_storage"])
    assert {"uint8", "float32"} <= storages
    # the committed figures should belong to the committed kernel; when they do not, bench.py prints `traffic: null` with
    # a "stale" note instead of carrying them over — reported here as an expected failure, not hidden
    if any(e["workload"]["kernel_source_sha16"] != h for e in doc["entries"]):
        pytest.xfail("march kernel changed since profiles/%s/traffic.json was taken: re-run tools/profile_bench.sh" % bench.PROFILE_ROUND)


def test_summarise():
    s = bench.summarise([3.0, 1.0, 2.0])
    assert s == {"n": 3, "median_ms": 2.0, "min_ms": 1.0, "max_ms": 3.0}


def test_config4_corridor_store_holds_exactly_what_the_flythrough_reads(tmp_path):
    """bench.py --config C4 --source zarr3 at a rehearsal size: the sparse zarr v3 store written during setup holds
    every voxel any `center_on_position` window of the path can request (equal to the lazily generated volume there),
    and nothing but the shards those windows touch."""
    import __graft_entry__ as g
    from sub_volume_renderer_amd import Roi, SubVolume, SubVolumeMaterial, zarr3

    g.build_synth()
    g.build_host_codecs()
    n = 1024
    spec = bench.config4_spec(n, 320, 200)
    spec.ring_shapes = [(9, 9, 4), (10, 9, 4), (9, 10, 3)]       # small windows keep this test to a few seconds
    poses = bench.flythrough_poses(spec, 12, step=9.0)
    positions = [spec.cam_position] + [eye for eye, _ in poses]
    touched = bench.corridor_shards(spec, positions)
    assert all(len(t) > 0 for t in touched)
    arrays, stats = bench.write_corridor_store(str(tmp_path / "store"), n, 4096, touched)
    assert stats["shards"] == 2 * sum(len(t) for t in touched) and 0 < stats["bytes_on_disk"] < stats["raw_bytes"]
    assert sorted(zarr3.open_group(str(tmp_path / "store" / "raw.zarr")).keys()) == ["scale0", "scale1", "scale2"]
    lazy = spec.pairs
    plan = SubVolume(SubVolumeMaterial(lmip_threshold=1.0), list(lazy), list(spec.ring_shapes), list(spec.chunk_shapes))
    for pos in (positions[0], positions[5], positions[-1]):
        p = (plan.world.inverse_matrix @ np.array([*pos, 1.0]))[:3][::-1]
        for lod, b in enumerate(plan.wrapping_buffers):
            size = tuple((k - 1) * c for k, c in zip(b.shape_in_chunks, b.chunk_shape_in_pixels))
            roi = b.get_snapped_roi_in_pixels(Roi(tuple(int(c * f - s // 2) for c, s, f in zip(p, size, b.scale_factor)), size))
            roi = roi.intersect(Roi((0, 0, 0), lazy[lod][0].shape))
            # one slab of the window per axis, as a ring reload would ask for it
            for axis in range(3):
                lo, hi = list(roi.begin), list(roi.end)
                hi[axis] = min(hi[axis], lo[axis] + (16 if axis < 2 else 48))
                sl = tuple(slice(int(a), int(c)) for a, c in zip(lo, hi))
                np.testing.assert_array_equal(arrays[lod][0][sl], lazy[lod][0][sl])
                np.testing.assert_array_equal(arrays[lod][1][sl], lazy[lod][1][sl])
    # sparse: a shard far from the corridor is not stored and reads as the fill value
    far = next(i for i in [(0, 0, 0), (n // 64 - 1,) * 3] if i not in touched[0])
    assert not os.path.exists(arrays[0][0]._file(far))
    assert not arrays[0][0][far[0] * 64:far[0] * 64 + 8, far[1] * 64:far[1] * 64 + 8, far[2] * 64:far[2] * 64 + 8].any()
