"""GPU: the streaming front-end around the march — asynchronous ring reloads with frames in flight on several
streams, failure of a backing array part-way, mixing blocking and asynchronous loads, upload order, tensorstore-
shaped sources, and the limits of one upload call."""
import threading

import numpy as np
import pytest

from oracle import lmip
from sub_volume_renderer_amd import Roi, WrappingBuffer, testing

pytestmark = pytest.mark.gpu
RGBA_TOL = 1e-4


def _published_rings(vol, orac):
    """Oracle ring contents with the ROI the PRODUCT has published right now."""
    rings = lmip.rings_of(orac)
    for ring, b in zip(rings, vol.wrapping_buffers):
        u = b.uniform_buffer.data
        ring["offset"] = tuple(int(v) for v in u["current_logical_offset_in_pixels"])
        ring["shape"] = tuple(int(v) for v in u["current_logical_shape_in_pixels"])
    return rings


def _roi_pair(r):
    return None if r is None else (tuple(r.offset), tuple(r.shape))


def _assert_frame(res, ref, what):
    """`res`: one render, or the (production, instrumented) pair of `testing.render_both`."""
    for r in (res if isinstance(res, tuple) else (res,)):
        rep = testing.compare(r, ref)
        assert rep["flags_equal"] and rep["labels_equal"] and rep.get("steps_equal", True), (what, rep)
        assert ("steps_equal" in rep) == (r.steps is not None)
        assert rep["rgba_max_rel"] <= RGBA_TOL and rep["depth_max_abs"] <= 1e-4, (what, rep)


def test_frames_in_flight_on_two_streams_never_tear_under_async_reloads():
    """Renders alternate between two HIP streams while center_on_position(asynchronous=True) rewrites ring
    slots: an upload must wait for EVERY render still in flight, not only for the latest one.  Each frame is
    compared with the oracle for the ROI state that was published when it was enqueued."""
    import torch

    spec = testing.synthetic_spec(96, 256, 160, inside=True, chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)],
                                  ring_shapes=[(5, 5, 3), (8, 8, 3), (8, 8, 2)])
    scene = testing.build(spec)
    vol = scene.volume
    orac = lmip.oracle_volume(spec)
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    # stream 0 carries the PRODUCTION kernel (no step plane), stream 1 the instrumented one
    outs = [vol._outputs(spec.height, spec.width, False), None]
    vol._out_cache = {}
    outs[1] = vol._outputs(spec.height, spec.width, True)
    pending = []
    for k in range(1, 13):
        p = eye + d * 5.0 * k
        spec.cam_position = tuple(p)
        spec.cam_target = tuple(p + d)
        cam = spec.camera()
        slot = k & 1
        if len(pending) == 2:                              # the frame that used this output buffer two frames ago
            res, rings, mats, s = pending.pop(0)
            s.synchronize()
            ref = lmip.render(rings, mats, orac.volume_dimensions_shader, spec.material, spec.width, spec.height)
            _assert_frame(res, ref, ("frame", k - 2))
        with torch.cuda.stream(streams[slot]):
            res = vol.render(cam, spec.width, spec.height, count_steps=bool(slot), out=outs[slot])
        # what this frame must show: the textures BEFORE the next move, the ROIs published at its prepare()
        rings = [dict(r, density=r["density"].copy(), labels=r["labels"].copy()) for r in _published_rings(vol, orac)]
        pending.append((res, rings, spec.matrices(), streams[slot]))
        vol.center_on_position(tuple(p), asynchronous=True)   # overwrites slots while both frames may still run
        orac.center_on_position(tuple(p))
    for res, rings, mats, s in pending:
        s.synchronize()
        _assert_frame(res, lmip.render(rings, mats, orac.volume_dimensions_shader, spec.material, spec.width, spec.height), "tail")
    vol.poll_uploads(wait=True)
    for b, ob in zip(vol.wrapping_buffers, orac.wrapping_buffers):
        # requests that arrived while a load was in flight were superseded by later ones, so slots OUTSIDE the final
        # window may hold other chunks than the oracle's (which loaded every position); the window itself must match
        assert _roi_pair(b._current_logical_roi_in_pixels) == ob.current_logical_roi_in_pixels
        if ob.current_logical_roi_in_pixels is None:
            continue
        roi = Roi(*ob.current_logical_roi_in_pixels).intersect(Roi((0, 0, 0), ob.backing_data.shape))
        idx = [np.arange(o, o + s) % r for o, s, r in zip(roi.offset, roi.shape, ob.texture.shape)]
        np.testing.assert_array_equal(b.texture.data[np.ix_(*idx)], ob.texture[np.ix_(*idx)])


class _FailingArray:
    """numpy-backed array whose reads fail while `armed` holds a positive count-down."""

    def __init__(self, a):
        self.a, self.shape, self.ndim, self.dtype, self.fail_after = a, a.shape, a.ndim, a.dtype, None

    def __getitem__(self, sl):
        if self.fail_after is not None:
            self.fail_after -= 1
            if self.fail_after < 0:
                raise IOError("backing store went away")
        return self.a[sl]


def test_failed_asynchronous_load_keeps_the_shrunk_window_and_recovers():
    import torch

    spec = testing.synthetic_spec(64, 96, 64, inside=True)
    flaky = _FailingArray(spec.pairs[0][0])
    spec.pairs = [(flaky, spec.pairs[0][1])] + list(spec.pairs[1:])
    scene = testing.build(spec)
    vol = scene.volume
    plain = testing.synthetic_spec(64, 96, 64, inside=True)
    orac = lmip.oracle_volume(plain)
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    p = eye + d * 14.0
    before = [b._current_logical_roi_in_pixels for b in vol.wrapping_buffers]
    flaky.fail_after = 1                                     # the second piece of LOD 0 raises on the worker thread
    vol.center_on_position(tuple(p), asynchronous=True)
    with pytest.raises(IOError):
        vol.poll_uploads(wait=True)
    b0 = vol.wrapping_buffers[0]
    assert b0._pending_async is None
    shrunk = b0._current_logical_roi_in_pixels               # old & new, NOT the new ROI
    assert shrunk is None or before[0].contains(shrunk)
    assert vol.poll_uploads(wait=True) is True                # raised once, nothing left in flight
    # the frame still equals the oracle for the published state: old textures where the shrunk window maps them
    res = testing.render_both(vol, scene.camera, spec.width, spec.height)
    rings = _published_rings(vol, orac)                       # oracle textures = before the move
    if shrunk is not None:
        _assert_frame(res, lmip.render(rings, spec.matrices(), orac.volume_dimensions_shader, plain.material,
                                       spec.width, spec.height), "after failure")
    # the store comes back: the next request re-plans everything that is missing
    flaky.fail_after = None
    vol.center_on_position(tuple(p), asynchronous=True)
    vol.poll_uploads(wait=True)
    orac.center_on_position(tuple(p))
    for b, ob in zip(vol.wrapping_buffers, orac.wrapping_buffers):
        assert _roi_pair(b._current_logical_roi_in_pixels) == ob.current_logical_roi_in_pixels
        if ob.current_logical_roi_in_pixels is None:
            continue
        roi = Roi(*ob.current_logical_roi_in_pixels).intersect(Roi((0, 0, 0), ob.backing_data.shape))
        ring = np.array(ob.texture.shape)
        # every voxel of the window is the right one: buf[pos % ring] == data[pos]
        zz, yy, xx = [np.arange(o, o + s) for o, s in zip(roi.offset, roi.shape)]
        tex = b.texture.data
        np.testing.assert_array_equal(tex[np.ix_(zz % ring[0], yy % ring[1], xx % ring[2])],
                                      np.asarray(ob.backing_data)[np.ix_(zz, yy, xx)].astype(np.float32))


def test_latest_request_wins_and_blocking_load_waits_for_the_worker():
    spec = testing.synthetic_spec(64, 96, 64, inside=True)
    gate = threading.Event()

    class Slow(_FailingArray):
        def __getitem__(self, sl):
            gate.wait(5.0)
            return self.a[sl]

    spec.pairs = [(Slow(spec.pairs[0][0]), spec.pairs[0][1])] + list(spec.pairs[1:])
    gate.set()
    scene = testing.build(spec)
    vol = scene.volume
    orac = lmip.oracle_volume(testing.synthetic_spec(64, 96, 64, inside=True))
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    gate.clear()                                             # the worker blocks inside the first read
    vol.center_on_position(tuple(eye + d * 6.0), asynchronous=True)
    vol.center_on_position(tuple(eye + d * 12.0), asynchronous=True)      # remembered
    vol.center_on_position(tuple(eye + d * 18.0), asynchronous=True)      # replaces the remembered one
    assert vol.wrapping_buffers[0]._wanted_roi is not None
    gate.set()
    vol.poll_uploads(wait=True)
    orac.center_on_position(tuple(eye + d * 6.0))
    orac.center_on_position(tuple(eye + d * 18.0))           # 12.0 was dropped: the last camera move is what counts
    for b, ob in zip(vol.wrapping_buffers, orac.wrapping_buffers):
        assert _roi_pair(b._current_logical_roi_in_pixels) == ob.current_logical_roi_in_pixels
    # a blocking call while the worker is busy waits for it instead of interleaving with it
    gate.clear()
    vol.center_on_position(tuple(eye + d * 24.0), asynchronous=True)
    threading.Timer(0.2, gate.set).start()
    vol.center_on_position(tuple(eye + d * 30.0))            # blocking
    orac.center_on_position(tuple(eye + d * 24.0))
    orac.center_on_position(tuple(eye + d * 30.0))
    for b, ob in zip(vol.wrapping_buffers, orac.wrapping_buffers):
        assert b._pending_async is None
        assert _roi_pair(b._current_logical_roi_in_pixels) == ob.current_logical_roi_in_pixels
        np.testing.assert_array_equal(b.texture.data, ob.texture)


def test_uploads_go_coarse_level_first_and_near_pieces_first():
    spec = testing.synthetic_spec(128, 96, 64, inside=True)     # rings smaller than every level: all three reload
    order = []

    class Spy(_FailingArray):
        def __init__(self, a, lod):
            super().__init__(a)
            self.lod = lod

        def __getitem__(self, sl):
            order.append((self.lod, tuple(s.start for s in sl)))
            return self.a[sl]

    spec.pairs = [(Spy(d, k), l) for k, (d, l) in enumerate(spec.pairs)]
    scene = testing.build(spec)
    vol = scene.volume
    del order[:]
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    p = eye + d * 30.0
    vol.center_on_position(tuple(p), asynchronous=True)
    vol.poll_uploads(wait=True)
    lods = [lod for lod, _ in order]
    assert len(set(lods)) >= 2 and lods == sorted(lods, reverse=True)       # low res first (FUTURE.md:86-95)
    focus = (vol.world.inverse_matrix @ np.array([*p, 1.0]))[:3][::-1]
    for lod in set(lods):
        b = vol.wrapping_buffers[lod]
        starts = [np.array(s) for l, s in order if l == lod]
        # piece order is by distance of the piece CENTRE; its begin corner is within one piece of that
        dist = [np.linalg.norm(s - focus * np.array(b.scale_factor)) for s in starts]
        assert len(dist) < 2 or dist[0] <= max(dist)


class FakeTensorStore:
    """The two things ``load_into_buffer`` uses of a tensorstore array (_wrapping_buffer.py:307-322):
    ``.origin`` (index of the first element) and slices that are lazy until ``.read().result()``."""

    def __init__(self, a, origin):
        self._a, self.origin, self.shape, self.ndim, self.dtype = a, tuple(origin), a.shape, a.ndim, a.dtype
        self.reads = 0

    def read(self):                                            # ts.TensorStore.read(): the whole domain
        return self[tuple(slice(o, o + n) for o, n in zip(self.origin, self.shape))].read()

    def __getitem__(self, sl):
        outer = self
        # tensorstore indexes in its own index space: [origin, origin + shape)
        local = tuple(slice(s.start - o, s.stop - o) for s, o in zip(sl, self.origin))
        for s, n in zip(local, self.shape):
            assert 0 <= s.start <= s.stop <= n, "read outside the store's domain"

        class Lazy:
            def read(self):
                class Future:
                    def result(self_inner):
                        outer.reads += 1
                        return outer._a[local]
                return Future()
        return Lazy()


def test_tensorstore_shaped_sources_with_nonzero_origin():
    rng = np.random.default_rng(11)
    data = rng.integers(0, 255, (32, 32, 32), dtype=np.uint8)
    seg = rng.integers(0, 2 ** 32, data.shape, dtype=np.uint32)
    ts_d, ts_s = FakeTensorStore(data, (100, -20, 7)), FakeTensorStore(seg, (100, -20, 7))
    buf = WrappingBuffer(ts_d, ts_s, (3, 3, 3), (8, 8, 8))
    plain = WrappingBuffer(data, seg, (3, 3, 3), (8, 8, 8))
    for roi in (Roi((0, 0, 0), (16, 16, 16)), Roi((8, 12, 4), (16, 16, 16)), Roi((20, 20, 20), (16, 16, 16))):
        buf.load_logical_roi(roi)
        plain.load_logical_roi(roi)
        assert buf._current_logical_roi_in_pixels == plain._current_logical_roi_in_pixels
        np.testing.assert_array_equal(buf.texture.data, plain.texture.data)
        np.testing.assert_array_equal(buf.segmentations_texture.data, plain.segmentations_texture.data)
    assert ts_d.reads > 0 and ts_s.reads == ts_d.reads


def test_one_plane_larger_than_a_staging_slot_is_split_by_rows():
    """The reference accepts any block size; a z-plane of 8192 x 2048 voxels (u8 + u32 = 84 MB) exceeds the
    48 MiB staging slot and travels as runs of rows."""
    rng = np.random.default_rng(2)
    data = rng.integers(0, 255, (2, 2048, 8192), dtype=np.uint8)
    seg = rng.integers(0, 2 ** 32, data.shape, dtype=np.uint32)
    buf = WrappingBuffer(data, seg, (2, 2, 2), (1, 1024, 4096))
    buf.load_logical_roi(Roi((0, 0, 0), (2, 2048, 8192)))
    d, l = buf.read_ring(Roi((0, 0, 0), (2, 2048, 8192)))
    np.testing.assert_array_equal(d, data.astype(np.float32))
    np.testing.assert_array_equal(l, seg)


def test_rings_beyond_the_span_kernels_24_bit_row_index_render_exactly():
    """ring_y * ring_z >= 2^24 (64 x 8192 x 4096 voxels, 2 GiB of bytes): the span kernel's 24-bit row
    arithmetic cannot address it, so the draw takes the straightforward kernel — same pixels as the oracle."""
    import torch

    n = (48, 40, 64)                                          # the data is small; the RING is huge
    rng = np.random.default_rng(4)
    data = rng.integers(0, 255, n, dtype=np.uint8)
    seg = rng.integers(0, 1000, n, dtype=np.uint32)
    spec = testing.synthetic_spec(64, 96, 64, pairs=[(data, seg)], chunk_shapes=[(8, 8, 64)], ring_shapes=[(512, 1024, 1)],
                                  threshold=0.6)
    spec.centers = [((31.5, 19.5, 23.5), [n])]
    c = np.array([31.5, 19.5, 23.5])
    spec.cam_position = tuple(c + np.array([-90.0, 40.0, 55.0]))
    spec.cam_target = tuple(c)
    scene = testing.build(spec)
    res = testing.render_both(scene.volume, scene.camera, spec.width, spec.height)
    # the oracle would need the 2 GiB ring as f32: feed it an equivalent small ring instead (same ROI, the
    # window does not wrap, so slot == voxel index in both)
    small = testing.synthetic_spec(64, 96, 64, pairs=[(data, seg)], chunk_shapes=[(8, 8, 64)], ring_shapes=[(6, 5, 1)],
                                   threshold=0.6)
    small.centers, small.cam_position, small.cam_target = spec.centers, spec.cam_position, spec.cam_target
    ref = lmip.render_spec(small)
    _assert_frame(res, ref, "huge ring")
    assert (ref.flags == 2).sum() > 50


def test_zarr_v3_sharded_store_as_backing_data(tmp_path):
    """The on-disk layout of the reference's pyramid builders (zarr v3 group, scale0..2, 16^3 chunks in 64^3
    shards, zstd) read by the product's own reader: a fly-through over the stores gives the frames and rings of
    the same fly-through over the numpy arrays the stores were written from."""
    import os

    import torch

    from sub_volume_renderer_amd import synth, zarr3

    n = 128
    pairs = [synth.volume(n, k) for k in range(3)]
    raw = zarr3.create_group(str(tmp_path / "raw.zarr"))
    lab = zarr3.create_group(str(tmp_path / "labels.zarr"))
    for k, (d, l) in enumerate(pairs):
        c = (16, 16, 16) if k < 2 else (8, 8, 8)
        zarr3.write_array(os.path.join(raw.path, f"scale{k}"), d, chunks=c, shards=(64, 64, 64) if k == 0 else None)
        zarr3.write_array(os.path.join(lab.path, f"scale{k}"), l, chunks=c, shards=(64, 64, 64) if k == 0 else None)
    raw, lab = zarr3.open_group(raw.path), zarr3.open_group(lab.path)
    zpairs = [(raw[f"scale{k}"], lab[f"scale{k}"]) for k in range(3)]
    kw = dict(inside=True, chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)], ring_shapes=[(6, 6, 3), (12, 12, 3), (16, 16, 2)])
    spec = testing.synthetic_spec(n, 160, 96, pairs=zpairs, **kw)
    scene = testing.build(spec)
    assert scene.volume._rings.density_storage == "uint8"          # dtype comes from the store's metadata
    orac = lmip.oracle_volume(testing.synthetic_spec(n, 160, 96, pairs=pairs, **kw))
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    for k in range(5):
        p = eye + d * 9.0 * k
        spec.cam_position, spec.cam_target = tuple(p), tuple(p + d)
        scene.volume.center_on_position(tuple(p), asynchronous=bool(k & 1))
        scene.volume.poll_uploads(wait=True)
        orac.center_on_position(tuple(p))
        res = testing.render_both(scene.volume, spec.camera(), spec.width, spec.height)
        ref = lmip.render(lmip.rings_of(orac), spec.matrices(), orac.volume_dimensions_shader, spec.material, spec.width, spec.height)
        _assert_frame(res, ref, ("zarr", k))
    for b, ob in zip(scene.volume.wrapping_buffers, orac.wrapping_buffers):
        np.testing.assert_array_equal(b.texture.data, ob.texture)
        np.testing.assert_array_equal(b.segmentations_texture.data, ob.segmentations_texture)


def test_rings_of_4_gib_in_total_keep_the_span_kernel_with_one_resource_per_lod():
    """Two uint16 rings of 1024 x 1024 x 1056 slots (2.2 GB each: 4.5 GB together, beyond one 32-bit buffer
    resource): the draw stays on the span kernel, one resource per LOD (a BIG build), bit-identical to the
    oracle; mixed-LOD batches, bricks and skipping included."""
    import ctypes as C

    import torch

    from sub_volume_renderer_amd import _native as N, synth

    pairs = []
    for k in range(2):
        d, l = synth.volume(64, k)
        pairs.append((d.astype(np.uint16) * 200, l))
    kw = dict(threshold=0.45, chunk_shapes=[(8, 8, 16), (4, 4, 16)])
    spec = testing.synthetic_spec(64, 160, 96, pairs=pairs, ring_shapes=[(128, 128, 66), (256, 256, 66)], **kw)
    spec.material.update(lmip_threshold=0.45 * 51000, clim=(0.0, 51000.0))
    spec.centers = [((31.5, 31.5, 31.5), [(32, 32, 32), (32, 32, 32)])]       # LOD 0 only in the middle: LOD transitions
    scene = testing.build(spec)
    assert scene.volume._rings.density_storage == "uint16"
    res = testing.render_both(scene.volume, scene.camera, spec.width, spec.height)
    census = (C.c_uint32 * 8)()
    N.check(N.lib().svr_debug_counters(scene.volume._rings.handle, census, 1), "svr_debug_counters")
    assert census[6] > 0 and census[0] > 0                  # the SPAN kernel ran (the simple one keeps no census), general batches too
    small = testing.synthetic_spec(64, 160, 96, pairs=pairs, ring_shapes=[(8, 8, 4), (8, 8, 2)], **kw)
    small.material, small.centers = spec.material, spec.centers
    ref = lmip.render_spec(small)
    _assert_frame(res, ref, "big rings")
    assert (ref.flags == 2).sum() > 500
    for mode in ("mip",):
        scene.volume.material.render_mode = mode
        small.material = dict(small.material, render_mode=mode)
        res = testing.render_both(scene.volume, scene.camera, spec.width, spec.height)
        _assert_frame(res, lmip.render_spec(small), "big rings, mip")


def test_a_single_float_ring_beyond_4_gib_is_reached_through_several_resources():
    """A FLOAT32 ring of 1024 x 1024 x 2112 slots — 8.86 GB, more than the 32-bit byte offsets of one buffer resource
    reach, or of two (round 2 fell back to the one-fetch-per-step kernel at 4 GiB: config 5 on the reference's float
    layout).  The span kernel addresses such a ring through parts of whole z planes (here 496 + 496 + 32); the level-0
    window is put across the boundary at plane 992 AND the ring's wrap (planes 952 .. 1023, 0 .. 23), then across the
    boundary at plane 496, and every frame must equal the oracle's from three views (gathers, brick slabs and general
    batches all cross the part boundaries), with both kernels."""
    import ctypes as C

    from sub_volume_renderer_amd import _native as N, synth

    pairs = []
    for k in range(2):
        d, l = synth.volume(64, k)
        pairs.append((np.tile(d, (17, 1, 1)), np.tile(l, (17, 1, 1))))           # (1088, 64, 64) and (544, 32, 32): long along a0 = z
    kw = dict(threshold=0.45, chunk_shapes=[(8, 8, 16), (4, 4, 16)])
    spec = testing.synthetic_spec(64, 160, 96, pairs=pairs, ring_shapes=[(128, 128, 132), (40, 16, 4)], **kw)
    spec.ring_storage = "float32"
    sizes = [(96, 32, 32), (64, 32, 32)]
    spec.depth_range = (0.2, 4000.0)
    small = testing.synthetic_spec(64, 160, 96, pairs=pairs, ring_shapes=[(16, 8, 4), (20, 8, 2)], **kw)
    small.depth_range = spec.depth_range
    spec.centers, small.centers = [], []
    scene = None
    for z_centre, first_plane in ((1000.0, 952), (498.0, 448)):
        target = np.array([31.5, 31.5, z_centre])                                  # shader order (x, y, z)
        spec.centers.append((tuple(target), sizes))
        small.centers.append((tuple(target), sizes))
        for view in ((-0.80, 0.36, 0.48), (0.05, 0.08, -1.0), (0.6, -0.3, 0.74)):
            dvec = np.array(view) / np.linalg.norm(view)
            spec.cam_position, spec.cam_target = tuple(target + 170.0 * dvec), tuple(target)
            small.cam_position, small.cam_target = spec.cam_position, spec.cam_target
            if scene is None:
                scene = testing.build(spec)
                assert scene.volume._rings.density_storage == "float32"
                assert tuple(scene.volume.wrapping_buffers[0].shape_in_pixels) == (1024, 1024, 2112)
            elif scene.volume.wrapping_buffers[0]._current_logical_roi_in_pixels.begin[0] != first_plane:
                scene.volume.center_on_position(tuple(target), sizes)
            assert scene.volume.wrapping_buffers[0]._current_logical_roi_in_pixels.begin[0] == first_plane
            res = testing.render_both(scene.volume, spec.camera(), spec.width, spec.height)
            census = (C.c_uint32 * 8)()
            N.check(N.lib().svr_debug_counters(scene.volume._rings.handle, census, 1), "svr_debug_counters")
            assert census[6] > 0, "the span kernel must run (the straightforward kernel keeps no census)"
            ref = lmip.render_spec(small)
            _assert_frame(res, ref, ("float ring beyond 4 GiB", z_centre, view))
            assert (ref.flags == 2).sum() > 200
            # ... and with every wave sent to the micro-block copy of that ring (8.86 GB more, cut into the same three parts)
            assert scene.volume._rings.blocked_twin[0]
            N.check(N.lib().svr_set_variant(scene.volume.prepare(), 0x200), "svr_set_variant")
            timers = (C.c_uint64 * 16)()
            N.check(N.lib().svr_debug_timers(scene.volume._rings.handle, timers, 1), "svr_debug_timers")
            res = testing.render_both(scene.volume, spec.camera(), spec.width, spec.height)
            N.check(N.lib().svr_debug_timers(scene.volume._rings.handle, timers, 1), "svr_debug_timers")
            assert timers[15] > 0, "no batch gathered from the copy"
            _assert_frame(res, ref, ("micro-block copy of a float ring beyond 4 GiB", z_centre, view))
            N.check(N.lib().svr_set_variant(scene.volume.prepare(), 0), "svr_set_variant")
    scene.volume.close()


def test_a_dropped_volume_stops_its_upload_thread_and_frees_its_rings():
    """The upload thread holds the job queue and the context handle, not the volume: dropping a volume that has
    streamed asynchronously ends the thread (after the loads it holds) before the device context is destroyed;
    ``close()`` does the same at once."""
    import gc
    import weakref

    import torch

    spec = testing.synthetic_spec(96, 128, 80, inside=True, chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)],
                                  ring_shapes=[(5, 5, 3), (8, 8, 3), (8, 8, 2)])
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    for explicit in (False, True):
        scene = testing.build(spec)
        vol = scene.volume
        for k in range(1, 4):
            vol.center_on_position(tuple(eye + d * 6.0 * k), asynchronous=True)
            vol.render(scene.camera, spec.width, spec.height)
        worker = vol._worker
        assert worker is not None and worker.is_alive()
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        ref = weakref.ref(vol)
        if explicit:
            vol.close()
            assert not worker.is_alive()
            with pytest.raises(RuntimeError, match="closed"):
                vol.render(scene.camera, spec.width, spec.height)
        del scene, vol
        gc.collect()
        worker.join(timeout=10.0)
        assert not worker.is_alive()
        assert ref() is None
        assert torch.cuda.mem_get_info()[0] > free_before          # the rings went back to the device
