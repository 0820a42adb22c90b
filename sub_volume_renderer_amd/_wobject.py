"""``SubVolume`` — the multi-LOD volume object and its draw call.

Mirror of the reference's ``SubVolume(gfx.Volume)``
(``src/sub_volume/_wobject.py:13-226``): same constructor, validation errors,
attributes and ``center_on_position()``.  The reference is drawn by pygfx's
``renderer.render(scene, camera)`` through the plugin in ``_shader.py``; pygfx does
not exist on the target, so the draw is the explicit :meth:`SubVolume.render`,
which hands the camera matrices and the frame region to ``svr_render``
(include/svr.h) — one HIP kernel launch for vs_main + fs_main + raycast.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N
from ._geometry import Coordinate, Roi
from ._material import SubVolumeMaterial
from ._transform import _HasWorld
from ._wrapping_buffer import DeviceRings, WrappingBuffer, native_density_storage


@dataclass
class FrameRegion:
    """Which pixels of the full frame one call renders (``svr_frame``)."""

    x0: int = 0
    y0: int = 0
    out_w: int = 0
    out_h: int = 0
    band_h: int = 0       # 0: one contiguous tile
    band_pitch: int = 0

    @staticmethod
    def full(width: int, height: int) -> "FrameRegion":
        return FrameRegion(0, 0, width, height, height, height)

    @staticmethod
    def tile(x0: int, y0: int, w: int, h: int) -> "FrameRegion":
        return FrameRegion(x0, y0, w, h, h, h)

    @staticmethod
    def stripes(width: int, height: int, rank: int, nranks: int, band_h: int = 8) -> "FrameRegion":
        """Rows dealt to ranks round-robin in bands of ``band_h`` rows (padded to equal size)."""
        nbands = -(-height // band_h)
        per_rank = -(-nbands // nranks)
        return FrameRegion(0, rank * band_h, width, per_rank * band_h, band_h, band_h * nranks)


@dataclass
class RenderResult:
    """Device tensors written by one draw (fragment outputs before blending)."""

    rgba: "object"            # torch f32 [h, w, 4]   out.color   (fs_main.wgsl:86,95)
    depth: "object | None"    # torch f32 [h, w]      out.depth   (fs_main.wgsl:72,97)
    label: "object | None"    # torch i32 [h, w]      bit pattern of the u32 label (raycast.wgsl:81)
    flags: "object | None"    # torch u8  [h, w]      0 discard / 1 miss / 2 hit
    steps: "object | None"    # torch i32 [h, w]      executed march iterations (instrumented)
    pick: "object | None" = None   # torch i64 [h, w]   bit pattern of the 64-bit pick word (fs_main.wgsl:89-92)

    def label_numpy(self) -> np.ndarray:
        return self.label.cpu().numpy().view(np.uint32)


def _stop_upload_worker(jobs_queue, thread):
    """Let the upload thread finish what it holds and end (called by ``SubVolume.close`` or when a volume is collected)."""
    import threading

    jobs_queue.put(None)
    if thread.is_alive() and thread is not threading.current_thread():
        thread.join(timeout=30.0)


class SubVolume(_HasWorld):
    material: SubVolumeMaterial

    def __init__(
        self,
        material: SubVolumeMaterial,
        data_segmentation_pairs,
        buffer_shape_in_chunks,
        chunk_shape_in_pixels=None,
        *,
        device: int | None = None,
        ring_storage: str = "native",
        blocked_twin="auto",
    ):
        super().__init__()
        pairs = list(data_segmentation_pairs)
        levels = len(pairs)
        finest = pairs[0][0]

        # One value for every scale (a tuple) or one value per scale (any other sequence, whose length
        # must then equal the number of scales) — the two spellings the reference accepts (_wobject.py:34-63).
        def per_scale(value, name):
            if isinstance(value, tuple):
                return [value] * levels
            if len(value) != levels:
                raise ValueError(f"{name} list length ({len(value)}) must match number of scales ({levels})")
            return value

        buffer_shapes = per_scale(buffer_shape_in_chunks, "buffer_shape_in_chunks")
        if chunk_shape_in_pixels is None:
            # chunked array types (zarr, tensorstore) know their own chunking (_wobject.py:46-53)
            native = getattr(finest, "chunks", None)
            if native is None:
                raise ValueError("if chunk_shape_in_pixels is not provided, base data must have a 'chunks' attribute")
            chunk_shape_in_pixels = tuple(native)
        chunk_shapes = per_scale(chunk_shape_in_pixels, "chunk_shape_in_pixels")
        for level, ((density, _), chunk) in enumerate(zip(pairs, chunk_shapes)):
            if len(chunk) != density.ndim:
                raise ValueError(f"chunk_shape_in_pixels[{level}] length must match data dimensions")
        data_segmentation_pairs, num_scales, base_data = pairs, levels, finest
        # segmentations are optional here (the reference wishes for it, FUTURE.md:178-193): either every scale has
        # one or none has; without them no label ring exists and every hit is coloured with colors[0]
        unlabelled = [seg is None for _, seg in pairs]
        if any(unlabelled) and not all(unlabelled):
            raise ValueError("either every scale has a segmentation array or none has")
        if num_scales > N.SVR_MAX_LODS:
            raise ValueError(f"at most {N.SVR_MAX_LODS} scales are supported")

        self.material = material
        # all LODs share one device context (one svr_ctx); created lazily on first device use
        self._rings = DeviceRings(
            [tuple(Coordinate(b) * Coordinate(c)) for b, c in zip(buffer_shapes, chunk_shapes)],
            device=device,
            # "native": byte rings when every density source is uint8 (identical results, 4x less
            # memory traffic); "float32": always the reference's r32float layout
            density_storage=native_density_storage([d for d, _ in data_segmentation_pairs], ring_storage),
            labels=not all(unlabelled),
            # "auto": the finest scale's density ring is kept in two layouts — rows, and 128-byte micro-blocks that are
            # compact in 3-D; each wave of the march gathers from the one that suits its view (include/svr.h)
            blocked_twin=blocked_twin,
        )
        self.wrapping_buffers: list[WrappingBuffer] = []
        for i, (scale_data, scale_segmentations) in enumerate(data_segmentation_pairs):
            # same-sized voxels: lower resolutions sample at scaled-down coordinates (_wobject.py:79-81)
            scale_factor = tuple(float(scale_data.shape[j]) / float(base_data.shape[j]) for j in range(3))
            self.wrapping_buffers.append(
                WrappingBuffer(
                    backing_data=scale_data,
                    segmentations=scale_segmentations,
                    shape_in_chunks=buffer_shapes[i],
                    chunk_shape_in_pixels=chunk_shapes[i],
                    scale_factor=scale_factor,
                    _rings=self._rings,
                    _lod=i,
                )
            )
        self._volume_dimensions = np.zeros(3, np.float32)
        self.volume_dimensions = base_data.shape
        self._material_version_pushed = -1
        # pygfx gives every world object a process-wide id (the shader packs its low 20 bits into the pick word)
        SubVolume._next_id = getattr(SubVolume, "_next_id", 0) + 1
        self.id = SubVolume._next_id
        self._out_cache = {}
        self._cam_cache = None
        self._frame_cache = {}
        self._ob_cache = {}
        self._worker = None
        self._worker_error = None
        self._inflight = []
        self._submitted = 0
        self._completed = 0

    # -- _wobject.py:103-133 ---------------------------------------------------
    @property
    def volume_dimensions(self):
        """The dimensions of the volume in pixels, numpy axis order."""
        return tuple(self._volume_dimensions[::-1])

    @volume_dimensions.setter
    def volume_dimensions(self, value):
        # stored reversed = shader order, like the reference's uniform (_wobject.py:121-123)
        self._volume_dimensions = np.array(tuple(value)[::-1], dtype=np.float32)

    @property
    def textures(self):
        """All scale level textures."""
        return [b.texture for b in self.wrapping_buffers]

    @property
    def segmentations_textures(self):
        """All scale level segmentation textures."""
        return [b.segmentations_texture for b in self.wrapping_buffers]

    # -- _wobject.py:135-208 ---------------------------------------------------
    def center_on_position(self, position, sizes=None, *, asynchronous: bool = False):
        """Center every LOD's ring window on a world position (x, y, z).

        ``asynchronous=True`` (not in the reference, whose uploads block the render thread —
        FUTURE.md:3-7): the call only plans the loads and publishes the shrunk ROIs; the chunk copies
        run on a worker thread / the upload stream beside later renders, and :meth:`render` publishes
        the full new ROIs once they have landed (:meth:`poll_uploads`).

        ``sizes``: window size per scale in that scale's voxels (numpy order).  By
        default one chunk less than the ring per axis, so that growing the window
        to the chunk grid can never exceed the ring (_wobject.py:151-177).
        """
        if sizes is None:
            sizes = [
                (b.shape_in_chunks - Coordinate(1, 1, 1)) * b.chunk_shape_in_pixels
                for b in self.wrapping_buffers
            ]
        if len(sizes) != len(self.wrapping_buffers):
            raise ValueError(
                f"sizes list length ({len(sizes)}) must match number of scales ({len(self.wrapping_buffers)})"
            )
        # world -> data space; the matrix works in shader order, so reverse to numpy order
        p = (self.world.inverse_matrix @ np.array([*position, 1.0]))[:3][::-1]
        if not asynchronous and self._submitted != self._completed:
            self.poll_uploads(wait=True)                  # never interleave a blocking load with the worker's
        jobs = []
        for size, buffer in zip(sizes, self.wrapping_buffers):
            offset = tuple(int(c * f - s // 2) for c, s, f in zip(p, size, buffer.scale_factor))
            roi = Roi(offset, size)
            if buffer.can_load_logical_roi(roi):
                if asynchronous:
                    buffer._async_focus = tuple(c * f for c, f in zip(p, buffer.scale_factor))
                    pieces = buffer.begin_async_load(roi)
                    if pieces:
                        jobs.append((buffer, pieces))
                else:
                    buffer.load_logical_roi(roi)
        if jobs:
            self._submit_uploads(jobs)

    def close(self):
        """Stop the upload thread (after the loads it already holds) and free the rings and every other device
        allocation of this volume now, instead of when the object is collected."""
        stop = getattr(self, "_worker_stop", None)
        if stop is not None:
            stop()
        self._worker = None
        self._inflight.clear()
        self._completed = self._submitted
        self._out_cache, self._ob_cache = {}, {}
        self._rings.close()

    # -- asynchronous streaming ------------------------------------------------------
    @staticmethod
    def _prioritise(jobs):
        """Upload order "low res near > low res far > high res near > high res far" (FUTURE.md:86-95): the
        coarsest level first — it covers the most space, and it is what the sampler falls back to while a
        finer level's window is shrunk — and inside a level the pieces nearest to the camera first."""
        ordered = []
        for buffer, pieces in sorted(jobs, key=lambda job: -job[0]._lod):
            focus = getattr(buffer, "_async_focus", None)
            if focus is not None:
                chunk = buffer.chunk_shape_in_pixels

                def distance2(piece, focus=focus, chunk=chunk):
                    roi = piece[1]                                           # logical ROI in chunks
                    return sum(((b + 0.5 * n) * c - f) ** 2 for b, n, c, f in zip(roi.begin, roi.shape, chunk, focus))

                pieces = sorted(pieces, key=distance2)
            ordered.append((buffer, pieces))
        return ordered

    def _submit_uploads(self, jobs):
        import queue
        import threading

        import weakref

        handle = self._rings.handle                       # create the context on this thread
        for b in self.wrapping_buffers:
            b._async_owner = weakref.ref(self)
        if self._worker is None:
            # the thread holds the queue, the result list and the context handle — not the volume: a volume that is
            # dropped is collected, and its finalizer stops the thread BEFORE the device context goes away
            jobs_queue = self._jobs = queue.Queue()
            inflight = self._inflight

            def run():
                lib = N.lib()
                while True:
                    item = jobs_queue.get()
                    if item is None:
                        return
                    buffer, pieces = item
                    ticket, error = C.c_uint64(0), None
                    try:
                        for buffer_roi, logical_roi in pieces:
                            buffer.load_into_buffer(buffer_roi, logical_roi)
                        N.check(lib.svr_upload_ticket(handle, C.byref(ticket)), "svr_upload_ticket")
                    except BaseException as exc:          # surfaced by poll_uploads on the render thread
                        error = exc
                    inflight.append((buffer, ticket.value, error))
                    del item, buffer, pieces              # an idle thread keeps no level (and through it no volume) alive

            self._worker = threading.Thread(target=run, name="svr-upload", daemon=True)
            self._worker.start()
            self._worker_stop = weakref.finalize(self, _stop_upload_worker, jobs_queue, self._worker)
        # one job per level, coarse levels first: each gets its own ticket and is published as soon as
        # ITS chunks have landed
        for job in self._prioritise(jobs):
            self._submitted += 1
            self._jobs.put(job)

    def poll_uploads(self, wait: bool = False) -> bool:
        """Publish the full ROI of every asynchronous load whose chunks have landed in HBM.
        Returns True when nothing is in flight any more.  An exception raised by a backing array on the
        worker thread is re-raised here, once; that level keeps its shrunk window and re-plans next time."""
        import time

        lib = N.lib()
        while True:
            errors, replay = [], []
            while self._inflight:
                buffer, ticket, error = self._inflight[0]
                if error is None:
                    pending = C.c_int(0)
                    N.check(lib.svr_ticket_pending(self._rings.handle, ticket, C.byref(pending)), "svr_ticket_pending")
                    if pending.value:
                        break
                self._inflight.pop(0)
                self._completed += 1
                if error is None:
                    wanted = buffer.finish_async_load()
                else:
                    errors.append(error)
                    wanted = buffer.abort_async_load()
                if wanted is not None:
                    replay.append((buffer, wanted))
            jobs = []
            for buffer, roi in replay:                     # requests that arrived meanwhile: the latest one wins
                if buffer.can_load_logical_roi(roi):
                    pieces = buffer.begin_async_load(roi)
                    if pieces:
                        jobs.append((buffer, pieces))
            if jobs:
                self._submit_uploads(jobs)
            if errors:
                raise errors[0]
            if self._submitted == self._completed or not wait:
                return self._submitted == self._completed
            time.sleep(0.0002)

    # -- the draw --------------------------------------------------------------
    def _push_material(self):
        m = self.material
        if self._material_version_pushed == m._version:
            return
        u = m._u
        cm = N.Material()
        cm.clim[:] = [float(u["clim"][0]), float(u["clim"][1])]
        cm.gamma = float(u["gamma"])
        cm.opacity = float(u["opacity"])
        cm.lmip_threshold, cm.lmip_fall_off, cm.lmip_max_samples = m.lmip_uniforms()
        cm.fog_density = float(u["fog_density"])
        cm.fog_color[:] = [float(v) for v in u["fog_color"]]
        colors = np.ascontiguousarray(u["colors"], np.float32)
        cm.color_count = int(colors.shape[0])
        cm.colors = colors.ctypes.data_as(C.POINTER(C.c_float))
        # _shader.py:68: colorspace of the first texture; 'srgb' selects srgb2physical
        cm.colorspace_srgb = 1 if self.textures[0].colorspace == "srgb" else 0
        planes = np.ascontiguousarray(u["clipping_planes"], np.float32)
        cm.clipping_plane_count = int(planes.shape[0])
        cm.clipping_mode_all = 1 if u["clipping_mode"] == "ALL" else 0
        cm.clipping_planes = planes.ctypes.data_as(C.POINTER(C.c_float))
        cm.render_mode = N.SVR_MODE_WEIGHTED_AVERAGE if u["render_mode"] == "weighted_average" else N.SVR_MODE_LMIP
        cm.weight_falloff = float(u["weight_falloff"])
        N.check(N.lib().svr_set_material(self._rings.handle, C.byref(cm)), "svr_set_material")
        self._material_version_pushed = m._version

    def _camera_key(self, camera):
        w, cw = self.world, camera.world
        return (id(camera), w._position.tobytes(), w._rot.tobytes(), w._scale.tobytes(),
                cw._position.tobytes(), cw._rot.tobytes(), cw._scale.tobytes(),
                camera.fov, camera.aspect, camera.zoom, camera.near_far, self._volume_dimensions.tobytes())

    def camera_block(self, camera) -> "N.Camera":
        """The uniforms vs_main/fs_main read, as ``svr_camera`` (cached while nothing moved)."""
        key = self._camera_key(camera) if hasattr(camera, "near_far") else None
        if key is not None and self._cam_cache is not None and self._cam_cache[0] == key:
            return self._cam_cache[1]
        cb = self._camera_block_uncached(camera)
        self._cam_cache = (key, cb)
        return cb

    def _camera_block_uncached(self, camera) -> "N.Camera":
        cb = N.Camera()
        cb.world = N.mat_to_c(self.world.matrix)
        cb.world_inv = N.mat_to_c(self.world.inverse_matrix)
        cb.cam = N.mat_to_c(camera.view_matrix)
        cb.cam_inv = N.mat_to_c(camera.camera_matrix)
        cb.proj = N.mat_to_c(camera.projection_matrix)
        cb.proj_inv = N.mat_to_c(camera.projection_matrix_inverse)
        cb.volume_dimensions[:] = [float(v) for v in self._volume_dimensions]
        return cb

    def frame_block(self, width: int, height: int, region: FrameRegion | None) -> "N.Frame":
        r = region or FrameRegion.full(width, height)
        key = (width, height, r.x0, r.y0, r.out_w, r.out_h, r.band_h, r.band_pitch)
        fb = self._frame_cache.get(key)
        if fb is not None:
            return fb
        fb = self._frame_cache[key] = N.Frame()
        fb.frame_w, fb.frame_h = int(width), int(height)
        fb.x0, fb.y0, fb.out_w, fb.out_h = int(r.x0), int(r.y0), int(r.out_w), int(r.out_h)
        fb.band_h = int(r.band_h or r.out_h)
        fb.band_pitch = int(r.band_pitch or r.out_h)
        return fb

    def _outputs(self, h, w, want_steps, want_pick=False):
        import torch

        key = (h, w, bool(want_steps), bool(want_pick))
        res = self._out_cache.get(key)
        if res is None:
            dev = torch.device("cuda", self._rings.device if self._rings.device is not None else torch.cuda.current_device())
            res = RenderResult(
                rgba=torch.empty((h, w, 4), dtype=torch.float32, device=dev),
                depth=torch.empty((h, w), dtype=torch.float32, device=dev),
                label=torch.empty((h, w), dtype=torch.int32, device=dev),
                flags=torch.empty((h, w), dtype=torch.uint8, device=dev),
                steps=torch.empty((h, w), dtype=torch.int32, device=dev) if want_steps else None,
                pick=torch.empty((h, w), dtype=torch.int64, device=dev) if want_pick else None,
            )
            self._out_cache = {key: res}
        return res

    def prepare(self):
        """Push pending uniforms (material, per-LOD ROI/scale) to the device."""
        handle = self._rings.handle  # creates the context on first use
        if self._submitted != self._completed:
            self.poll_uploads()
        self._push_material()
        for b in self.wrapping_buffers:
            b._push_state()
        return handle

    def render(self, camera, width: int, height: int, *, region: FrameRegion | None = None,
               count_steps: bool = False, out: RenderResult | None = None, stream=None,
               pick: bool = False) -> RenderResult:
        """Draw this volume as seen by ``camera`` into device tensors.

        Replaces ``renderer.render(scene, camera)`` for the (SubVolume,
        SubVolumeMaterial) pair (scripts/multi_scale.py:76-80).  Asynchronous:
        the kernel is enqueued on the current torch stream.
        """
        import torch

        handle = self.prepare()
        cb = self.camera_block(camera)
        fb = self.frame_block(width, height, region)
        res = out or self._outputs(fb.out_h, fb.out_w, count_steps, pick)
        okey = (id(res), bool(count_steps), bool(pick))
        cached = self._ob_cache.get(okey)
        if cached is not None and cached[0] is res:
            ob = cached[1]
        else:
            ob = N.Outputs()
            ob.rgba = res.rgba.data_ptr()
            ob.depth = res.depth.data_ptr() if res.depth is not None else None
            ob.label = res.label.data_ptr() if res.label is not None else None
            ob.flags = res.flags.data_ptr() if res.flags is not None else None
            ob.steps = res.steps.data_ptr() if (count_steps and res.steps is not None) else None
            ob.pick = res.pick.data_ptr() if (pick and res.pick is not None) else None   # the `write_pick` shader variant
            ob.pick_id = self.id & 0xFFFFFFFF
            if len(self._ob_cache) > 8:
                self._ob_cache.clear()
            self._ob_cache[okey] = (res, ob)
        if stream is None:
            stream = torch.cuda.current_stream(self._rings.device).cuda_stream
        N.check(
            N.lib().svr_render(handle, C.byref(cb), C.byref(fb), C.byref(ob), C.c_void_p(stream)),
            "svr_render",
        )
        return res

    def synchronize(self):
        N.check(N.lib().svr_sync(self._rings.handle), "svr_sync")
