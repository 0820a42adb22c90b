"""``WrappingBuffer`` — one LOD's 3D ring buffer, resident in HBM.

Host-side mirror of the reference's ``WrappingBuffer``
(``src/sub_volume/_wrapping_buffer.py:8-377``): same constructor, attributes,
method names, argument meaning, silent no-ops and assertion behaviour.  What
differs is where the voxels live.  The reference keeps a numpy mirror per
texture and lets pygfx push dirty ranges to the GPU on the next draw
(``_wrapping_buffer.py:325-335``, FUTURE.md:47-58).  Here the ring textures exist
only on the device (``svr_create``); chunk slabs travel host -> pinned staging
-> HBM on a side stream (``svr_upload_region``) in their native dtype and are
converted to f32 / u32 by the scatter kernel.  ``texture.data`` is a read-back
view for tests and debugging, not a mirror.

All ROIs handled here are in numpy axis order (a0, a1, a2); vectors are reversed
into shader order (x, y, z) only when they cross the C ABI, at the same places
the reference reverses them (``_wrapping_buffer.py:84-96,113-115``).
"""

from __future__ import annotations

import ctypes as C
from itertools import product

import numpy as np

from . import _native as N
from ._geometry import Coordinate, Roi


# ---------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------
class DeviceRings:
    """The ``svr_ctx`` that owns the ring textures of all LODs of one volume."""

    def __init__(self, ring_shapes, device: int | None = None, density_storage: str = "float32", labels: bool = True,
                 blocked_twin="auto"):
        # ring_shapes: numpy-order voxel extents, one per LOD
        self.ring_shapes = [tuple(int(v) for v in s) for s in ring_shapes]
        self.device = device
        # "float32": the reference's r32float texture.  "uint8": byte rings for uint8 sources
        # (values 0..255 are exact in f32, so every sample and result is identical)
        if density_storage not in ("float32", "uint8", "uint16"):
            raise ValueError("density_storage must be 'float32', 'uint8' or 'uint16'")
        self.density_storage = density_storage
        self.labels = bool(labels)            # False: a volume without segmentation, no label rings at all
        # Which LODs keep a second copy of their density ring in 128-byte micro-blocks (svr_lod_desc::blocked_twin), and
        # what for — per LOD 0 (none), 1 (waves whose gathers touch many rows take it INSTEAD of staging bricks) or 2 (only
        # where such a wave stages no bricks: "fallback").  "auto" = 1 for the finest LOD when its extents allow (multiples of
        # (8, 4, 4) in x, y, z); "all" = that plus 2 for the coarser LODs whose extents allow (views whose slab boxes do not
        # fit the LDS regions gain 10 - 16 %, every upload of every LOD writes twice); True = 1 for every LOD whose extents
        # allow; False = none; or one value per LOD (False / True / 2 / "fallback"; an extent that does not allow: ValueError).
        self.blocked_twin = blocked_twin_lods(self.ring_shapes, blocked_twin)
        self._twin_optional = isinstance(blocked_twin, str)      # "auto": the copy is given up when the device has no room for it
        self._handle = None
        self._closed = False

    @property
    def handle(self):
        if self._closed:
            raise RuntimeError("this volume's device context was closed")
        if self._handle is None:
            lib = N.lib()
            descs = (N.LodDesc * len(self.ring_shapes))()
            for d, s in zip(descs, self.ring_shapes):
                d.ring_dims[:] = s[::-1]
                d.density_storage = {"uint8": N.SVR_U8, "uint16": N.SVR_U16}.get(self.density_storage, N.SVR_F32)
                d.no_labels = 0 if self.labels else 1
            for d, twin in zip(descs, self.blocked_twin):
                d.blocked_twin = int(twin)
            device = self.device
            if device is None:
                import torch

                device = torch.cuda.current_device()
            h = C.c_void_p()
            status = lib.svr_create(int(device), len(descs), descs, C.byref(h))
            if status == -3 and self._twin_optional and any(self.blocked_twin):
                # no room for rings + copy (SVR_ERR_NOMEM): the copy is an optimisation, the rings are not
                for d in descs:
                    d.blocked_twin = 0
                self.blocked_twin = [0] * len(self.ring_shapes)
                status = lib.svr_create(int(device), len(descs), descs, C.byref(h))
            N.check(status, "svr_create")
            self._handle = h
            self.device = int(device)
        return self._handle

    def close(self):
        self._closed = True
        if self._handle is not None:
            N.lib().svr_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RingTexture:
    """Stand-in for the ``gfx.Texture`` of a ring (``_wrapping_buffer.py:50-59``)."""

    colorspace = "srgb"  # gfx.Texture default; selects srgb2physical (raycast.wgsl:71-72)
    dim = 3

    def __init__(self, owner: "WrappingBuffer", labels: bool):
        self._owner = owner
        self._labels = labels
        self.format = "r32uint" if labels else "r32float"

    @property
    def size(self):
        return tuple(self._owner.shape_in_pixels[::-1])

    @property
    def data(self) -> np.ndarray:
        """Read the whole ring back from HBM (numpy order)."""
        o = self._owner
        return o.read_ring(Roi((0, 0, 0), o.shape_in_pixels))[1 if self._labels else 0]


class _UniformView:
    """Read-only view with the field names of the reference's uniform buffer."""

    def __init__(self, owner):
        self._owner = owner

    @property
    def data(self):
        o = self._owner
        roi = o._current_logical_roi_in_pixels
        off = (0, 0, 0) if roi is None else tuple(roi.offset)
        shp = (0, 0, 0) if roi is None else tuple(roi.shape)
        return {
            "current_logical_offset_in_pixels": np.array(off, np.int32)[::-1],
            "current_logical_shape_in_pixels": np.array(shp, np.int32)[::-1],
            "scale_factor": np.array(o._scale_factor[::-1], np.float32),
        }


def _is_device_tensor(a) -> bool:
    return type(a).__module__.split(".")[0] == "torch" and bool(getattr(a, "is_cuda", False))


def blocked_twin_lods(ring_shapes, blocked_twin="auto") -> list[int]:
    """Per LOD: svr_lod_desc::blocked_twin — 0 no micro-block copy of the density ring, 1 a copy that waves take instead of
    staging bricks, 2 a copy for waves that stage none.  (ring_shapes in numpy order)"""
    fits = [s[2] % 8 == 0 and s[1] % 4 == 0 and s[0] % 4 == 0 for s in ring_shapes]
    if isinstance(blocked_twin, str):
        if blocked_twin not in ("auto", "all"):
            raise ValueError("blocked_twin must be 'auto', 'all', a bool or one value per LOD")
        return [(1 if i == 0 else (2 if blocked_twin == "all" else 0)) if ok else 0 for i, ok in enumerate(fits)]
    if isinstance(blocked_twin, (bool, np.bool_)):
        return [1 if (blocked_twin and ok) else 0 for ok in fits]
    wanted = [2 if (v == "fallback" or (v == 2 and v is not True)) else (1 if v else 0) for v in blocked_twin]
    if len(wanted) != len(fits):
        raise ValueError(f"blocked_twin list length ({len(wanted)}) must match number of scales ({len(fits)})")
    for i, (w, ok) in enumerate(zip(wanted, fits)):
        if w and not ok:
            raise ValueError(f"blocked_twin[{i}]: ring extents {tuple(ring_shapes[i])} are not multiples of (4, 4, 8)")
    return wanted


def native_density_storage(arrays, ring_storage: str = "native") -> str:
    """Ring element type for a set of backing density arrays: the sources' own type when every one of
    them is uint8 (or every one uint16) — such values are exact in f32, so every result is identical to the
    reference's r32float layout at a quarter (half) of the bytes — else float32."""
    if ring_storage not in ("native", "float32"):
        raise ValueError("ring_storage must be 'native' or 'float32'")
    if ring_storage == "float32":
        return "float32"
    kinds = {str(getattr(a, "dtype", "")).replace("torch.", "") for a in arrays}
    if kinds == {"uint8"}:
        return "uint8"
    if kinds == {"uint16"}:
        return "uint16"
    return "float32"


# ---------------------------------------------------------------------------
class WrappingBuffer:
    """A buffer for volumetric data that wraps around like a 3D ring."""

    uniform_type = {
        "current_logical_offset_in_pixels": "3xi4",
        "current_logical_shape_in_pixels": "3xi4",
        "scale_factor": "3xf4",
    }

    def __init__(
        self,
        backing_data,
        segmentations,
        shape_in_chunks,
        chunk_shape_in_pixels=None,
        scale_factor=(1.0, 1.0, 1.0),
        *,
        ring_storage: str = "native",
        _rings: DeviceRings | None = None,
        _lod: int = 0,
    ):
        self._ring_storage = ring_storage
        self.backing_data = backing_data
        self.segmentations = segmentations
        self.shape_in_chunks = Coordinate(shape_in_chunks)
        if chunk_shape_in_pixels is None:
            if not hasattr(backing_data, "chunks"):
                raise ValueError(
                    "if chunk_shape_in_pixels is not provided, backing data must have a 'chunks' attribute"
                )
            chunk_shape_in_pixels = backing_data.chunks
        self.chunk_shape_in_pixels = Coordinate(chunk_shape_in_pixels)
        self.shape_in_pixels = self.shape_in_chunks * self.chunk_shape_in_pixels

        self._rings = _rings
        self._lod = int(_lod)
        self.texture = RingTexture(self, labels=False)
        self.segmentations_texture = RingTexture(self, labels=True)
        self.uniform_buffer = _UniformView(self)

        self._roi_px: Roi | None = None
        self._pending_async = None
        self.superseded_requests = 0                     # asynchronous requests that were replaced by a later one before they started
        self._wanted_roi = None          # newest request that arrived while an asynchronous load was in flight
        self._async_owner = None         # weakref to the SubVolume whose upload worker serves this buffer
        self._current_logical_roi_in_chunks: Roi | None = None
        self._scale_factor = (1.0, 1.0, 1.0)
        self._state_dirty = True
        self.scale_factor = tuple(float(x) for x in scale_factor)

    # -- device plumbing -----------------------------------------------------
    @property
    def rings(self) -> DeviceRings:
        if self._rings is None:
            self._rings = DeviceRings([tuple(self.shape_in_pixels)],
                                      density_storage=native_density_storage([self.backing_data], self._ring_storage),
                                      labels=self.segmentations is not None)
            self._lod = 0
        return self._rings

    def _push_state(self):
        """``svr_set_lod_state``: the uniform write of ``_wrapping_buffer.py:83-97,113-116``."""
        if not self._state_dirty:
            return
        st = N.LodState()
        u = self.uniform_buffer.data
        st.offset[:] = [int(v) for v in u["current_logical_offset_in_pixels"]]
        st.shape[:] = [int(v) for v in u["current_logical_shape_in_pixels"]]
        st.scale[:] = [float(v) for v in u["scale_factor"]]
        N.check(N.lib().svr_set_lod_state(self.rings.handle, self._lod, C.byref(st)), "svr_set_lod_state")
        self._state_dirty = False

    # -- uniform-backed properties ---------------------------------------------
    @property
    def _current_logical_roi_in_pixels(self) -> Roi | None:
        return self._roi_px

    @_current_logical_roi_in_pixels.setter
    def _current_logical_roi_in_pixels(self, value: Roi | None):
        self._roi_px = value
        self._state_dirty = True

    @property
    def scale_factor(self) -> tuple[float, float, float]:
        """Scale of this level relative to the base resolution, numpy axis order."""
        return tuple(np.float32(v) for v in self._scale_factor)

    @scale_factor.setter
    def scale_factor(self, value):
        self._scale_factor = tuple(float(np.float32(v)) for v in value)
        self._state_dirty = True

    # -- ROI logic (pure host) ---------------------------------------------------
    def _data_roi(self) -> Roi:
        return Roi((0,) * len(self.backing_data.shape), tuple(self.backing_data.shape))

    def get_snapped_roi_in_pixels(self, logical_roi_in_pixels: Roi) -> Roi:
        """Clip to the data extent, then grow to the chunk grid (``_wrapping_buffer.py:118-143``)."""
        inside = logical_roi_in_pixels.intersect(self._data_roi())
        if inside.empty:
            return inside
        return inside.snap_to_grid(voxel_size=self.chunk_shape_in_pixels, mode="grow")

    def can_load_logical_roi(self, logical_roi_in_pixels: Roi) -> bool:
        """Size-only check against the ring extent (``_wrapping_buffer.py:145-158``)."""
        return all(r <= b for r, b in zip(logical_roi_in_pixels.shape, self.shape_in_pixels))

    def plan_logical_roi(self, logical_roi_in_pixels: Roi):
        """What ``load_logical_roi`` would do, without doing it.

        Returns ``None`` for the silent no-op cases, otherwise
        ``(snapped_roi_in_pixels, roi_in_chunks, [(buffer_roi, logical_roi), ...])``
        with the pieces in chunk units, in upload order.
        """
        snapped = self.get_snapped_roi_in_pixels(logical_roi_in_pixels)
        if not self.can_load_logical_roi(logical_roi_in_pixels) or snapped.empty:
            return None
        if (self._pending_async is None and self._roi_px is not None and self._current_logical_roi_in_chunks is not None
                and snapped == self._roi_px):
            return snapped, self._current_logical_roi_in_chunks, []      # the window did not move off its chunk grid: nothing to do
        in_chunks = snapped / self.chunk_shape_in_pixels
        if self._current_logical_roi_in_chunks is None:
            slabs = [in_chunks]
        else:
            slabs = subtract_rois(in_chunks, self._current_logical_roi_in_chunks)
        pieces = [p for slab in slabs for p in self.wrap_logical_roi_into_buffer_rois(slab)]
        return snapped, in_chunks, pieces

    def load_logical_roi(self, logical_roi_in_pixels: Roi):
        """Make the ring hold every chunk touched by the ROI (``_wrapping_buffer.py:160-194``).

        Only chunks not already resident are uploaded.  Too-large or empty
        requests are silently ignored and leave the state untouched (:171-172).
        """
        if self._pending_async is not None:
            # an asynchronous load of this ring is still streaming in: let it land (and be published) first,
            # so that this load diffs against what is really resident and the two never interleave
            self._drain_async()
        plan = self.plan_logical_roi(logical_roi_in_pixels)
        if plan is None:
            return
        snapped, in_chunks, pieces = plan
        self._current_logical_roi_in_pixels = snapped
        self._current_logical_roi_in_chunks = in_chunks
        for buffer_roi, logical_roi in pieces:
            self.load_into_buffer(buffer_roi, logical_roi)
        self.publish()

    def _drain_async(self):
        owner = self._async_owner() if self._async_owner is not None else None      # a weak reference: buffers do not keep their volume alive
        if owner is None:
            raise RuntimeError("an asynchronous load is pending on this buffer and nobody owns it")
        owner.poll_uploads(wait=True)

    def begin_async_load(self, logical_roi_in_pixels: Roi):
        """First half of an asynchronous ``load_logical_roi``: plan the load, publish the SHRUNK ROI
        (old ROI intersected with the new one) and return the upload pieces for a worker thread.

        The ring slots the new chunks will overwrite belong to chunks outside that intersection, so
        renders enqueued from now on never read a slot while it is being rewritten (the tearing the
        reference documents in FUTURE.md:60-67); coarser LODs cover the gap meanwhile.  Returns
        ``None`` for the silent no-op cases.  While a previous asynchronous load is still in flight
        the request is only remembered (the latest one wins) and replayed by :meth:`finish_async_load`.
        """
        if self._pending_async is not None:
            older = getattr(self, "_wanted_roi", None)
            if older is not None and self.get_snapped_roi_in_pixels(older) != self.get_snapped_roi_in_pixels(logical_roi_in_pixels):
                self.superseded_requests += 1            # a remembered request for ANOTHER chunk window is dropped unserved (the latest wins)
            self._wanted_roi = logical_roi_in_pixels
            return None
        self._wanted_roi = None
        plan = self.plan_logical_roi(logical_roi_in_pixels)
        if plan is None:
            return None
        snapped, in_chunks, pieces = plan
        if not pieces:                                   # nothing new to fetch: plain state change
            if snapped != self._roi_px:                  # (an unchanged window leaves the uniform alone)
                self._current_logical_roi_in_pixels = snapped
            self._current_logical_roi_in_chunks = in_chunks
            return None
        old = self._roi_px
        shrunk = None
        if old is not None:
            inter = old.intersect(snapped)
            shrunk = None if inter.empty else inter
        self._current_logical_roi_in_pixels = shrunk      # what the sampler may see while chunks stream in
        self._current_logical_roi_in_chunks = in_chunks   # what will be resident: later plans diff against it
        self._pending_async = (snapped, pieces, shrunk)
        return pieces

    def finish_async_load(self):
        """Second half: all chunks are in HBM — publish the full new ROI.  Returns the ROI of a request
        that arrived while this load was in flight (the caller starts it next), else ``None``."""
        if self._pending_async is None:
            return None
        snapped = self._pending_async[0]
        self._pending_async = None
        self._current_logical_roi_in_pixels = snapped
        wanted, self._wanted_roi = self._wanted_roi, None
        return wanted

    def abort_async_load(self):
        """The upload of an asynchronous load failed part-way: the new chunks are NOT all resident.  Keep the
        shrunk ROI published (every slot it maps still holds its old chunk: the pieces only overwrite slots
        outside it) and forget the rest, so that the next request re-plans and re-fetches what is missing."""
        if self._pending_async is None:
            return None
        shrunk = self._pending_async[2]
        self._pending_async = None
        self._current_logical_roi_in_pixels = shrunk
        self._current_logical_roi_in_chunks = None if shrunk is None else shrunk / self.chunk_shape_in_pixels
        wanted, self._wanted_roi = self._wanted_roi, None
        return wanted

    def publish(self):
        """Order the uploads before later renders and push the new ROI uniform."""
        N.check(N.lib().svr_publish_uploads(self.rings.handle), "svr_publish_uploads")
        self._push_state()

    def wrap_logical_roi_into_buffer_rois(self, logical_roi_in_chunks: Roi) -> list[tuple[Roi, Roi]]:
        """Cut a chunk-space ROI at the ring's period boundaries (``_wrapping_buffer.py:196-266``).

        The ROI is at most one ring period long per axis, so it crosses at most
        one boundary per axis: up to 2**dims ``(buffer_roi, logical_roi)`` pairs.
        """
        roi, ring = logical_roi_in_chunks, self.shape_in_chunks
        assert roi.shape.dims == ring.dims, "ROI and buffer must have same number of dimensions"
        for i in range(roi.dims):
            assert roi.shape[i] <= ring[i], (
                f"Logical ROI shape {roi.shape} cannot be larger than buffer shape {ring} in any dimension"
            )
        if roi.empty:
            return []
        # per axis: the 1 or 2 intervals [lo, hi) the ROI decomposes into
        per_axis = []
        for lo, hi, n in zip(roi.begin, roi.end, ring):
            cut = (lo // n + 1) * n if n else hi
            per_axis.append([(lo, cut), (cut, hi)] if cut < hi else [(lo, hi)])
        out = []
        for combo in product(*per_axis):
            begin = Coordinate(lo for lo, _ in combo)
            shape = Coordinate(hi - lo for lo, hi in combo)
            if 0 in shape:
                continue
            out.append((Roi(begin % ring, shape), Roi(begin, shape)))
        return out

    def load_into_buffer(self, buffer_roi_in_chunks: Roi, logical_roi_in_chunks: Roi):
        """Copy one non-wrapping block of chunks into the ring (``_wrapping_buffer.py:268-335``)."""
        dst = buffer_roi_in_chunks * self.chunk_shape_in_pixels
        src = logical_roi_in_chunks * self.chunk_shape_in_pixels
        if src.empty or dst.empty:
            return
        src = self._data_roi().intersect(src)  # only what the backing data has (:297-301)
        if src.empty:
            return
        dst = Roi(dst.offset, src.shape)  # shrink the destination to match (:303-306)
        read_roi = src
        if hasattr(self.backing_data, "origin") and hasattr(self.backing_data, "read"):
            read_roi = src + Coordinate(self.backing_data.origin)  # tensorstore (:307-310)
        density = _materialise(self.backing_data[read_roi.to_slices()])
        # a volume without segmentation (FUTURE.md:178-193): nothing to read, no label ring to write
        labels = None if self.segmentations is None else _materialise(self.segmentations[read_roi.to_slices()])
        self._upload(dst, density, labels)

    def _upload(self, dst_px: Roi, density, labels):
        lib = N.lib()
        off = N.i3(dst_px.offset[::-1])
        shp = N.i3(dst_px.shape[::-1])
        if labels is not None and _is_device_tensor(density) != _is_device_tensor(labels):
            raise TypeError("density and segmentation sources must both be host or both be device arrays")
        null = (None, 0, N.l3((0, 0, 0)))
        if _is_device_tensor(density):
            import torch

            # the source may still be being written on torch's stream; uploads run on the side stream
            torch.cuda.current_stream(density.device).synchronize()
            args = []
            for t in (density, labels):
                if t is None:
                    args += list(null)
                    continue
                dt = np.dtype(str(t.dtype).replace("torch.", ""))
                es = t.element_size()
                args += [C.c_void_p(t.data_ptr()), N.dtype_code(dt), N.l3([s * es for s in t.stride()][::-1])]
            N.check(lib.svr_upload_region_device(self.rings.handle, self._lod, off, shp, *args),
                    "svr_upload_region_device")
            return
        density = np.asarray(density)
        if tuple(density.shape) != tuple(dst_px.shape):
            raise ValueError(f"source block shape {density.shape} != destination {dst_px.shape}")
        largs = list(null)
        if labels is not None:
            labels = np.asarray(labels)
            if tuple(labels.shape) != tuple(dst_px.shape):
                raise ValueError(f"segmentation block shape {labels.shape} != destination {dst_px.shape}")
            largs = [C.c_void_p(labels.ctypes.data), N.dtype_code(labels.dtype), N.l3(labels.strides[::-1])]
        N.check(
            lib.svr_upload_region(
                self.rings.handle, self._lod, off, shp,
                C.c_void_p(density.ctypes.data), N.dtype_code(density.dtype), N.l3(density.strides[::-1]), *largs,
            ),
            "svr_upload_region",
        )

    # -- read-back -----------------------------------------------------------
    def read_ring(self, ring_roi_px: Roi) -> tuple[np.ndarray, np.ndarray]:
        """Ring voxels of a ring-space ROI as (f32 density, u32 labels), numpy order."""
        shape = tuple(ring_roi_px.shape)
        dens = np.empty(shape, np.float32)
        labs = np.empty(shape, np.uint32)
        if dens.size:
            N.check(
                N.lib().svr_read_region(
                    self.rings.handle, self._lod, N.i3(ring_roi_px.offset[::-1]), N.i3(shape[::-1]),
                    C.c_void_p(dens.ctypes.data), C.c_void_p(labs.ctypes.data),
                ),
                "svr_read_region",
            )
        return dens, labs


def _materialise(block):
    """Resolve lazy array types (tensorstore futures, zarr) to something with memory."""
    if hasattr(block, "read") and callable(block.read):
        block = block.read().result()  # tensorstore (:318-322)
    if _is_device_tensor(block):
        return block
    return np.asarray(block)


def set_dim(coord: Coordinate, dim: int, value) -> Coordinate:
    """Return a copy of coord with coord[dim] replaced by value."""
    return Coordinate(tuple(coord[:dim]) + (value,) + tuple(coord[dim + 1:]))


def subtract_rois(roi_a: Roi, roi_b: Roi) -> list[Roi]:
    """``roi_a`` minus ``roi_b`` as at most ``2 * dims`` disjoint boxes.

    Same decomposition as the reference (``_wrapping_buffer.py:343-377``): peel the
    part of A below and above B along axis 0, then continue inside B's extent on
    that axis with axis 1, and so on.
    """
    if roi_a.empty:
        return []
    if roi_b.empty or not roi_a.intersects(roi_b):
        return [roi_a]
    core = roi_a.intersect(roi_b)
    lo, hi = list(roi_a.begin), list(roi_a.end)  # the not-yet-peeled remainder of A
    slabs = []
    for d in range(roi_a.dims):
        if lo[d] < core.begin[d]:
            top = list(hi)
            top[d] = core.begin[d]
            slabs.append(Roi(lo, [e - b for b, e in zip(lo, top)]))
            lo[d] = core.begin[d]
        if core.end[d] < hi[d]:
            bottom = list(lo)
            bottom[d] = core.end[d]
            slabs.append(Roi(bottom, [e - b for b, e in zip(bottom, hi)]))
            hi[d] = core.end[d]
    return slabs
