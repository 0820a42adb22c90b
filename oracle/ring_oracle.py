"""CPU restatement of the reference's ring-buffer host logic (numpy).

TEST INFRASTRUCTURE ONLY — nothing under ``sub_volume_renderer_amd/`` may import
this module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, as the checker.

PARITY STATUS: **pinned** by the reference's own tests — every known-answer
vector of ``tests/wrapping_buffer/*`` (transcribed as data into
``tests/golden/ring_known_answers.json``) is checked against this restatement in
``tests/test_oracle_ring.py``.  The reference itself cannot be imported here
(``pygfx`` / ``funlib.geometry`` are not installed), so the ROI algebra that the
reference delegates to funlib.geometry 0.3.0 (pixi.lock:449) is restated on plain
tuples ``(offset, shape)`` from its published semantics.

Follows, line by line:
  ``src/sub_volume/_wrapping_buffer.py:118-377`` (WrappingBuffer, subtract_rois)
  ``src/sub_volume/_wobject.py:135-208``          (center_on_position)
"""

from __future__ import annotations

from itertools import product

import numpy as np

# A ROI here is a pair (offset, shape) of equal-length int tuples.


def roi(offset, shape):
    return (tuple(int(v) for v in offset), tuple(int(v) for v in shape))


def roi_begin(r):
    return r[0]


def roi_end(r):
    return tuple(o + s for o, s in zip(*r))


def roi_size(r):
    n = 1
    for s in r[1]:
        n *= s
    return n


def roi_empty(r):
    return roi_size(r) == 0


def roi_intersects(a, b):
    # funlib.geometry Roi.intersects: empty ROIs intersect nothing
    if roi_empty(a) or roi_empty(b):
        return False
    return not any(b1 >= e2 or b2 >= e1 for b1, e1, b2, e2 in zip(a[0], roi_end(a), b[0], roi_end(b)))


def roi_intersect(a, b):
    # funlib.geometry Roi.intersect: an empty ROI when they do not intersect
    if not roi_intersects(a, b):
        return roi((0,) * len(a[0]), (0,) * len(a[0]))
    begin = tuple(max(x, y) for x, y in zip(a[0], b[0]))
    end = tuple(min(x, y) for x, y in zip(roi_end(a), roi_end(b)))
    return roi(begin, tuple(e - b_ for b_, e in zip(begin, end)))


def roi_contains(a, b):
    if roi_empty(b):
        return all(lo <= c < hi for c, lo, hi in zip(b[0], a[0], roi_end(a)))
    return all(b1 <= b2 and e2 <= e1 for b1, e1, b2, e2 in zip(a[0], roi_end(a), b[0], roi_end(b)))


def roi_snap_grow(r, voxel):
    # funlib.geometry Roi.snap_to_grid(mode="grow"): floor the begin, ceil the end
    begin = tuple(b // v for b, v in zip(r[0], voxel))
    end = tuple(-((-e) // v) for e, v in zip(roi_end(r), voxel))
    return roi(tuple(b * v for b, v in zip(begin, voxel)), tuple((e - b) * v for b, e, v in zip(begin, end, voxel)))


def set_dim(coord, dim, value):
    """_wrapping_buffer.py:338-340"""
    return tuple(coord[:dim]) + (value,) + tuple(coord[dim + 1:])


def subtract_rois(roi_a, roi_b):
    """_wrapping_buffer.py:343-377"""
    if roi_empty(roi_a):                                        # :344-345
        return []
    if roi_empty(roi_b) or not roi_intersects(roi_a, roi_b):    # :346-347
        return [roi_a]
    roi_b = roi_intersect(roi_a, roi_b)                         # :349
    result = []
    base_begin = roi_begin(roi_a)                               # :352-353
    base_end = roi_end(roi_a)
    for d in range(len(roi_a[0])):                              # :355
        a0, a1 = roi_begin(roi_a)[d], roi_end(roi_a)[d]
        b0, b1 = roi_begin(roi_b)[d], roi_end(roi_b)[d]
        if a0 < b0:                                             # :360-366 slab before B
            slab_begin = base_begin
            slab_end = set_dim(base_end, d, b0)
            result.append(roi(slab_begin, tuple(e - o for o, e in zip(slab_begin, slab_end))))
            base_begin = set_dim(base_begin, d, b0)
        if b1 < a1:                                             # :369-375 slab after B
            slab_begin = set_dim(base_begin, d, b1)
            slab_end = base_end
            result.append(roi(slab_begin, tuple(e - o for o, e in zip(slab_begin, slab_end))))
            base_end = set_dim(base_end, d, b1)
    return result


class OracleWrappingBuffer:
    """_wrapping_buffer.py:8-335 with numpy arrays in place of gfx.Texture."""

    def __init__(self, backing_data, segmentations, shape_in_chunks, chunk_shape_in_pixels,
                 scale_factor=(1.0, 1.0, 1.0)):
        self.backing_data = backing_data
        self.segmentations = segmentations
        self.shape_in_chunks = tuple(int(v) for v in shape_in_chunks)                     # :45
        self.chunk_shape_in_pixels = tuple(int(v) for v in chunk_shape_in_pixels)         # :46
        self.shape_in_pixels = tuple(a * b for a, b in zip(self.shape_in_chunks, self.chunk_shape_in_pixels))  # :47
        self.texture = np.zeros(self.shape_in_pixels, np.float32)                         # :50-53
        self.segmentations_texture = np.zeros(self.shape_in_pixels, np.uint32)            # :56-59
        self.current_logical_roi_in_pixels = None                                         # :68
        self.current_logical_roi_in_chunks = None                                         # :69
        # uniform holds f32 (:113-115); getter returns them (:103)
        self.scale_factor = tuple(np.float32(x) for x in scale_factor)                    # :70
        self.uploads = []   # (buffer_roi_px, logical_roi_px) actually copied, for the tests

    # the uniform block as the shader sees it (:83-97), shader order
    def uniform(self):
        r = self.current_logical_roi_in_pixels
        off = (0, 0, 0) if r is None else r[0]
        shp = (0, 0, 0) if r is None else r[1]
        return {
            "offset": tuple(int(v) for v in off[::-1]),
            "shape": tuple(int(v) for v in shp[::-1]),
            "scale": tuple(float(v) for v in self.scale_factor[::-1]),
        }

    def get_snapped_roi_in_pixels(self, logical_roi_in_pixels):
        """:118-143"""
        data_roi = roi((0, 0, 0), self.backing_data.shape)                                # :131-132
        inter = roi_intersect(logical_roi_in_pixels, data_roi)                            # :134-136
        if roi_empty(inter):                                                              # :137-138
            return inter
        return roi_snap_grow(inter, self.chunk_shape_in_pixels)                           # :140-143

    def can_load_logical_roi(self, logical_roi_in_pixels):
        """:145-158"""
        return not any(r > b for r, b in zip(logical_roi_in_pixels[1], self.shape_in_pixels))

    def load_logical_roi(self, logical_roi_in_pixels):
        """:160-194"""
        snapped = self.get_snapped_roi_in_pixels(logical_roi_in_pixels)                   # :170
        if not self.can_load_logical_roi(logical_roi_in_pixels) or roi_empty(snapped):    # :171-172
            return
        c = self.chunk_shape_in_pixels
        in_chunks = roi(tuple(o // v for o, v in zip(snapped[0], c)),                     # :173 (exact: snapped is aligned)
                        tuple(s // v for s, v in zip(snapped[1], c)))
        if self.current_logical_roi_in_chunks is None:                                    # :178-183
            to_load = [in_chunks]
        else:
            to_load = subtract_rois(in_chunks, self.current_logical_roi_in_chunks)
        self.current_logical_roi_in_pixels = snapped                                      # :186-187
        self.current_logical_roi_in_chunks = in_chunks
        for slab in to_load:                                                              # :189-194
            for buffer_roi, logical_roi in self.wrap_logical_roi_into_buffer_rois(slab):
                self.load_into_buffer(buffer_roi, logical_roi)

    def wrap_logical_roi_into_buffer_rois(self, logical_roi_in_chunks):
        """:196-266"""
        assert len(logical_roi_in_chunks[1]) == len(self.shape_in_chunks)                 # :215-217
        for i in range(len(logical_roi_in_chunks[1])):                                    # :218-221
            assert logical_roi_in_chunks[1][i] <= self.shape_in_chunks[i]
        if roi_empty(logical_roi_in_chunks):                                              # :223-224
            return []
        offset, shape = logical_roi_in_chunks                                             # :228-232
        end = tuple(o + s for o, s in zip(offset, shape))
        grid_shape = self.shape_in_chunks
        split_coords = []
        for d in range(len(offset)):                                                      # :236-252
            start, stop, buffer = offset[d], end[d], grid_shape[d]
            if buffer == 0:
                split_coords.append([start, stop])
                continue
            boundary = (start // buffer + 1) * buffer
            if boundary < stop:
                split_coords.append([start, boundary, stop])
            else:
                split_coords.append([start, stop])
        result = []
        for corner in product(*[range(len(s) - 1) for s in split_coords]):                # :256-265
            sub_offset = tuple(split_coords[d][i] for d, i in enumerate(corner))
            sub_end = tuple(split_coords[d][i + 1] for d, i in enumerate(corner))
            sub_shape = tuple(e - o for o, e in zip(sub_offset, sub_end))
            if any(s == 0 for s in sub_shape):
                continue
            buffer_offset = tuple(o % s for o, s in zip(sub_offset, grid_shape))
            result.append((roi(buffer_offset, sub_shape), roi(sub_offset, sub_shape)))
        return result

    def load_into_buffer(self, buffer_roi_in_chunks, logical_roi_in_chunks):
        """:268-335"""
        c = self.chunk_shape_in_pixels
        buffer_px = roi(tuple(o * v for o, v in zip(buffer_roi_in_chunks[0], c)),         # :283-284
                        tuple(s * v for s, v in zip(buffer_roi_in_chunks[1], c)))
        logical_px = roi(tuple(o * v for o, v in zip(logical_roi_in_chunks[0], c)),
                         tuple(s * v for s, v in zip(logical_roi_in_chunks[1], c)))
        if roi_empty(logical_px) or roi_empty(buffer_px):                                 # :293-294
            return
        loadable = roi_intersect(roi((0, 0, 0), self.backing_data.shape), logical_px)     # :297-301
        if roi_empty(loadable):
            return
        actual_buffer = roi(buffer_px[0], loadable[1])                                    # :303-306
        src = tuple(slice(o, o + s) for o, s in zip(*loadable))                           # :312
        dst = tuple(slice(o, o + s) for o, s in zip(*actual_buffer))                      # :313
        self.texture[dst] = np.array(self.backing_data[src], dtype=np.float32)            # :325
        self.segmentations_texture[dst] = np.array(self.segmentations[src], dtype=np.uint32)  # :330-332
        self.uploads.append((actual_buffer, loadable))


class OracleSubVolume:
    """_wobject.py:20-208 without pygfx: buffers, volume_dimensions, center_on_position."""

    def __init__(self, data_segmentation_pairs, buffer_shape_in_chunks, chunk_shape_in_pixels,
                 world_inverse_matrix=None):
        base = data_segmentation_pairs[0][0]
        n = len(data_segmentation_pairs)
        buffer_shapes = [buffer_shape_in_chunks] * n if isinstance(buffer_shape_in_chunks, tuple) else buffer_shape_in_chunks
        chunk_shapes = [chunk_shape_in_pixels] * n if isinstance(chunk_shape_in_pixels, tuple) else chunk_shape_in_pixels
        self.wrapping_buffers = []
        for i, (data, seg) in enumerate(data_segmentation_pairs):
            scale = tuple(float(data.shape[j]) / float(base.shape[j]) for j in range(3))  # :79-81
            self.wrapping_buffers.append(OracleWrappingBuffer(data, seg, buffer_shapes[i], chunk_shapes[i], scale))
        self.volume_dimensions_shader = tuple(np.float32(v) for v in tuple(base.shape)[::-1])  # :121-123
        self.world_inverse_matrix = np.eye(4) if world_inverse_matrix is None else np.asarray(world_inverse_matrix, float)

    def center_on_position(self, position, sizes=None):
        """_wobject.py:135-208"""
        if sizes is None:                                                                 # :173-177
            sizes = [tuple((n - 1) * c for n, c in zip(b.shape_in_chunks, b.chunk_shape_in_pixels))
                     for b in self.wrapping_buffers]
        if len(sizes) != len(self.wrapping_buffers):                                      # :178-181
            raise ValueError("sizes list length must match number of scales")
        camera_data_pos = tuple(self.world_inverse_matrix @ np.array([*position, 1]))[:3]  # :186-188
        camera_data_pos = camera_data_pos[::-1]                                           # :190
        for size, buffer in zip(sizes, self.wrapping_buffers):                            # :191-208
            offset = tuple(int(c * f - s // 2) for c, s, f in zip(camera_data_pos, size, buffer.scale_factor))
            r = roi(offset, size)
            if buffer.can_load_logical_roi(r):
                buffer.load_logical_roi(r)


def brute_force_ring(data, ring_shape, logical_roi_px):
    """Independent expectation for a freshly loaded ROI: ``buf[pos % ring] == data[pos]``
    for every voxel of the ROI that exists in the data (test_boundary_loading.py:133-160)."""
    out = np.zeros(ring_shape, np.float32)
    (o0, o1, o2), (s0, s1, s2) = logical_roi_px
    a0 = np.arange(max(o0, 0), min(o0 + s0, data.shape[0]))
    a1 = np.arange(max(o1, 0), min(o1 + s1, data.shape[1]))
    a2 = np.arange(max(o2, 0), min(o2 + s2, data.shape[2]))
    if len(a0) and len(a1) and len(a2):
        out[np.ix_(a0 % ring_shape[0], a1 % ring_shape[1], a2 % ring_shape[2])] = data[np.ix_(a0, a1, a2)]
    return out
