"""ctypes front-end of ``oracle/lmip_oracle.c`` (the CPU restatement of the
reference's shader) plus the glue that renders a ``SceneSpec`` entirely on the
CPU: ring contents from ``oracle/ring_oracle.py``, pixels from the C oracle.

TEST INFRASTRUCTURE ONLY (see the header of ``lmip_oracle.c``): used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the checker.
Render parity is *unpinned* by the reference (no runnable reference, no rendered
fixture in its tests); the ring-buffer half is pinned (``ring_oracle.py``).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

from . import ring_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "lmip_oracle.c")
_LIB = os.path.join(_HERE, "_build", "liblmip_oracle.so")
_FLAGS = ["-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC", "-shared"]


class _LodState(C.Structure):
    _fields_ = [("offset", C.c_int32 * 3), ("shape", C.c_int32 * 3), ("scale", C.c_float * 3)]


class _OracleLod(C.Structure):
    _fields_ = [("density", C.c_void_p), ("labels", C.c_void_p), ("ring_dims", C.c_int32 * 3), ("st", _LodState)]


class _Material(C.Structure):
    _fields_ = [
        ("clim", C.c_float * 2), ("gamma", C.c_float), ("opacity", C.c_float),
        ("lmip_threshold", C.c_float), ("lmip_fall_off", C.c_float), ("lmip_max_samples", C.c_int32),
        ("fog_density", C.c_float), ("fog_color", C.c_float * 3), ("color_count", C.c_uint32),
        ("colors", C.POINTER(C.c_float)), ("colorspace_srgb", C.c_int32),
        ("clipping_plane_count", C.c_uint32), ("clipping_mode_all", C.c_int32), ("clipping_planes", C.POINTER(C.c_float)),
        ("render_mode", C.c_int32), ("weight_falloff", C.c_float),
    ]


class _Camera(C.Structure):
    _fields_ = [(n, C.c_float * 16) for n in ("world", "world_inv", "cam", "cam_inv", "proj", "proj_inv")] + [
        ("volume_dimensions", C.c_float * 3)
    ]


class _Frame(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frame_w", "frame_h", "x0", "y0", "out_w", "out_h", "band_h", "band_pitch")]


_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.run(["gcc", *_FLAGS, "-o", _LIB, _SRC, "-lm"], check=True, cwd=_HERE)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.svr_oracle_render.restype = C.c_int
        _lib.svr_oracle_render.argtypes = [
            C.POINTER(_Camera), C.POINTER(_Frame), C.c_int, C.POINTER(_OracleLod), C.POINTER(_Material),
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
        ]
        _lib.svr_oracle_render_pick.restype = C.c_int
        _lib.svr_oracle_render_pick.argtypes = _lib.svr_oracle_render.argtypes[:-1] + [C.c_void_p, C.c_uint32, C.c_int]
        _lib.svr_oracle_max_threads.restype = C.c_int
    return _lib


@dataclass
class OracleResult:
    rgba: np.ndarray     # f32 [h, w, 4]
    depth: np.ndarray    # f32 [h, w]
    label: np.ndarray    # u32 [h, w]
    flags: np.ndarray    # u8  [h, w]
    steps: np.ndarray    # u32 [h, w]
    pick: np.ndarray = None   # u64 [h, w] (only when a pick id was given)


def _mat(m):
    a = np.asarray(m, np.float32).reshape(4, 4)
    return (C.c_float * 16)(*a.T.reshape(-1).tolist())      # column-major, like WGSL


DEFAULT_MATERIAL = dict(lmip_fall_off=0.5, lmip_max_samples=10, fog_density=0.5, fog_color=(0.5, 0.5, 0.5),
                        colors=None, clim=(0, 1), gamma=1.0, opacity=1.0,       # _material.py:26-37
                        clipping_planes=(), clipping_mode="ANY",                # pygfx Material defaults
                        render_mode="lmip", weight_falloff=0.5)
DEFAULT_COLORS = [(0.0, 1.0, 1.0), (0.25, 1.0, 1.0), (0.5, 1.0, 1.0), (0.75, 1.0, 1.0)]  # _material.py:51-57


def render(rings, matrices, volume_dimensions_shader, material, width, height, region=None,
           colorspace_srgb=True, nthreads=0, pick_id=None) -> OracleResult:
    """``rings``: list of dicts(density=f32 [z,y,x], labels=u32, offset, shape, scale) in shader order."""
    m = dict(DEFAULT_MATERIAL)
    m.update(material)
    colors = m["colors"] if m["colors"] is not None else DEFAULT_COLORS
    col = np.array([(*c, 1.0) for c in colors], np.float32)             # vec4-padded (_material.py:146-149)
    cm = _Material()
    cm.clim[:] = [float(np.float32(v)) for v in m["clim"]]
    cm.gamma = float(m["gamma"]); cm.opacity = float(m["opacity"])
    cm.lmip_threshold = float(m["lmip_threshold"]); cm.lmip_fall_off = float(m["lmip_fall_off"])
    cm.lmip_max_samples = int(m["lmip_max_samples"]); cm.fog_density = float(m["fog_density"])
    if m["render_mode"] == "mip":
        # MIP (FUTURE.md:97-120; pygfx's own volume raycast without its refinement) through raycast.wgsl:35-61:
        # the first sample is significant (|s| >= -inf), every larger one replaces it (strict >, :50), and the
        # loop never breaks (:58: since < 2^31 - 1, |s| < max * 0 is false).  lmip_numpy.py states MIP directly.
        cm.lmip_threshold, cm.lmip_fall_off, cm.lmip_max_samples = float("-inf"), 0.0, 2**31 - 1
    cm.render_mode = 1 if m["render_mode"] == "weighted_average" else 0      # svr_render_mode (include/svr.h)
    cm.weight_falloff = float(m["weight_falloff"])
    cm.fog_color[:] = [float(v) for v in m["fog_color"]]
    cm.color_count = len(col)
    cm.colors = col.ctypes.data_as(C.POINTER(C.c_float))
    cm.colorspace_srgb = 1 if colorspace_srgb else 0
    planes = np.ascontiguousarray(np.array(m["clipping_planes"], np.float32).reshape(-1, 4))
    cm.clipping_plane_count = len(planes)
    cm.clipping_mode_all = 1 if str(m["clipping_mode"]).upper() == "ALL" else 0
    cm.clipping_planes = planes.ctypes.data_as(C.POINTER(C.c_float))

    cam = _Camera()
    for k in ("world", "world_inv", "cam", "cam_inv", "proj", "proj_inv"):
        setattr(cam, k, _mat(matrices[k]))
    cam.volume_dimensions[:] = [float(v) for v in volume_dimensions_shader]

    fr = _Frame()
    fr.frame_w, fr.frame_h = width, height
    if region is None:
        fr.x0, fr.y0, fr.out_w, fr.out_h, fr.band_h, fr.band_pitch = 0, 0, width, height, height, height
    else:
        fr.x0, fr.y0, fr.out_w, fr.out_h = region.x0, region.y0, region.out_w, region.out_h
        fr.band_h = region.band_h or region.out_h
        fr.band_pitch = region.band_pitch or region.out_h

    lods = (_OracleLod * len(rings))()
    keep = []
    for L, r in zip(lods, rings):
        d = np.ascontiguousarray(r["density"], np.float32)
        s = np.ascontiguousarray(r["labels"], np.uint32)
        keep += [d, s]
        L.density = d.ctypes.data; L.labels = s.ctypes.data
        L.ring_dims[:] = d.shape[::-1]
        L.st.offset[:] = r["offset"]; L.st.shape[:] = r["shape"]; L.st.scale[:] = r["scale"]

    h, w = fr.out_h, fr.out_w
    out = OracleResult(np.zeros((h, w, 4), np.float32), np.zeros((h, w), np.float32),
                       np.zeros((h, w), np.uint32), np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint32))
    if pick_id is None:
        rc = lib().svr_oracle_render(C.byref(cam), C.byref(fr), len(rings), lods, C.byref(cm),
                                     out.rgba.ctypes.data, out.depth.ctypes.data, out.label.ctypes.data,
                                     out.flags.ctypes.data, out.steps.ctypes.data, int(nthreads))
    else:
        out.pick = np.zeros((h, w), np.uint64)
        rc = lib().svr_oracle_render_pick(C.byref(cam), C.byref(fr), len(rings), lods, C.byref(cm),
                                          out.rgba.ctypes.data, out.depth.ctypes.data, out.label.ctypes.data,
                                          out.flags.ctypes.data, out.steps.ctypes.data, out.pick.ctypes.data,
                                          int(pick_id), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"svr_oracle_render failed: {rc}")
    return out


class _Zeros:
    """An all-zero uint32 array of a given shape that materialises only the blocks asked of it."""

    def __init__(self, shape):
        self.shape, self.ndim, self.dtype = tuple(shape), len(shape), np.dtype(np.uint32)

    def __getitem__(self, sl):
        return np.zeros([s.stop - s.start for s in sl], np.uint32)


def oracle_volume(spec) -> ring_oracle.OracleSubVolume:
    """Replay the spec's ``center_on_position`` calls on the CPU restatement."""
    mats = spec.matrices()
    # a volume without segmentation (FUTURE.md:178-193) is a volume whose every label is 0 (the reference's
    # "unlabelled data has segmentation id 0", FUTURE.md:170-176)
    pairs = [(d, _Zeros(d.shape) if l is None else l) for d, l in spec.pairs]
    vol = ring_oracle.OracleSubVolume(pairs, list(spec.ring_shapes), list(spec.chunk_shapes),
                                      world_inverse_matrix=np.linalg.inv(spec.world().matrix))
    del mats
    for position, sizes in spec.centers:
        vol.center_on_position(position, sizes)
    return vol


def rings_of(vol: ring_oracle.OracleSubVolume) -> list:
    rings = []
    for b in vol.wrapping_buffers:
        u = b.uniform()
        rings.append(dict(density=b.texture, labels=b.segmentations_texture,
                          offset=u["offset"], shape=u["shape"], scale=u["scale"]))
    return rings


def render_spec(spec, region=None, nthreads=0, vol=None, pick_id=None) -> OracleResult:
    vol = vol or oracle_volume(spec)
    return render(rings_of(vol), spec.matrices(), vol.volume_dimensions_shader, spec.material,
                  spec.width, spec.height, region=region if region is not None else spec.region,
                  colorspace_srgb=(spec.colorspace == "srgb"), nthreads=nthreads, pick_id=pick_id)


def render_scene(scene, nthreads=0) -> OracleResult:
    return render_spec(scene.spec, nthreads=nthreads)
