"""Scene descriptions shared by tests, smoke() and bench.py.

A :class:`SceneSpec` is *inputs only* (arrays, shapes, material arguments, camera
pose, the sequence of ``center_on_position`` calls).  The product builds a
``SubVolume`` from it (:func:`build`); the oracle (``oracle/lmip.py``) builds its
own restatement from the same spec.  This module never imports the oracle.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ._material import SubVolumeMaterial
from ._transform import AffineTransform, PerspectiveCamera
from ._wobject import FrameRegion, RenderResult, SubVolume


@dataclass
class SceneSpec:
    pairs: list                                   # [(density, labels)] per LOD, numpy-like
    chunk_shapes: list
    ring_shapes: list                             # buffer_shape_in_chunks
    material: dict
    width: int
    height: int
    cam_position: tuple = (0.0, 0.0, 0.0)
    cam_target: tuple = (0.0, 0.0, -1.0)
    fov: float = 45.0
    depth_range: tuple = (0.1, 1000.0)
    world_position: tuple = (0.0, 0.0, 0.0)
    world_scale: tuple = (1.0, 1.0, 1.0)
    centers: list = field(default_factory=list)   # [(position_xyz, sizes | None)]
    region: FrameRegion | None = None
    colorspace: str = "srgb"
    ring_storage: str = "native"
    blocked_twin: object = "auto"

    def camera(self) -> PerspectiveCamera:
        cam = PerspectiveCamera(self.fov, self.width / self.height, depth_range=self.depth_range)
        cam.world.position = self.cam_position
        cam.look_at(self.cam_target)
        return cam

    def world(self) -> AffineTransform:
        t = AffineTransform()
        t.position = self.world_position
        t.scale = self.world_scale
        return t

    def matrices(self) -> dict:
        """The six mat4 of ``svr_camera`` as row-major float32 numpy arrays."""
        cam, world = self.camera(), self.world()
        m = {
            "world": world.matrix, "world_inv": world.inverse_matrix,
            "cam": cam.view_matrix, "cam_inv": cam.camera_matrix,
            "proj": cam.projection_matrix, "proj_inv": cam.projection_matrix_inverse,
        }
        return {k: np.asarray(v, np.float32) for k, v in m.items()}


@dataclass
class BuiltScene:
    spec: SceneSpec
    volume: SubVolume
    camera: PerspectiveCamera
    width: int
    height: int


def build(spec: SceneSpec, device: int | None = None) -> BuiltScene:
    """Product side: a ``SubVolume`` with the spec's loads applied (needs the GPU)."""
    # keys that are constructor arguments in the reference vs properties inherited from pygfx's Material /
    # added here (set after construction, as a caller of the reference would)
    later = ("clipping_planes", "clipping_mode", "render_mode", "weight_falloff")
    material = SubVolumeMaterial(**{k: v for k, v in spec.material.items() if k not in later})
    for k in later:
        if k in spec.material:
            setattr(material, k, spec.material[k])
    vol = SubVolume(
        material,
        data_segmentation_pairs=list(spec.pairs),
        buffer_shape_in_chunks=list(spec.ring_shapes),
        chunk_shape_in_pixels=list(spec.chunk_shapes),
        device=device,
        ring_storage=spec.ring_storage,
        blocked_twin=spec.blocked_twin,
    )
    vol.world.position = spec.world_position
    vol.world.scale = spec.world_scale
    for b in vol.wrapping_buffers:
        b.texture.colorspace = spec.colorspace
    for position, sizes in spec.centers:
        vol.center_on_position(position, sizes)
    return BuiltScene(spec, vol, spec.camera(), spec.width, spec.height)


# ---------------------------------------------------------------------------
# canned scenes
# ---------------------------------------------------------------------------
def multiscale_demo_spec(width: int = 480, height: int = 480, tiles: int = 16) -> SceneSpec:
    """BASELINE config 1: the literal arrays, material, ring shapes and camera of the
    reference's ``scripts/multi_scale.py:31-85`` (``tiles=16`` gives its
    (256,256,768)/(128,128,768)/(64,64,768) volumes)."""
    chunk_0 = np.zeros((16, 16, 48))
    chunk_0[:4, :4, :4] = 1
    chunk_1 = np.zeros((8, 8, 48))
    chunk_1[:2, :2, :4] = 1
    chunk_2 = np.zeros((4, 4, 48))
    chunk_2[:1, :1, :4] = 1
    datas = [np.tile(c, (tiles, tiles, tiles)) for c in (chunk_0, chunk_1, chunk_2)]
    segs = [k * np.ones(d.shape, dtype=np.uint8) for k, d in enumerate(datas)]
    cam_pos = (-19.81, 7.5, 7.5)
    return SceneSpec(
        pairs=list(zip(datas, segs)),
        chunk_shapes=[(16, 16, 48), (8, 8, 48), (4, 4, 48)],
        ring_shapes=[(2, 2, 2), (4, 4, 4), (8, 8, 8)],
        material=dict(lmip_threshold=0.5, fog_density=0.01,
                      colors=[(0.0, 1.0, 1.0), (0.33, 1.0, 1.0), (0.66, 1.0, 1.0)]),
        width=width, height=height,
        cam_position=cam_pos, cam_target=(-1.0, 0.0, 0.0), fov=45.0, depth_range=(0.05, 5000.0),
        centers=[(cam_pos, None)],
    )


def make_multiscale_demo_scene(width: int = 480, height: int = 480, tiles: int = 4) -> BuiltScene:
    return build(multiscale_demo_spec(width, height, tiles))


def synthetic_spec(n: int = 64, width: int = 96, height: int = 64, *, inside: bool = False,
                   threshold: float = 0.5, full: bool = False, n_labels: int = 4096,
                   chunk_shapes=None, ring_shapes=None, sizes=None, fog_density: float = 0.01,
                   ncolors: int = 4, pairs=None) -> SceneSpec:
    """A small 3-LOD synthetic scene with the structure of BASELINE config 2
    (SURVEY.md §8d): LOD0 window around the volume centre, coarser LODs covering more."""
    from . import synth

    if pairs is None:
        pairs = [synth.volume(n, k, n_labels) for k in range(3)]
    chunk_shapes = chunk_shapes or [(8, 8, 16), (4, 4, 16), (2, 2, 16)]
    ring_shapes = ring_shapes or [(5, 5, 3), (8, 8, 2), (8, 8, 1)]
    c = (n - 1) / 2.0
    centre = np.array([c, c, c])
    if inside:
        eye = centre + np.array([0.1 * n, 0.05 * n, -0.2 * n])
        target = eye + np.array([0.6, 0.3, 0.74])
    else:
        d = np.array([-0.80, 0.36, 0.48])
        eye = centre + 1.6 * n * d / np.linalg.norm(d)
        target = centre
    colors = [(k / ncolors, 1.0, 1.0) for k in range(ncolors)]
    return SceneSpec(
        pairs=pairs, chunk_shapes=chunk_shapes, ring_shapes=ring_shapes,
        material=dict(lmip_threshold=float("inf") if full else threshold * 255.0, lmip_fall_off=0.5,
                      lmip_max_samples=10, fog_density=fog_density, fog_color=(0.5, 0.5, 0.5),
                      colors=colors, clim=(0.0, 255.0)),
        width=width, height=height,
        cam_position=tuple(eye), cam_target=tuple(target), fov=45.0,
        depth_range=(n / 500.0, n * 20.0),
        centers=[(tuple(centre), sizes)],
    )


# ---------------------------------------------------------------------------
SHARED_PLANES = ("rgba", "depth", "label", "flags", "pick")


def planes_identical(a: RenderResult, b: RenderResult) -> dict:
    """Bit-for-bit equality of every plane two renders of the same frame share (float planes compared as bit
    patterns, so a NaN equals the same NaN)."""
    import torch

    out = {}
    for name in SHARED_PLANES:
        x, y = getattr(a, name, None), getattr(b, name, None)
        if x is None or y is None:
            continue
        if x.dtype == torch.float32:
            x, y = x.view(torch.int32), y.view(torch.int32)
        out[name] = bool(torch.equal(x, y))
    return out


def render_both(volume: SubVolume, camera, width: int, height: int, **kw):
    """One frame from BOTH instantiations of the march kernel: ``production`` (``count_steps=False``: the code object
    ``bench.py`` times and every user's draw runs) and ``instrumented`` (``count_steps=True``: the COUNT build that also
    writes the exact executed-iteration counts).  They are different code objects (register allocation, spills), so
    a test that only ran the instrumented one would leave the shipped kernel unchecked: this asserts that every
    plane they share (RGBA, depth, label, flags, pick) is identical bit for bit, and returns
    ``(production, instrumented)`` for the caller to hold against the oracle."""
    import torch

    kw.pop("count_steps", None)
    out = kw.pop("out", None)
    prod = volume.render(camera, width, height, count_steps=False, **kw)
    torch.cuda.synchronize()
    prod = RenderResult(**{k: (None if getattr(prod, k) is None else getattr(prod, k).clone())
                           for k in ("rgba", "depth", "label", "flags", "steps", "pick")})
    inst = volume.render(camera, width, height, count_steps=True, out=out, **kw)
    torch.cuda.synchronize()
    same = planes_identical(prod, inst)
    assert same and all(same.values()), f"production (COUNT=false) and instrumented (COUNT=true) kernels disagree: {same}"
    return prod, inst


def hold_both_to(ref, volume: SubVolume, camera, width: int, height: int, *, tol: float = 1e-4, **kw) -> dict:
    """`render_both`, then BOTH frames against an oracle result `ref`: flags and labels bit-exact, RGBA (relative above
    1) and depth within `tol` (BASELINE.json north_star: 1e-4), step counts bit-exact for the instrumented kernel.
    Returns the instrumented kernel's report with the renders under "production" / "instrumented"."""
    prod, inst = render_both(volume, camera, width, height, **kw)
    rep = None
    for which, r in (("production", prod), ("instrumented", inst)):
        rep = compare(r, ref)
        assert rep["flags_equal"] and rep["labels_equal"], (which, rep)
        assert rep["rgba_max_rel"] <= tol and rep["depth_max_abs"] <= tol, (which, rep)
    assert rep["steps_equal"], rep
    rep["production"], rep["instrumented"] = prod, inst
    return rep


def compare(res: RenderResult, ref) -> dict:
    """Compare a device render with an oracle result (numpy arrays with the same
    attribute names).  Integer planes must match exactly; float planes are reported
    as max abs / max relative-or-abs error."""
    rgba = res.rgba.cpu().numpy()

    def err(a, b):
        """|a - b| where a NaN on both sides counts as equal (pow of a negative base: a clim above the
        sample with a fractional gamma gives NaN in the shader too) and a NaN on one side as infinite."""
        na, nb = np.isnan(a), np.isnan(b)
        with np.errstate(invalid="ignore"):
            d = np.abs(a - b)
        d = np.where(na & nb, 0.0, d)
        return np.where(na ^ nb, np.inf, d)

    e_rgba = err(rgba, ref.rgba)
    out = {
        "flags_equal": bool(np.array_equal(res.flags.cpu().numpy(), ref.flags)),
        "labels_equal": bool(np.array_equal(res.label.cpu().numpy().view(np.uint32), ref.label)),
        "rgba_max_abs": float(np.max(e_rgba)) if rgba.size else 0.0,
        "rgba_max_rel": float(np.max(e_rgba / np.maximum(1.0, np.abs(np.nan_to_num(ref.rgba))))) if rgba.size else 0.0,
        "depth_max_abs": float(np.max(err(res.depth.cpu().numpy(), ref.depth))) if rgba.size else 0.0,
        "n_hit": int(np.count_nonzero(ref.flags == 2)),
        "n_miss": int(np.count_nonzero(ref.flags == 1)),
        "n_discard": int(np.count_nonzero(ref.flags == 0)),
    }
    if res.steps is not None and getattr(ref, "steps", None) is not None:
        out["steps_equal"] = bool(np.array_equal(res.steps.cpu().numpy().view(np.uint32), ref.steps))
        out["total_steps"] = int(ref.steps.astype(np.int64).sum())
    return out


# ---------------------------------------------------------------------------
def single_voxel_spec(offset_xyz=(0, 5, 0), size: int = 33, frame: int = 65, distance: float = 80.0) -> SceneSpec:
    """A known-answer scene whose picture can be derived by hand (no oracle, no shared matrices): one bright voxel in a
    volume of zeros, ``offset_xyz`` voxels (shader order) away from the volume's centre, seen by a camera on the +x
    axis that looks at the centre down -x with +y up.  pygfx / three.js conventions (camera looks along its local -z,
    right-handed, NDC y up, pixel rows top to bottom): local x = y_cam x z_cam = (0,1,0) x (1,0,0) = -z_world, so the
    voxel must appear ``offset z`` columns LEFT and ``offset y`` rows ABOVE the centre pixel, scaled by the pixel's
    footprint: the field of view spans the mean of width and height, i.e. 2 * distance * tan(fov / 2) world units over
    ``frame`` pixels at the volume's centre."""
    c = size // 2
    data = np.zeros((size, size, size), np.float32)
    seg = np.zeros((size, size, size), np.uint32)
    x, y, z = c + offset_xyz[0], c + offset_xyz[1], c + offset_xyz[2]
    data[z, y, x] = 200.0                                       # numpy [a0, a1, a2] = texel (x = a2, y = a1, z = a0)
    seg[z, y, x] = 7
    return SceneSpec(
        pairs=[(data, seg)], chunk_shapes=[(size, size, size)], ring_shapes=[(1, 1, 1)],
        material=dict(lmip_threshold=100.0, fog_density=0.0, colors=[(0.0, 1.0, 1.0)] * 8, clim=(0.0, 200.0)),
        width=frame, height=frame, cam_position=(c + distance, float(c), float(c)), cam_target=(float(c), float(c), float(c)),
        fov=45.0, depth_range=(1.0, 1000.0), centers=[((float(c), float(c), float(c)), [(size, size, size)])])


def expected_single_voxel_pixel(offset_xyz=(0, 5, 0), frame: int = 65, distance: float = 80.0, fov: float = 45.0):
    """(row, column) of the pixel that must show the voxel of :func:`single_voxel_spec` (derivation in its docstring;
    the voxel's distance from the camera is ``distance - offset x``)."""
    per_pixel = 2.0 * np.tan(np.radians(fov) / 2.0) / frame     # world units per pixel at unit distance
    depth = distance - offset_xyz[0]
    col = (frame - 1) / 2.0 - offset_xyz[2] / (depth * per_pixel)
    row = (frame - 1) / 2.0 - offset_xyz[1] / (depth * per_pixel)
    return row, col


# ---------------------------------------------------------------------------
# micro-block copy of a density ring (svr_lod_desc::blocked_twin)
# ---------------------------------------------------------------------------
def micro_blocks_of(ring: np.ndarray) -> np.ndarray:
    """What the micro-block copy of a density ring [z][y][x] must hold (flat): 128-byte blocks of 8x4x4 (1-byte voxels),
    4x4x4 (2-byte) or 4x4x2 (4-byte) slots in [bz][by][bx] order, the slots of a block in [z][y][x] order (include/svr.h)."""
    xb, yb, zb = {1: (8, 4, 4), 2: (4, 4, 4), 4: (4, 4, 2)}[ring.dtype.itemsize]
    rz, ry, rx = ring.shape
    v = ring.reshape(rz // zb, zb, ry // yb, yb, rx // xb, xb)
    return np.ascontiguousarray(v.transpose(0, 2, 4, 1, 3, 5)).reshape(-1)


def read_micro_block_copy(rings, lod: int):
    """The micro-block copy of LOD `lod` as it lies in HBM (flat numpy array of the ring's element type), or None."""
    import ctypes as C

    import torch

    from . import _native as N

    twin = C.c_void_p()
    N.check(N.lib().svr_lod_twin_ptr(rings.handle, lod, C.byref(twin)), "svr_lod_twin_ptr")
    if not twin.value:
        return None
    N.check(N.lib().svr_sync(rings.handle), "svr_sync")
    dt = {"uint8": np.uint8, "uint16": np.uint16}.get(rings.density_storage, np.float32)
    n = int(np.prod(rings.ring_shapes[lod]))

    class _Raw:
        __cuda_array_interface__ = {"shape": (n * np.dtype(dt).itemsize,), "typestr": "|u1", "data": (twin.value, False), "version": 2}

    return torch.as_tensor(_Raw(), device="cuda").cpu().numpy().view(dt).copy()
