"""Affine transform + perspective camera: the caller-side objects the path needs.

In the reference these come from pygfx (``WorldObject.world`` /
``gfx.PerspectiveCamera``, built on pylinalg 0.6.7) and only their matrices
reach the hot path: ``world.matrix`` / ``world.inverse_matrix``
(``_wobject.py:186``; ``u_wobject.world_transform[_inv]``) and the camera's view /
projection matrices (``u_stdinfo`` in vs_main.wgsl:19-22).  These small classes
produce those six matrices; they are host plumbing, not part of the kernel
contract (``svr_camera`` takes any matrices).  Conventions follow pygfx: objects
look down their local -z with +y up, projection maps depth to [0, 1].
"""

from __future__ import annotations

import math

import numpy as np


def _normalize(v):
    v = np.asarray(v, dtype=np.float64)
    n = float(np.linalg.norm(v))
    return v / n if n > 0 else v


def _quat_to_mat(q):
    x, y, z, w = (float(c) for c in q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ],
        dtype=np.float64,
    )


class AffineTransform:
    """position / rotation / scale with ``matrix`` and ``inverse_matrix`` (float64, row-major)."""

    def __init__(self):
        self._position = np.zeros(3)
        self._rot = np.eye(3)
        self._scale = np.ones(3)

    @property
    def position(self):
        return tuple(float(v) for v in self._position)

    @position.setter
    def position(self, value):
        self._position = np.array([float(v) for v in value], dtype=np.float64)

    x = property(lambda s: float(s._position[0]))
    y = property(lambda s: float(s._position[1]))
    z = property(lambda s: float(s._position[2]))

    @property
    def scale(self):
        return tuple(float(v) for v in self._scale)

    @scale.setter
    def scale(self, value):
        if isinstance(value, (int, float)):
            value = (value, value, value)
        self._scale = np.array([float(v) for v in value], dtype=np.float64)

    def _scale_axis(i):  # noqa: N805
        def get(self):
            return float(self._scale[i])

        def set_(self, v):
            self._scale[i] = float(v)

        return property(get, set_)

    scale_x = _scale_axis(0)
    scale_y = _scale_axis(1)
    scale_z = _scale_axis(2)

    @property
    def rotation_matrix(self):
        return self._rot.copy()

    @rotation_matrix.setter
    def rotation_matrix(self, m):
        self._rot = np.asarray(m, dtype=np.float64).reshape(3, 3).copy()

    def set_rotation_quaternion(self, q):
        self._rot = _quat_to_mat(q)

    @property
    def matrix(self) -> np.ndarray:
        m = np.eye(4)
        m[:3, :3] = self._rot * self._scale[None, :]
        m[:3, 3] = self._position
        return m

    @property
    def inverse_matrix(self) -> np.ndarray:
        return np.linalg.inv(self.matrix)

    def look_at(self, target, up=(0.0, 1.0, 0.0)):
        """Rotate so that local -z points at ``target`` (pygfx ``WorldObject.look_at``)."""
        fwd = _normalize(np.asarray(target, dtype=np.float64) - self._position)
        zax = -fwd
        upv = np.asarray(up, dtype=np.float64)
        xax = np.cross(upv, zax)
        if np.linalg.norm(xax) < 1e-12:  # looking straight along `up`
            xax = np.cross(np.array([0.0, 0.0, 1.0]), zax)
            if np.linalg.norm(xax) < 1e-12:
                xax = np.array([1.0, 0.0, 0.0])
        xax = _normalize(xax)
        yax = np.cross(zax, xax)
        self._rot = np.stack([xax, yax, zax], axis=1)


class _HasWorld:
    def __init__(self):
        self.world = AffineTransform()
        self.local = self.world

    def look_at(self, target):
        self.world.look_at(target)


class PerspectiveCamera(_HasWorld):
    """Perspective camera producing the ``u_stdinfo`` matrices.

    ``fov`` in degrees applies to the mean of the view width and height, as in
    pygfx's ``PerspectiveCamera``; ``aspect`` is width / height of the frame.
    ``depth_range=(near, far)`` or ``depth`` (near = depth/1000, far = depth*1000).
    """

    def __init__(self, fov: float = 50.0, aspect: float = 1.0, *, depth: float | None = None,
                 depth_range: tuple[float, float] | None = None, zoom: float = 1.0):
        super().__init__()
        self.fov = float(fov)
        self.aspect = float(aspect)
        self.zoom = float(zoom)
        self.depth = depth
        self.depth_range = depth_range

    @property
    def near_far(self) -> tuple[float, float]:
        if self.depth_range is not None:
            return float(self.depth_range[0]), float(self.depth_range[1])
        d = 1.0 if self.depth is None else float(self.depth)
        return d / 1000.0, d * 1000.0

    @property
    def view_matrix(self) -> np.ndarray:
        """``u_stdinfo.cam_transform`` = inverse of the camera's world matrix."""
        return self.world.inverse_matrix

    @property
    def camera_matrix(self) -> np.ndarray:
        """``u_stdinfo.cam_transform_inv``."""
        return self.world.matrix

    @property
    def projection_matrix(self) -> np.ndarray:
        near, far = self.near_far
        size = 2.0 * near * math.tan(math.radians(self.fov) * 0.5) / self.zoom
        height = 2.0 * size / (1.0 + self.aspect)
        width = height * self.aspect
        r, t = 0.5 * width, 0.5 * height
        m = np.zeros((4, 4))
        m[0, 0] = near / r
        m[1, 1] = near / t
        m[2, 2] = far / (near - far)
        m[2, 3] = near * far / (near - far)
        m[3, 2] = -1.0
        return m

    @property
    def projection_matrix_inverse(self) -> np.ndarray:
        return np.linalg.inv(self.projection_matrix)
