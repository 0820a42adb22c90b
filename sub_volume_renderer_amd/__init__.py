"""MI355X-native sub-volume LMIP ray-march renderer.

Drop-in for the hot path of gyoge0/sub_volume_renderer: the same
``SubVolume`` / ``SubVolumeMaterial`` / ``WrappingBuffer`` surface
(reference ``src/sub_volume/__init__.py:9-21``) over a C ABI (``include/svr.h``)
implemented by hand-written HIP kernels for gfx950.  There is no CPU fallback:
device work raises if ``csrc/libsvr_hip.so`` has not been built.
"""

from ._geometry import Coordinate, Roi
from .compose import compose
from ._material import SubVolumeMaterial
from ._transform import AffineTransform, PerspectiveCamera
from ._wobject import FrameRegion, RenderResult, SubVolume
from ._wrapping_buffer import WrappingBuffer, subtract_rois

__all__ = [
    "SubVolume",
    "SubVolumeMaterial",
    "WrappingBuffer",
    # replacements for what the reference imports from funlib.geometry / pygfx
    "Roi",
    "Coordinate",
    "PerspectiveCamera",
    "AffineTransform",
    "FrameRegion",
    "RenderResult",
    "subtract_rois",
    # display-side output of a render (pygfx's job in the reference)
    "compose",
]
