"""Display-side output of one render: blend over a background, depth test, sRGB 8-bit.

The reference leaves this to pygfx (blending, canvas encode); here it is one HBM-streaming kernel
(``svr_compose``) so that a frame can be shown or saved without leaving the GPU.
"""

from __future__ import annotations

import ctypes as C

from . import _native as N


def compose(volume, result, *, background=((0.0, 0.0, 0.0, 1.0), (0.0, 0.0, 0.0, 1.0)), depth_buffer=None,
            srgb: bool = True, out=None):
    """``result``: a :class:`RenderResult` of ``volume.render``.  ``background`` = (bottom, top) RGBA in linear
    light (a vertical gradient, like ``gfx.BackgroundMaterial(bottom, top)``).  ``depth_buffer``: optional f32
    CUDA tensor [h, w] holding the depth of what is already on the canvas (updated in place, "<" test).
    Returns a uint8 CUDA tensor [h, w, 4]."""
    import torch

    h, w = result.rgba.shape[:2]
    if out is None:
        out = torch.empty((h, w, 4), dtype=torch.uint8, device=result.rgba.device)
    if tuple(out.shape) != (h, w, 4) or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError("out must be a contiguous uint8 tensor [h, w, 4]")
    if depth_buffer is not None:
        if result.depth is None:
            raise ValueError("a depth test needs the render's depth plane")
        if tuple(depth_buffer.shape) != (h, w) or depth_buffer.dtype != torch.float32 or not depth_buffer.is_contiguous():
            raise ValueError("depth_buffer must be a contiguous float32 tensor [h, w]")
    q = N.ComposeParams()
    bottom, top = background
    q.bg_bottom[:] = [float(v) for v in bottom]
    q.bg_top[:] = [float(v) for v in top]
    q.srgb_encode = 1 if srgb else 0
    N.check(
        N.lib().svr_compose(
            volume._rings.handle, C.c_void_p(result.rgba.data_ptr()),
            C.c_void_p(result.depth.data_ptr()) if result.depth is not None else None,
            C.c_void_p(result.flags.data_ptr()) if result.flags is not None else None,
            w, h, C.byref(q), C.c_void_p(out.data_ptr()),
            C.c_void_p(depth_buffer.data_ptr()) if depth_buffer is not None else None,
            C.c_void_p(torch.cuda.current_stream(result.rgba.device).cuda_stream)),
        "svr_compose")
    return out
