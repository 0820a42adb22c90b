"""GPU LOD pyramid builder.

The reference prepares its multi-scale inputs offline with numpy: 2x mean-pool for the density
(scripts/create_mouse_multiscale.py:23-54) and 2x max-pool for the labels
(scripts/create_platynereis_multiscale.py:86-134).  ``build_pyramid`` does the same on the device with
the ``svr_pool2x`` kernels; sources and results are torch CUDA tensors in numpy axis order."""

from __future__ import annotations

import ctypes as C

from . import _native as N


def pool2x(t, mode: str):
    """One 2x2x2 pooling step of a contiguous 3-D CUDA tensor: ``mode='mean'`` (uint8: floor of the
    mean; float32) or ``mode='max'`` (int32 holding uint32 label bit patterns, or uint8)."""
    import torch

    if not (t.is_cuda and t.dim() == 3 and t.is_contiguous()):
        raise ValueError("pool2x needs a contiguous 3-D CUDA tensor")
    if any(s % 2 or s < 2 for s in t.shape):
        raise ValueError("every extent must be even and >= 2")
    code = {("mean", torch.uint8): (N.SVR_U8, 0), ("mean", torch.float32): (N.SVR_F32, 0),
            ("max", torch.int32): (2, 1)}.get((mode, t.dtype))
    if code is None:
        raise TypeError(f"unsupported combination mode={mode!r} dtype={t.dtype}")
    out = torch.empty(tuple(s // 2 for s in t.shape), dtype=t.dtype, device=t.device)
    dims = N.i3(tuple(t.shape)[::-1])
    N.check(N.lib().svr_pool2x(t.device.index, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), dims,
                               code[0], code[1], C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)),
            "svr_pool2x")
    return out


def build_pyramid(density, labels, levels: int):
    """``[(density_0, labels_0), ..., (density_{levels-1}, labels_{levels-1})]`` ready for ``SubVolume``."""
    pairs = [(density, labels)]
    for _ in range(1, levels):
        density, labels = pool2x(density, "mean"), pool2x(labels, "max")
        pairs.append((density, labels))
    return pairs
