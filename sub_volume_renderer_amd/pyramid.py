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


def write_multiscale_store(root: str, source, levels: int, mode: str, chunks=(16, 16, 16), shards=(64, 64, 64),
                           compressor: str | None = "zstd", fill_value=0, device: int = 0, names=None):
    """The reference's offline builders (scripts/create_mouse_multiscale.py:98-160, create_platynereis_multiscale.py:
    136-200) in one pass: write ``source`` — anything sliceable with ``shape`` / ``dtype`` (numpy, ``zarr3.ZarrV3Array``,
    a tensorstore-shaped object) — as ``scale0`` of a zarr v3 group at ``root`` and its 2x pooled levels as ``scale1`` ..
    (``mode='mean'``: density, uint8 or float32; ``mode='max'``: labels, uint32), 16^3 chunks in 64^3 shards like theirs.

    The volume streams through the device slab by slab (slabs of ``shards[0] * 2^(levels-1)`` planes along the first
    axis, so every level receives whole shard rows): the pooling runs on the GPU (``svr_pool2x``), the encoding in
    ``csrc/host_codecs.c``; nothing larger than a slab of each level is resident.  Extents must be divisible by
    ``2^(levels-1)`` (the reference asserts powers of two).  Returns the opened arrays, finest first."""
    import numpy as np
    import torch

    from . import zarr3

    shape = tuple(int(v) for v in source.shape)
    if len(shape) != 3:
        raise ValueError("write_multiscale_store handles 3-D volumes")
    if any(s % (1 << (levels - 1)) for s in shape):
        raise ValueError(f"every extent of {shape} must be divisible by 2^(levels-1) = {1 << (levels - 1)}")
    dtype = np.dtype(source.dtype)
    if mode == "max" and dtype != np.uint32:
        raise TypeError("mode='max' (labels) needs uint32 data")
    if mode == "mean" and dtype not in (np.dtype(np.uint8), np.dtype(np.float32)):
        raise TypeError("mode='mean' (density) needs uint8 or float32 data")
    zarr3.create_group(root)
    names = names or [f"scale{k}" for k in range(levels)]
    arrays = [zarr3.create_array(f"{root}/{names[k]}", tuple(s >> k for s in shape), dtype, chunks, shards, compressor, fill_value)
              for k in range(levels)]
    dev = torch.device("cuda", device)
    slab0 = int(shards[0]) << (levels - 1)
    for z0 in range(0, shape[0], slab0):
        z1 = min(shape[0], z0 + slab0)
        block = np.ascontiguousarray(np.asarray(source[z0:z1, :, :]), dtype)
        t = torch.from_numpy(block.view(np.int32) if dtype == np.uint32 else block).to(dev)
        for k in range(levels):
            if k:
                t = pool2x(t, mode)
            host = t.cpu().numpy()
            host = host.view(np.uint32) if dtype == np.uint32 else host
            lo = z0 >> k
            for i0 in range(0, host.shape[0], shards[0]):                   # whole shard rows of this level
                for i1 in range(0, host.shape[1], shards[1]):
                    for i2 in range(0, host.shape[2], shards[2]):
                        zarr3.write_block(arrays[k], ((lo + i0) // shards[0], i1 // shards[1], i2 // shards[2]),
                                          host[i0:i0 + shards[0], i1:i1 + shards[1], i2:i2 + shards[2]])
        del t
    return [zarr3.open_zarr(a.path) for a in arrays]
