"""A minimal chunked on-disk array (raw chunk directory) usable as ``backing_data``.

The reference accepts numpy, ``zarr.Array`` and tensorstore arrays (README.md:18,
``_wrapping_buffer.py:307-322``) and its pyramid scripts write zarr v3 stores with 16^3
chunks (scripts/create_mouse_multiscale.py:98-131).  Neither zarr nor tensorstore exists on
the target, so this is the smallest stand-in with the same *shape of access*: an object with
``shape``, ``dtype``, ``ndim``, ``chunks`` and ``__getitem__`` over slices that assembles the
requested block from per-chunk files.  Layout: ``<root>/meta.json`` + one uncompressed
C-order file per chunk, ``c.<i0>.<i1>.<i2>``; edge chunks are stored full-size (zero padded),
as zarr does.  Chunks are memory-mapped, so a ring-buffer load touches only the bytes it needs.
"""

from __future__ import annotations

import json
import os

import numpy as np


def write_chunk_dir(array, root: str, chunks) -> "ChunkDirArray":
    """Store ``array`` (anything sliceable with ``shape``/``dtype``) chunk by chunk under ``root``."""
    shape = tuple(int(v) for v in array.shape)
    chunks = tuple(int(v) for v in chunks)
    if len(chunks) != len(shape) or any(c <= 0 for c in chunks):
        raise ValueError("chunks must be positive and match the array rank")
    dtype = np.dtype(str(array.dtype).replace("torch.", ""))
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "meta.json"), "w") as f:
        json.dump({"shape": shape, "chunks": chunks, "dtype": dtype.str, "order": "C", "format": "svr-chunkdir-1"}, f)
    grid = [-(-s // c) for s, c in zip(shape, chunks)]
    for idx in np.ndindex(*grid):
        lo = [i * c for i, c in zip(idx, chunks)]
        hi = [min(l + c, s) for l, c, s in zip(lo, chunks, shape)]
        block = np.zeros(chunks, dtype)
        src = np.asarray(array[tuple(slice(l, h) for l, h in zip(lo, hi))])
        block[tuple(slice(0, h - l) for l, h in zip(lo, hi))] = src
        block.tofile(os.path.join(root, "c." + ".".join(str(i) for i in idx)))
    return ChunkDirArray(root)


class ChunkDirArray:
    """Read-only view of a chunk directory written by :func:`write_chunk_dir`."""

    def __init__(self, root: str, cache_chunks: int = 64):
        with open(os.path.join(root, "meta.json")) as f:
            meta = json.load(f)
        if meta.get("format") != "svr-chunkdir-1":
            raise ValueError(f"{root} is not an svr chunk directory")
        self.root = root
        self.shape = tuple(meta["shape"])
        self.chunks = tuple(meta["chunks"])
        self.dtype = np.dtype(meta["dtype"])
        self.ndim = len(self.shape)
        self._cache: dict = {}
        self._cache_chunks = cache_chunks

    def _chunk(self, idx):
        m = self._cache.get(idx)
        if m is None:
            path = os.path.join(self.root, "c." + ".".join(str(i) for i in idx))
            m = np.memmap(path, dtype=self.dtype, mode="r", shape=self.chunks)
            if len(self._cache) >= self._cache_chunks:
                self._cache.pop(next(iter(self._cache)))
            self._cache[idx] = m
        return m

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) != self.ndim or not all(isinstance(k, slice) for k in key):
            raise IndexError("ChunkDirArray supports one slice per axis")
        lo, hi = [], []
        for k, s in zip(key, self.shape):
            start, stop, step = k.indices(s)
            if step != 1:
                raise IndexError("only unit-stride slices are supported")
            lo.append(start)
            hi.append(max(start, stop))
        out = np.empty([h - l for l, h in zip(lo, hi)], self.dtype)
        if out.size == 0:
            return out
        first = [l // c for l, c in zip(lo, self.chunks)]
        last = [(h - 1) // c for h, c in zip(hi, self.chunks)]
        for idx in np.ndindex(*[b - a + 1 for a, b in zip(first, last)]):
            cidx = tuple(a + i for a, i in zip(first, idx))
            c_lo = [i * c for i, c in zip(cidx, self.chunks)]
            s_lo = [max(l, cl) for l, cl in zip(lo, c_lo)]
            s_hi = [min(h, cl + c) for h, cl, c in zip(hi, c_lo, self.chunks)]
            src = self._chunk(cidx)[tuple(slice(a - cl, b - cl) for a, b, cl in zip(s_lo, s_hi, c_lo))]
            out[tuple(slice(a - l, b - l) for a, b, l in zip(s_lo, s_hi, lo))] = src
        return out

    def __repr__(self):
        return f"ChunkDirArray({self.root!r}, shape={self.shape}, chunks={self.chunks}, dtype={self.dtype})"
