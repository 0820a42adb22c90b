"""Display-side compose, restated in numpy float32.

TEST INFRASTRUCTURE ONLY.  What a canvas shows for the (SubVolume, SubVolumeMaterial) pair alone: the
fragment outputs (fs_main.wgsl:86-87) blended over a vertical-gradient background like the one the
reference's tests add (tests/conftest.py:17-22), with pygfx's default alpha blending (src_alpha,
one_minus_src_alpha), depth_compare "<" and an sRGB 8-bit canvas.  Blending and the final encode live in
pygfx (third party, absent): restated from its published behaviour — **parity unpinned**.
"""

from __future__ import annotations

import numpy as np

f32 = np.float32


def srgb_encode(c):
    c = np.clip(c, f32(0), f32(1)).astype(f32)
    hi = f32(1.055) * np.power(c, f32(1.0 / 2.4), dtype=f32) - f32(0.055)
    return np.where(c <= f32(0.0031308), f32(12.92) * c, hi).astype(f32)


def compose(rgba, depth=None, flags=None, bg_bottom=(0, 0, 0, 1), bg_top=(0, 0, 0, 1), zbuf=None, srgb=True):
    """Returns (uint8 [h, w, 4], zbuf or None)."""
    rgba = np.asarray(rgba, f32)
    h, w = rgba.shape[:2]
    t = ((np.arange(h, dtype=f32) + f32(0.5)) / f32(h))[:, None, None]
    c = (np.asarray(bg_top, f32)[None, None, :] * (f32(1) - t) + np.asarray(bg_bottom, f32)[None, None, :] * t)
    c = np.broadcast_to(c, (h, w, 4)).astype(f32).copy()
    draw = np.ones((h, w), bool) if flags is None else (np.asarray(flags) != 0)
    if zbuf is not None:
        zbuf = np.array(zbuf, f32)
        draw &= np.asarray(depth, f32) < zbuf
    a = rgba[..., 3:4]
    over = np.concatenate([rgba[..., :3] * a + c[..., :3] * (f32(1) - a), a + c[..., 3:4] * (f32(1) - a)], axis=-1)
    c = np.where(draw[..., None], over, c).astype(f32)
    if zbuf is not None:
        zbuf = np.where(draw, np.asarray(depth, f32), zbuf).astype(f32)
    if srgb:
        c[..., :3] = srgb_encode(c[..., :3])
    q = (np.clip(c, f32(0), f32(1)) * f32(255.0) + f32(0.5)).astype(np.uint8)
    return q, zbuf
