#!/usr/bin/env python3
"""The reference's multi-scale demo (scripts/multi_scale.py) on this package, off screen: a lattice of small
bright boxes at three resolutions, coloured by the level each hit was read from (the label arrays hold 0 / 1 / 2),
seen from the demo's camera pose.  Where the reference opens a pygfx canvas and redraws forever, this renders one
frame per render mode, composes it over the demo's grey background and writes PNG files.

usage: python examples/multi_scale.py [out_dir]        (needs an MI355X and the built libsvr_hip.so)
"""
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sub_volume_renderer_amd import PerspectiveCamera, SubVolume, SubVolumeMaterial, compose  # noqa: E402


def write_png(path, rgba8):
    """uint8 [h, w, 4] -> RGBA PNG (zlib only)."""
    h, w = rgba8.shape[:2]
    raw = b"".join(b"\x00" + rgba8[y].tobytes() for y in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def lattice(chunk, box, tiles=16):
    """One bright box in the corner of every chunk."""
    c = np.zeros(chunk, np.float32)
    c[: box[0], : box[1], : box[2]] = 1.0
    return np.tile(c, (tiles, tiles, tiles))


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
    os.makedirs(out_dir, exist_ok=True)
    width = height = 480
    chunks = [(16, 16, 48), (8, 8, 48), (4, 4, 48)]
    levels = [lattice(chunks[0], (4, 4, 4)), lattice(chunks[1], (2, 2, 4)), lattice(chunks[2], (1, 1, 4))]
    pairs = [(d, np.full(d.shape, k, np.uint8)) for k, d in enumerate(levels)]       # label = level of the hit

    material = SubVolumeMaterial(lmip_threshold=0.5, fog_density=0.01,
                                 colors=[(0.0, 1.0, 1.0), (0.33, 1.0, 1.0), (0.66, 1.0, 1.0)])
    volume = SubVolume(material, data_segmentation_pairs=pairs, chunk_shape_in_pixels=chunks,
                       buffer_shape_in_chunks=[(2, 2, 2), (4, 4, 4), (8, 8, 8)])
    volume.world.position = 0, 0, 0

    camera = PerspectiveCamera(fov=45, aspect=width / height, depth_range=(0.1, 2000.0))
    camera.world.position = -19.81, 7.5, 7.5
    camera.look_at((-1, 0, 0))

    volume.center_on_position(camera.world.position)          # the windows of all three levels follow the camera
    grey = lambda v: (v / 255.0,) * 3 + (1.0,)                # noqa: E731
    background = (grey(100), grey(168))                       # bottom, top of the demo's gradient
    for mode in ("lmip", "mip", "weighted_average"):
        material.render_mode = mode
        frame = volume.render(camera, width, height)
        image = compose(volume, frame, background=background).cpu().numpy()
        hits = int((frame.flags == 2).sum())
        by_level = np.bincount(frame.label[frame.flags == 2].cpu().numpy().astype(np.int64), minlength=3)
        path = os.path.join(out_dir, f"multi_scale_{mode}.png")
        write_png(path, image)
        print(f"{mode}: {hits} of {width * height} pixels hit; hits read from level 0 / 1 / 2: {by_level.tolist()} -> {path}")


if __name__ == "__main__":
    main()
