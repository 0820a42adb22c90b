#!/usr/bin/env python3
"""The reference's zarr-backed demo (scripts/mouse.py, with the stores its builders write: scripts/create_*_multiscale.py)
end to end on this package, off screen and without zarr-python:

1. build the multi-scale stores from a level-0 volume — `write_multiscale_store`: GPU pooling, zarr v3 groups `raw.zarr` /
   `labels.zarr` with arrays `scale0 ..`, 16^3 chunks in 64^3 shards, zstd;
2. open them with `zarr3.open_group` and hand the arrays to `SubVolume` exactly as mouse.py does (same material, five
   levels, 4 x 4 x 4 rings of 16^3 chunks, world.scale_z = 6);
3. fly the camera through the volume, `center_on_position(asynchronous=True)` after every frame as its draw callback
   does, and write a PNG every few frames.

The level-0 volume is this package's synthetic one (the mouse data is not available offline); pass a directory that
already holds `raw.zarr` / `labels.zarr` as the second argument to skip step 1.

usage: python examples/zarr_multiscale.py [out_dir] [store_dir]        (needs an MI355X and the built libraries)
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from multi_scale import write_png  # noqa: E402
from sub_volume_renderer_amd import PerspectiveCamera, SubVolume, SubVolumeMaterial, compose, synth, zarr3  # noqa: E402
from sub_volume_renderer_amd.pyramid import write_multiscale_store  # noqa: E402

LEVELS = 5


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "."
    os.makedirs(out_dir, exist_ok=True)
    store = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="svr_example_store_")
    if not os.path.exists(os.path.join(store, "raw.zarr", "zarr.json")):
        n = 512
        t = time.time()
        density, labels = synth.volume(n, 0, n_labels=5)
        write_multiscale_store(os.path.join(store, "raw.zarr"), density, LEVELS, "mean")
        write_multiscale_store(os.path.join(store, "labels.zarr"), labels, LEVELS, "max")
        print(f"stores written under {store} in {time.time() - t:.1f} s")

    raw_group = zarr3.open_group(os.path.join(store, "raw.zarr"))                 # mouse.py:32-33
    labels_group = zarr3.open_group(os.path.join(store, "labels.zarr"))
    pairs = [(raw_group[f"scale{k}"], labels_group[f"scale{k}"]) for k in range(LEVELS)]
    volume = SubVolume(
        SubVolumeMaterial(lmip_threshold=150, clim=(47, 135), lmip_fall_off=0.5, lmip_max_samples=25, fog_density=0.1,
                          fog_color=(0, 0, 0),
                          colors=[(0.0, 1.0, 1.0), (0.20, 1.0, 1.0), (0.40, 1.0, 1.0), (0.60, 1.0, 1.0), (0.86, 1.0, 1.0)]),
        data_segmentation_pairs=pairs,
        chunk_shape_in_pixels=[(16, 16, 16)] * LEVELS,
        buffer_shape_in_chunks=[(4, 4, 4)] * LEVELS,
    )
    volume.world.position = 0, 0, 0
    volume.world.scale_z = 6                                                       # mouse.py:90-91

    width, height = 960, 540
    camera = PerspectiveCamera(fov=45, aspect=width / height, depth_range=(0.5, 20000.0))
    extent = np.array(raw_group["scale0"].shape[::-1], float) * np.array([1.0, 1.0, 6.0])    # world size (x, y, z)
    eye = extent * np.array([0.15, 0.2, 0.1])
    direction = extent * np.array([0.85, 0.8, 0.9]) - eye
    direction /= np.linalg.norm(direction)
    grey = lambda v: (v / 255.0,) * 3 + (1.0,)  # noqa: E731
    volume.center_on_position(tuple(eye))                                          # blocking first fill
    frames, t0 = 120, time.time()
    for k in range(frames):
        position = eye + direction * 6.0 * k
        camera.world.position = tuple(position)
        camera.look_at(tuple(position + direction))
        frame = volume.render(camera, width, height)                               # renderer.render(scene, camera)
        volume.center_on_position(camera.world.position, asynchronous=True)        # mouse.py:98-100
        if k % 30 == 0:
            image = compose(volume, frame, background=(grey(100), grey(168))).cpu().numpy()
            path = os.path.join(out_dir, f"zarr_multiscale_{k:03d}.png")
            write_png(path, image)
            levels_hit = np.bincount(np.minimum(frame.label[frame.flags == 2].cpu().numpy().astype(np.int64), 4), minlength=5)
            print(f"frame {k}: {int((frame.flags == 2).sum())} pixels hit (labels 0..4: {levels_hit.tolist()}) -> {path}")
    volume.poll_uploads(wait=True)
    reads = sum(a.read_bytes for pair in pairs for a in pair)
    seconds = sum(a.read_seconds for pair in pairs for a in pair)
    print(f"{frames} frames in {time.time() - t0:.2f} s; {reads / 1e6:.0f} MB decoded from the stores in {seconds:.2f} s of reads")
    volume.close()


if __name__ == "__main__":
    main()
