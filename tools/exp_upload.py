#!/usr/bin/env python3
"""Experiment: host -> ring upload throughput (PCIe-inclusive) and per-frame streaming cost."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import synth, testing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W, H = 1920, 1080
t = time.time()
dev = torch.device("cuda", 0)
pairs_dev = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
pairs = [(d.cpu().numpy(), l.cpu().numpy().view(np.uint32)) for d, l in pairs_dev]      # host-resident backing arrays
del pairs_dev
print(f"host volume ready in {time.time()-t:.1f}s", flush=True)
spec = bench.config2_spec(n, W, H, "K2", pairs)
t = time.perf_counter()
scene = testing.build(spec)
scene.volume.synchronize()
dt = time.perf_counter() - t
vox = sum(int(np.prod(b._current_logical_roi_in_pixels.intersect(type(b._current_logical_roi_in_pixels)((0,0,0), b.backing_data.shape)).shape)) for b in scene.volume.wrapping_buffers)
print(f"initial ring fill: {vox/1e6:.1f} Mvoxels ({vox*5/1e9:.2f} GB u8+u32) in {dt*1e3:.1f} ms -> {vox*5/dt/1e9:.2f} GB/s host->HBM", flush=True)
# fly-through: 2 voxels per frame along the view direction
vol, cam = scene.volume, scene.camera
eye = np.array(spec.cam_position); d = np.array(spec.cam_target) - eye; d /= np.linalg.norm(d)
times = []
for k in range(60):
    p = eye + d * 2.0 * (k + 1)
    t = time.perf_counter()
    vol.center_on_position(tuple(p))
    vol.synchronize()
    times.append(time.perf_counter() - t)
times = np.array(times) * 1e3
print(f"center_on_position per frame: median {np.median(times):.2f} ms, max {times.max():.2f} ms, frames with uploads (>1ms): {(times>1).sum()}/60")
