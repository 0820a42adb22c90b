#!/usr/bin/env python3
"""Minimal driver for rocprofv3 passes: C2 scene, N renders of ONE march mode (so every
march dispatch in the trace is the same workload).  usage: prof_driver.py [full|lmip] [n] [iters] [variant] [camera]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "full"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
variant = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
camera = sys.argv[5] if len(sys.argv) > 5 else "K1"
W, H = 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
torch.cuda.synchronize()
import numpy as np  # noqa: E402

spec = bench.config2_spec(n, W, H, camera if camera in ("K1", "K2") else "K1", pairs)
dirs = {"-x": (-1, 0.02, 0.03), "-y": (0.02, -1, 0.03), "-z": (0.02, 0.03, -1), "diag": (-1, -1, -1)}
if camera in dirs:
    c = (n - 1) / 2.0
    d = np.array(dirs[camera], float); d /= np.linalg.norm(d)
    spec.cam_position = tuple(np.array([c, c, c]) + 1.6 * n * d); spec.cam_target = (c, c, c)
scene = testing.build(spec)
vol = scene.volume
N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
vol.material.lmip_threshold = float("inf") if mode == "full" else 127.5
for _ in range(iters):
    vol.render(scene.camera, W, H)
torch.cuda.synchronize()
print("done", mode, n, iters, variant, camera)
