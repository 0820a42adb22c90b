#!/usr/bin/env python3
"""Experiment: march throughput vs viewing direction (is the texel-fetch path line-bound?)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
torch.cuda.synchronize()
spec = bench.config2_spec(n, W, H, "K1", pairs)
scene = testing.build(spec)
vol = scene.volume
N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
c = (n - 1) / 2.0
dirs = {"K1(-.80,.36,.48)": (-0.80, 0.36, 0.48), "-x": (-1, 0.02, 0.03), "-y": (0.02, -1, 0.03), "-z": (0.02, 0.03, -1),
        "diag": (-1, -1, -1)}
for mode in ("full", "lmip"):
    vol.material.lmip_threshold = float("inf") if mode == "full" else 127.5
    for name, d in dirs.items():
        d = np.array(d, float); d /= np.linalg.norm(d)
        spec.cam_position = tuple(np.array([c, c, c]) + 1.6 * n * d)
        spec.cam_target = (c, c, c)
        cam = spec.camera()
        r = vol.render(cam, W, H, count_steps=True)
        torch.cuda.synchronize()
        steps = int(r.steps.to(torch.int64).sum().item())
        vol.prepare()
        cb, fb = vol.camera_block(cam), vol.frame_block(W, H, None)
        ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
        ms = C.c_float(0)
        N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), 5, C.byref(ms)), "time")
        print(f"{mode:5s} {name:18s} steps={steps/1e6:8.1f}M  {ms.value:7.3f} ms  {steps/ms.value/1e6:7.1f} Gsteps/s  {4*steps/ms.value/1e6/8000*100:5.1f}% of 8TB/s", flush=True)
