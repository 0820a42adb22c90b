#!/usr/bin/env python3
"""Config-4 style fly-through: 240 frames, 2 voxels per frame along the view direction, calling
center_on_position every frame; host-resident (or lazily generated) backing volume.
Compares the reference's blocking streaming with the asynchronous publish protocol.
usage: exp_fly.py [n=1024] [frames=240] [lazy=0|1]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import synth, testing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 240
lazy = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, H = 1920, 1080
dev = torch.device("cuda", 0)
if lazy:
    pairs = [(synth.LazyLod(n, k, False), synth.LazyLod(n, k, True)) for k in range(3)]
else:
    pd = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
    pairs = [(d.cpu().numpy(), l.cpu().numpy().view(np.uint32)) for d, l in pd]
    del pd
out = {"n": n, "frames": frames, "lazy": lazy}
for mode in ("blocking", "async"):
    spec = bench.config2_spec(n, W, H, "K2", pairs)
    if lazy:
        spec.ring_storage = "native"
    t = time.perf_counter()
    scene = testing.build(spec)
    scene.volume.synchronize()
    out[f"{mode}_initial_fill_s"] = round(time.perf_counter() - t, 3)
    vol = scene.volume
    eye = np.array(spec.cam_position); d = np.array(spec.cam_target) - eye; d /= np.linalg.norm(d)
    times, steps = [], 0
    for k in range(frames):
        p = eye + d * 2.0 * (k + 1)
        spec.cam_position, spec.cam_target = tuple(p), tuple(p + d)
        cam = spec.camera()
        t = time.perf_counter()
        res = vol.render(cam, W, H)
        vol.center_on_position(tuple(p), asynchronous=(mode == "async"))
        torch.cuda.synchronize()
        if mode == "blocking":
            vol.synchronize()
        times.append(time.perf_counter() - t)
    vol.poll_uploads(wait=True)
    ts = np.array(times) * 1e3
    out[mode] = {"median_ms": round(float(np.median(ts)), 3), "p99_ms": round(float(np.percentile(ts, 99)), 3),
                 "max_ms": round(float(ts.max()), 3), "fps_mean": round(float(frames / (ts.sum() / 1e3)), 1),
                 "frames_over_5ms": int((ts > 5).sum())}
    del scene, vol
print(json.dumps(out))
