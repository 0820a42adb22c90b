#!/usr/bin/env python3
"""Experiment: kernel variants (tile shape / skew) on C2 for a given camera."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
variants = [int(v, 0) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
cams = sys.argv[3].split(",") if len(sys.argv) > 3 else ["K1"]
W, H = 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
torch.cuda.synchronize()
c = (n - 1) / 2.0
dirs = {"-x": (-1, 0.02, 0.03), "-y": (0.02, -1, 0.03), "-z": (0.02, 0.03, -1), "diag": (-1, -1, -1)}
for camname in cams:
    spec = bench.config2_spec(n, W, H, camname if camname in ("K1", "K2") else "K1", pairs)
    if camname in dirs:
        d = np.array(dirs[camname], float); d /= np.linalg.norm(d)
        spec.cam_position = tuple(np.array([c, c, c]) + 1.6 * n * d); spec.cam_target = (c, c, c)
    scene = testing.build(spec)
    vol, cam = scene.volume, scene.camera
    ref = {}
    for mode in ("full", "lmip"):
        vol.material.lmip_threshold = float("inf") if mode == "full" else 127.5
        for v in variants:
            N.check(N.lib().svr_set_variant(vol._rings.handle, v), "variant")
            dbg = (C.c_uint32 * 8)()
            N.lib().svr_debug_counters(vol._rings.handle, dbg, 1)
            r = vol.render(cam, W, H, count_steps=True)
            torch.cuda.synchronize()
            N.lib().svr_debug_counters(vol._rings.handle, dbg, 1)
            census = list(dbg)
            steps = int(r.steps.to(torch.int64).sum().item())
            sig = (steps, int(r.label.to(torch.int64).sum().item()), float(r.rgba.double().sum().item()))
            ref.setdefault(mode, sig)
            same = "same" if sig == ref[mode] else "DIFFERENT!"
            vol.prepare()
            cb, fb = vol.camera_block(cam), vol.frame_block(W, H, None)
            ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
            ms = C.c_float(0)
            for iters in (10, 10):      # first pass = warm-up (clocks, caches)
                N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(ms)), "time")
            print(f"{camname:5s} {mode:5s} variant={v:#06x} steps={steps/1e6:8.1f}M {ms.value:7.3f} ms {steps/ms.value/1e6:7.1f} Gsteps/s "
                  f"{4*steps/ms.value/1e6/8000*100:5.1f}% {same} census[gen,dir,brick,slabs,runs,zero,waves,skipped]={census[:8]}", flush=True)
    del scene, vol
