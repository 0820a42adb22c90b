#!/usr/bin/env python3
"""What the hit epilogue's label fetch costs: C2 kernel time per camera and mode with the label rings and without
(a label-less volume: every hit gets label 0, nothing is fetched)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402


def time_ms(vol, cam, W, H, iters=20):
    vol.prepare()
    cb, fb = vol.camera_block(cam), vol.frame_block(W, H, None)
    r = vol.render(cam, W, H)
    ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
    ms = C.c_float(0)
    for it in (5, iters):
        N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), it, C.byref(ms)), "time")
    return ms.value


n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
for camname in ("K1", "K2"):
    for labelled in (True, False):
        spec = bench.config2_spec(n, W, H, camname, pairs if labelled else [(d, None) for d, _ in pairs])
        scene = testing.build(spec)
        vol, cam = scene.volume, scene.camera
        for mode, thr in (("lmip", 127.5), ("full", float("inf")), ("all-hit", 1.0)):
            vol.material.lmip_threshold = thr
            r = vol.render(cam, W, H)
            torch.cuda.synchronize()
            hits = int((r.flags == 2).sum())
            print(f"{camname} {'labels' if labelled else 'no labels'} {mode}: {time_ms(vol, cam, W, H):.3f} ms, {hits} hit rays", flush=True)
        del scene, vol
