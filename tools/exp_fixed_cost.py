#!/usr/bin/env python3
"""Fixed cost of a frame: the march kernel with the march loop skipped (variant bit 12: prologue + epilogue only,
every fragment a MISS) and with threshold 0 (every ray hits at its first sample: prologue + two batches + the full
hit epilogue: label fetch, colour, fog, depth), against the full and LMIP frames.  usage: exp_fixed_cost.py [camera]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402

camname = sys.argv[1] if len(sys.argv) > 1 else "K1"
n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
scene = testing.build(bench.config2_spec(n, W, H, camname, pairs))
vol, cam = scene.volume, scene.camera
r = vol.render(cam, W, H)


def ms(variant, threshold):
    N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
    vol.material.lmip_threshold = threshold
    vol.prepare()
    cb, fb = vol.camera_block(cam), vol.frame_block(W, H, None)
    ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
    out = C.c_float(0)
    for iters in (10, 20):
        N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(out)), "time")
    return out.value * 1e3


print(f"{camname}: march loop skipped (prologue + MISS epilogue): {ms(1 << 12, float('inf')):7.1f} us")
print(f"{camname}: threshold 0 (prologue + 2 batches + HIT epilogue):  {ms(0, 0.0):7.1f} us")
print(f"{camname}: LMIP frame (threshold 127.5):                      {ms(0, 127.5):7.1f} us")
print(f"{camname}: full frame (threshold inf):                        {ms(0, float('inf')):7.1f} us")
