#!/usr/bin/env python3
"""Critical path of the frame: kernel time of small regions (one wave tile, a few tiles) of the C2/K1 frame.
usage: exp_lone.py [camera] [variant]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402
from sub_volume_renderer_amd._wobject import FrameRegion  # noqa: E402

camname = sys.argv[1] if len(sys.argv) > 1 else "K1"
variant = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
scene = testing.build(bench.config2_spec(n, W, H, camname, pairs))
vol, cam = scene.volume, scene.camera
N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
for mode in ("full", "lmip"):
    vol.material.lmip_threshold = float("inf") if mode == "full" else 127.5
    for (x0, y0, w, h) in [(956, 536, 8, 8), (400, 300, 8, 8), (1500, 800, 8, 8), (928, 508, 64, 64), (704, 284, 512, 512),
                           (0, 0, 1920, 8), (0, 536, 1920, 8), (0, 536, 1920, 64), (0, 0, 1920, 1080)]:
        reg = FrameRegion.tile(x0, y0, w, h)
        tm = (C.c_uint64 * 16)(); dbg = (C.c_uint32 * 8)()
        N.lib().svr_debug_timers(vol._rings.handle, tm, 1); N.lib().svr_debug_counters(vol._rings.handle, dbg, 1)
        r = vol.render(cam, W, H, region=reg, count_steps=True)
        torch.cuda.synchronize()
        N.lib().svr_debug_timers(vol._rings.handle, tm, 1); N.lib().svr_debug_counters(vol._rings.handle, dbg, 1)
        tot = max(1, sum(tm))
        names = ["prol", "span", "gen", "slabdma", "wait", "brick", "direct", "epil", "slabred", "slabsalu", "skip", "t11", "t12", "t13", "t14", "t15"]
        tline = " ".join(f"{nm}={100*t/tot:.1f}%" for nm, t in zip(names, tm)) + f" | cycles/wave={tot/max(1,dbg[6]):.0f} census={list(dbg)[:7]}"
        steps = r.steps.to(torch.int64)
        r = vol.render(cam, W, H, region=reg)
        vol.prepare()
        cb, fb = vol.camera_block(cam), vol.frame_block(W, H, reg)
        ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
        ms = C.c_float(0)
        for iters in (10, 20):
            N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(ms)), "time")
        print(f"{camname} {mode} region {w}x{h}@({x0},{y0}): {ms.value*1000:8.1f} us  steps total {int(steps.sum())/1e6:8.3f} M  max/ray {int(steps.max())}\n      {tline}", flush=True)
