#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: time the band set of every rank of a world of 1/2/4/8 ranks
(TiledFrame regions) and print the frame time a node of that many GPUs would reach (max over ranks,
without the gather), for interleaved row bands and for the most square grid of tiles (8 ranks: config 3's 2x4).
usage: exp_tiles.py [camera] [band_h] [variant]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402
from sub_volume_renderer_amd.distributed import TiledFrame  # noqa: E402

camname = sys.argv[1] if len(sys.argv) > 1 else "K1"
band_h = int(sys.argv[2]) if len(sys.argv) > 2 else 16
variant = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
scene = testing.build(bench.config2_spec(n, W, H, camname, pairs))
vol, cam = scene.volume, scene.camera
N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
for mode, tiling in (("full", "rows"), ("full", "grid"), ("lmip", "rows"), ("lmip", "grid")):
    vol.material.lmip_threshold = float("inf") if mode == "full" else 127.5
    base = None
    for world in (1, 2, 4, 8):
        times = []
        for rank in range(world):
            tf = TiledFrame(W, H, rank, world, band_h, tiling=tiling)
            r = vol.render(cam, W, H, region=tf.region)
            torch.cuda.synchronize()
            vol.prepare()
            cb, fb = vol.camera_block(cam), vol.frame_block(W, H, tf.region)
            ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
            ms = C.c_float(0)
            for iters in (10, 10):
                N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(ms)), "time")
            times.append(ms.value)
        t = max(times)
        base = base or t
        what = f"band_h={band_h}" if tiling == "rows" else "grid " + ("x".join(str(v) for v in TiledFrame(W, H, 0, world, tiling=tiling).grid[:2]) if world > 1 else "1x1")
        print(f"{camname} {mode} {what} world={world}: max {t:.4f} ms  min {min(times):.4f} ms  speedup {base/t:.2f}x", flush=True)
