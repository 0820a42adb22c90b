#!/usr/bin/env python3
"""Frames in flight on ONE GPU for the band set of rank 0 of a world of N ranks (no gather): does the
tail of frame k overlap the head of frame k+1?  usage: exp_inflight.py [camera] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import synth, testing  # noqa: E402
from sub_volume_renderer_amd.distributed import TiledFrame  # noqa: E402

camname = sys.argv[1] if len(sys.argv) > 1 else "K1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
scene = testing.build(bench.config2_spec(n, W, H, camname, pairs))
vol, cam = scene.volume, scene.camera
for mode in ("full", "lmip"):
    vol.material.lmip_threshold = float("inf") if mode == "full" else 127.5
    for world in (1, 2, 4, 8):
        region = TiledFrame(W, H, 0, world, 16).region
        for F in (1, 2, 3, 4):
            vol._out_cache = {}
            outs = []
            for _ in range(F):
                outs.append(vol._outputs(region.out_h, region.out_w, False)); vol._out_cache = {}
            streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
            def run(k):
                for i in range(k):
                    with torch.cuda.stream(streams[i % F]):
                        vol.render(cam, W, H, region=region, out=outs[i % F])
            run(20); torch.cuda.synchronize()
            t = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / steps
            print(f"{camname} {mode} world={world} in_flight={F}: {dt*1e3:.4f} ms/frame", flush=True)
