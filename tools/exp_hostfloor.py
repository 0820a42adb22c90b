#!/usr/bin/env python3
"""Host-side cost per frame of the bench loop (render call + async RCCL gather + finish/un-tile) when the GPU
work is negligible (threshold 0: every ray stops after a few samples).  One-rank RCCL group."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import synth, testing  # noqa: E402
from sub_volume_renderer_amd.distributed import TiledFrame  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29551")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n, W, H = 256, 1920, 1080
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=64) for k in range(3)]
scene = testing.build(bench.config2_spec(n, W, H, "K1", pairs))
vol, cam = scene.volume, scene.camera
vol.material.lmip_threshold = 0.0
for world, collective in ((8, True), (8, False), (1, False)):
    tiled = TiledFrame(W, H, 0, 1, 16, force_collective=collective)
    if collective and os.environ.get("SVR_HOSTFLOOR_TRANSPORT", "svr") == "svr":
        print("init_comm:", tiled.init_comm(vol), tiled.transport, flush=True)
    region = TiledFrame(W, H, 0, world, 16).region if not collective else tiled.region
    if collective:                                   # bands of an 8-rank tiling would need 8 ranks; use this rank's full band set
        region = tiled.region
    F = 4
    outs = []
    for _ in range(F):
        vol._out_cache = {}
        outs.append(vol._outputs(region.out_h, region.out_w, False))
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    def loop(k):
        for i in range(k):
            s = i % F
            with torch.cuda.stream(streams[s]):
                if collective:
                    tiled.finish(s)
                res = vol.render(cam, W, H, region=region, out=outs[s])
                if collective:
                    tiled.gather_async(res.rgba, s, volume=vol)
        if collective:
            for s in range(F):
                with torch.cuda.stream(streams[s]):
                    tiled.finish(s)
    loop(40); torch.cuda.synchronize()
    t = time.perf_counter(); loop(1000); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 1000
    print(f"rows={region.out_h} collective={collective}: {dt*1e6:.1f} us per frame (GPU work negligible)", flush=True)
dist.destroy_process_group()
