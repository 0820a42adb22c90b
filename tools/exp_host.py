#!/usr/bin/env python3
"""Host-side cost per frame: render 1/8 of the frame (what one rank of 8 does) in a tight loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sub_volume_renderer_amd import FrameRegion, synth, testing
dev = torch.device("cuda", 0)
n, W, H = 1024, 1920, 1080
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
spec = bench.config2_spec(n, W, H, "K1", pairs)
scene = testing.build(spec); vol, cam = scene.volume, scene.camera
vol.material.lmip_threshold = float("inf")
for world in (1, 2, 4, 8):
    reg = FrameRegion.stripes(W, H, 0, world, 16) if world > 1 else FrameRegion.full(W, H)
    out = vol._outputs(reg.out_h, reg.out_w, False)
    for _ in range(5): vol.render(cam, W, H, region=reg, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter(); K = 200
    for _ in range(K): vol.render(cam, W, H, region=reg, out=out)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"world={world}: {dt/K*1e3:.3f} ms per frame (host issue {t_issue/K*1e6:.1f} us per call) -> ideal speedup vs full {0:.0f}", flush=True)
