#!/usr/bin/env python3
"""The streaming kernels beside the march — ring scatter (both paths), macro-cell maxima, ring read-back, pyramid
pooling, display-side compose, un-tile — timed with HIP events on their own, against their algorithmic bytes
(read + written) and the HBM peak.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel summary
(profiles/r02/kernel_stats_streaming_kernels.csv).  usage: exp_kernels.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sub_volume_renderer_amd import Roi, WrappingBuffer, _native as N  # noqa: E402
from sub_volume_renderer_amd.compose import compose  # noqa: E402
from sub_volume_renderer_amd.pyramid import pool2x  # noqa: E402

HBM = 8000.0
dev = torch.device("cuda", 0)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def report(name, ms, nbytes):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(f"{name:58s} {ms:8.3f} ms  {nbytes / 1e6:9.1f} MB  {gbs:7.0f} GB/s  {gbs / HBM:5.2f} of HBM peak", flush=True)


# ---- ring scatter from a device-resident source (no PCIe): one LOD-0 slab of config 2 and the whole ring
n = 1024
d = torch.randint(0, 255, (n // 2, n // 2, 528), dtype=torch.uint8, device=dev)
l = torch.randint(0, 2 ** 31 - 1, (n // 2, n // 2, 528), dtype=torch.int32, device=dev)
# (EXP_NO_TWIN=1: rows only, without the micro-block copy every upload of the finest ring writes too)
from sub_volume_renderer_amd._wrapping_buffer import DeviceRings  # noqa: E402
rings = DeviceRings([(512, 512, 528)], density_storage="uint8", blocked_twin=os.environ.get("EXP_NO_TWIN") is None)
buf = WrappingBuffer(d, l, (32, 32, 11), (16, 16, 48), _rings=rings, _lod=0)
print("micro-block copy of the ring:", rings.blocked_twin, flush=True)
lib, h = N.lib(), buf.rings.handle


def upload(roi):
    buf._current_logical_roi_in_chunks = None
    buf._current_logical_roi_in_pixels = None
    buf.load_logical_roi(roi)
    N.check(lib.svr_sync_uploads(h), "sync")


for name, roi in (("whole ring 512x512x528", Roi((0, 0, 0), (512, 512, 528))), ("one chunk slab 16x512x528", Roi((0, 0, 0), (16, 512, 528)))):
    vox = int(np.prod(roi.shape))
    for label, env in (("scatter_rows16 (16-B groups)", None), ("scatter_kernel (one voxel per thread)", "1")):
        if env:
            # the A/B switch is read once per process: run this script with SVR_SCATTER_GENERAL=1 for the general path
            if not os.environ.get("SVR_SCATTER_GENERAL"):
                continue
        elif os.environ.get("SVR_SCATTER_GENERAL"):
            continue
        t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            upload(roi)
        ms = (time.perf_counter() - t) / reps * 1e3
        # scatter: 5 B read + 5 B written per voxel; cell maxima: + 1 B read per voxel (+ the tiny grids)
        report(f"ring upload from HBM, {name}: {label} + cell maxima", ms, vox * 11)

# ---- read-back gather (texture.data)
t = timed(lambda: buf.read_ring(Roi((0, 0, 0), (64, 512, 528))), 2)
report("read-back 64x512x528 (gather kernel + D2H copies, host wall)", t, 64 * 512 * 528 * 5)

# ---- pyramid pooling
for dtype, mode, es in ((torch.uint8, "mean", 1), (torch.int32, "max", 4), (torch.float32, "mean", 4)):
    src = torch.randint(0, 200, (n, n, n), device=dev).to(dtype)
    ms = timed(lambda: pool2x(src, mode))
    report(f"pool2x {mode} {str(dtype).replace('torch.', '')} 1024^3 -> 512^3", ms, n ** 3 * es * 9 // 8)
    del src

# ---- compose (blend + depth test + sRGB8): 21 B read + 4 B written + 8 B depth buffer read-modify-write per pixel
from types import SimpleNamespace  # noqa: E402

from sub_volume_renderer_amd._wobject import RenderResult  # noqa: E402

W, H = 1920, 1080
res = RenderResult(rgba=torch.rand((H, W, 4), device=dev), depth=torch.rand((H, W), device=dev) * 0.5,
                   label=None, flags=torch.full((H, W), 2, dtype=torch.uint8, device=dev), steps=None)
zbuf = torch.ones((H, W), device=dev)
out = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
holder = SimpleNamespace(_rings=buf.rings)
ms = timed(lambda: compose(holder, res, depth_buffer=zbuf, out=out), 50)
report("compose 1920x1080 (blend + depth test + sRGB8)", ms, W * H * 33)

# ---- un-tile of gathered planes (8 ranks): row bands and the 2 x 4 grid
g = torch.rand((8, 144, W, 4), device=dev)
frame = torch.empty((H, W, 4), device=dev)
s0 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
ms = timed(lambda: N.check(lib.svr_untile_stripes(h, C.c_void_p(g.data_ptr()), C.c_void_p(frame.data_ptr()), W, H, 16, 8, 144, 16, s0), "untile"), 50)
report("untile_stripes RGBA f32, 8 ranks x 16-row bands", ms, W * H * 32)
g2 = torch.rand((8, 270, 960, 4), device=dev)
ms = timed(lambda: N.check(lib.svr_untile_grid(h, C.c_void_p(g2.data_ptr()), C.c_void_p(frame.data_ptr()), W, H, 960, 270, 2, 4, 16, s0), "untile"), 50)
report("untile_grid RGBA f32, 2 x 4 tiles of 960 x 270", ms, W * H * 32)
