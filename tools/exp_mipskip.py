#!/usr/bin/env python3
"""MIP mode (and the weighted-average mode, which has its own kernel): batches skipped / marched (census of the instrumented kernel) and kernel time with and without skipping."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402


def census(vol):
    c = (C.c_uint32 * 8)()
    N.check(N.lib().svr_debug_counters(vol._rings.handle, c, 1), "svr_debug_counters")
    return list(c)


def time_ms(vol, cam, W, H, iters=20):
    vol.prepare()
    cb, fb = vol.camera_block(cam), vol.frame_block(W, H, None)
    r = vol.render(cam, W, H)
    ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
    ms = C.c_float(0)
    for it in (5, iters):
        N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), it, C.byref(ms)), "time")
    return ms.value


if len(sys.argv) > 1 and sys.argv[1] == "small":
    from test_gpu_skip import _scene, _sparse_pairs
    spec = _scene(128, _sparse_pairs(128, 3, count=16, noise=1), 150.0, "K1")
    spec.material.update(clim=(0.0, 255.0), render_mode="mip")
    scene = testing.build(spec)
    vol = scene.volume
    census(vol)
    r = vol.render(scene.camera, spec.width, spec.height, count_steps=True)
    torch.cuda.synchronize()
    print("small scene census [general, direct, brick, slabs, runs, zero, waves, skipped]:", census(vol), "steps", int(r.steps.sum()))
    sys.exit(0)

n, W, H = 1024, 1920, 1080
dev = torch.device("cuda", 0)
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
for camname in ("K1", "K2"):
    scene = testing.build(bench.config2_spec(n, W, H, camname, pairs))
    vol, cam = scene.volume, scene.camera
    for mode in ("lmip", "mip", "weighted_average"):
        vol.material.render_mode = mode
        for variant in ((0, 8) if mode != "weighted_average" else (0,)):
            N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
            census(vol)
            r = vol.render(cam, W, H, count_steps=True)
            torch.cuda.synchronize()
            c = census(vol)
            print(f"{camname} {mode} variant {variant}: {time_ms(vol, cam, W, H):.3f} ms; steps {int(r.steps.to(torch.int64).sum())/1e6:.1f} M; "
                  f"batches skipped {c[7]}, brick {c[2]}, direct {c[1]}, general {c[0]}; census {c}", flush=True)
