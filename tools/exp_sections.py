#!/usr/bin/env python3
"""Per-section cycle breakdown (instrumented build, `svr_debug_timers`) and batch census of the march, for any of the
bench's single-GPU workloads: whole frame plus a few small regions, full and / or LMIP mode.
usage: exp_sections.py [--config C2|C5] [--ring-storage native|float32] [--camera K1] [--variant 0] [--modes full,lmip]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prof_driver import build_scene  # noqa: E402
from sub_volume_renderer_amd import _native as N  # noqa: E402
from sub_volume_renderer_amd._wobject import FrameRegion  # noqa: E402

NAMES = ["prol", "span", "gen", "slabdma", "wait", "brick", "direct", "epil", "slabred", "slabsalu", "skip", "t11", "t12", "t13", "t14", "t15"]
CENSUS = ["general", "direct", "brick", "slabs", "runs", "zero", "waves", "skipped"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--ring-storage", default="native")
    ap.add_argument("--camera", default="K1")
    ap.add_argument("--variant", default="0")
    ap.add_argument("--modes", default="full")
    ap.add_argument("--regions", default="all")
    a = ap.parse_args()
    W, H = 1920, 1080
    scene, spec = build_scene(a.config, None, a.camera, a.ring_storage, W, H)
    vol, cam = scene.volume, scene.camera
    h = vol._rings.handle
    N.check(N.lib().svr_set_variant(h, int(a.variant, 0)), "variant")
    regions = [(956, 536, 8, 8), (400, 300, 8, 8), (928, 508, 64, 64), (0, 536, 1920, 64), (0, 0, 1920, 1080)]
    if a.regions == "frame":
        regions = regions[-1:]
    print(f"== {a.config} rings {vol._rings.density_storage} camera {a.camera} variant {a.variant}", flush=True)
    for mode in a.modes.split(","):
        vol.material.lmip_threshold = float("inf") if mode == "full" else float(spec.material["lmip_threshold"])
        for (x0, y0, w, hh) in regions:
            reg = FrameRegion.tile(x0, y0, w, hh)
            tm = (C.c_uint64 * 16)()
            dbg = (C.c_uint32 * 8)()
            N.lib().svr_debug_timers(h, tm, 1)
            N.lib().svr_debug_counters(h, dbg, 1)
            r = vol.render(cam, W, H, region=reg, count_steps=True)
            torch.cuda.synchronize()
            N.lib().svr_debug_timers(h, tm, 1)
            N.lib().svr_debug_counters(h, dbg, 1)
            tot = max(1, sum(tm[:15]))                                  # ([15] is a count: direct batches served by a micro-block copy)
            tline = " ".join(f"{nm}={100 * t / tot:.1f}%" for nm, t in zip(NAMES[:15], tm) if t)
            cline = " ".join(f"{nm}={v}" for nm, v in zip(CENSUS, dbg)) + f" direct_from_micro_blocks={tm[15]}"
            steps = r.steps.to(torch.int64)
            r = vol.render(cam, W, H, region=reg)
            vol.prepare()
            cb, fb = vol.camera_block(cam), vol.frame_block(W, H, reg)
            ob = N.Outputs()
            ob.rgba, ob.depth, ob.label, ob.flags = r.rgba.data_ptr(), r.depth.data_ptr(), r.label.data_ptr(), r.flags.data_ptr()
            ms = C.c_float(0)
            for iters in (5, 10):
                N.check(N.lib().svr_time_render(h, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(ms)), "time")
            print(f"{a.camera} {mode} region {w}x{hh}@({x0},{y0}): {ms.value * 1000:8.1f} us  steps {int(steps.sum()) / 1e6:9.3f} M  max/ray {int(steps.max())}"
                  f"\n      {tline} | cycles/wave={tot / max(1, dbg[6]):.0f}\n      census: {cline}", flush=True)


if __name__ == "__main__":
    main()
