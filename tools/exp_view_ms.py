#!/usr/bin/env python3
"""Kernel time (HIP events, svr_time_render) of the C2 / C5 frame from one view, full and LMIP mode.
usage: exp_view_ms.py [camera K1|K2|-x|-y|-z|diag] [ring storage native|float32] [config C2|C5] [twin|notwin]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prof_driver import build_scene  # noqa: E402
from sub_volume_renderer_amd import _native as N  # noqa: E402

cam = sys.argv[1] if len(sys.argv) > 1 else "K1"
storage = sys.argv[2] if len(sys.argv) > 2 else "native"
config = sys.argv[3] if len(sys.argv) > 3 else "C2"
twin = sys.argv[4] if len(sys.argv) > 4 else "twin"      # twin: the default ("auto": a micro-block copy of LOD 0, taken instead of bricks);
                                                          # twinall: + copies of the coarser LODs for waves that stage no bricks; notwin: rows only
W, H = 1920, 1080
scene, spec = build_scene(config, None, cam, storage, W, H, blocked_twin={"twin": "auto", "twinall": "all", "notwin": False}[twin])
vol = scene.volume
variant = int(os.environ.get("EXP_VARIANT", "0"), 0)                   # svr_set_variant bits for the run (include/svr.h)
N.check(N.lib().svr_set_variant(vol.prepare(), variant), "svr_set_variant")
r = vol.render(scene.camera, W, H)
out = []
for mode in ("full", "lmip"):
    vol.material.lmip_threshold = float("inf") if mode == "full" else float(spec.material["lmip_threshold"])
    vol.prepare()
    cb, fb = vol.camera_block(scene.camera), vol.frame_block(W, H, None)
    ob = N.Outputs()
    ob.rgba, ob.depth, ob.label, ob.flags = r.rgba.data_ptr(), r.depth.data_ptr(), r.label.data_ptr(), r.flags.data_ptr()
    ms = C.c_float(0)
    vals = []
    for _ in range(3):
        N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), 10, C.byref(ms)), "time")
        vals.append(ms.value)
    out.append(sorted(vals)[1])
print(f"{config} {storage:8s} {cam:5s} {twin:7s} full {out[0]:.4f} ms   lmip {out[1]:.4f} ms" + (f"   variant {variant:#x}" if variant else ""), flush=True)
