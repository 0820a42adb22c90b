import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from sub_volume_renderer_amd import _native as N, synth, testing, FrameRegion
dev = torch.device("cuda", 0)
n, W, H = 256, 640, 360
pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
spec = bench.config2_spec(n, W, H, "K1", pairs)
scene = testing.build(spec); vol, cam = scene.volume, scene.camera
def set_mode(full): vol.material.lmip_threshold = float("inf") if full else 127.5
def hits():
    r = vol.render(cam, W, H); torch.cuda.synchronize(); return int((r.flags == 2).sum())
set_mode(True); print("full hits", hits())
set_mode(False); print("lmip hits", hits())
# svr_time_render in lmip mode
vol.prepare(); cb, fb = vol.camera_block(cam), vol.frame_block(W, H, None)
r = vol.render(cam, W, H); torch.cuda.synchronize()
ob = N.Outputs(); ob.rgba = r.rgba.data_ptr(); ob.depth = r.depth.data_ptr(); ob.label = r.label.data_ptr(); ob.flags = r.flags.data_ptr()
ms = C.c_float(0); N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), 3, C.byref(ms)), "t")
set_mode(True); print("after time_render(lmip), full hits", hits())
set_mode(True); print("again full hits", hits())
