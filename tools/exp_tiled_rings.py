#!/usr/bin/env python3
"""EXPERIMENT: what a ring layout that is compact in 3-D would buy the gather path.

Two runs of the same scene in two processes (one library each):
  exp_tiled_rings.py ref   <camera> <storage> <config>     with SVR_LIB=_ab/libs/exp.so    (linear [z][y][x] rings)
  exp_tiled_rings.py tiled <camera> <storage> <config>     with SVR_LIB=_ab/libs/tiled.so  (-DSVR_EXPERIMENTS,-DSVR_EXP_TILED=<mask of LODs>)
                                                            and EXP_TILED_LODS=<the same mask> (default 7)
`ref` times the shipped routing and "gathers only" (variant bit 8) and keeps the frames; `tiled` re-lays the density rings
out in place into 128-byte micro-blocks (8x4x4 voxels of one byte, 4x4x2 of four), renders with the kernel that addresses
them that way (gathers only), checks its frames against the kept ones bit for bit and times it.
Rings of 4 GiB or more (C5) need SVR_FORCE_ZSPLIT=1020 in BOTH runs (parts of whole blocks of planes)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prof_driver import build_scene  # noqa: E402
from sub_volume_renderer_amd import _native as N  # noqa: E402

phase = sys.argv[1]
cam = sys.argv[2] if len(sys.argv) > 2 else "K1"
storage = sys.argv[3] if len(sys.argv) > 3 else "native"
config = sys.argv[4] if len(sys.argv) > 4 else "C5"
TILED = int(os.environ.get("EXP_TILED_LODS", "7"), 0)                    # mask of LODs to re-lay out: the library's -DSVR_EXP_TILED=<mask>
MODES = tuple(os.environ.get("EXP_MODES", "full,lmip").split(","))      # (EXP_MODES=full: a counter pass over one mode's dispatches)
W, H = 1920, 1080
KEEP = f"/tmp/exp_tiled_{config}_{storage}_{cam}"              # (two 50 MB frame sets: only needed between the two runs)
scene, spec = build_scene(config, None, cam, storage, W, H)
vol = scene.volume
r = vol.render(scene.camera, W, H)
torch.cuda.synchronize()


class _Raw:                      # a device allocation of the native library as a torch tensor (no copy)
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def retile(lod):
    """ring [z][y][x] -> [bz][by][bx][z in block][y in block][x in block], in place"""
    dens, lab = C.c_void_p(), C.c_void_p()
    N.check(N.lib().svr_lod_device_ptrs(vol._rings.handle, lod, C.byref(dens), C.byref(lab)), "ptrs")
    rz, ry, rx = [int(v) for v in vol._rings.ring_shapes[lod]]
    es = {"uint8": 1, "uint16": 2}.get(vol._rings.density_storage, 4)
    xb, yb, zb = (8, 4, 4) if es == 1 else (4, 4, 4) if es == 2 else (4, 4, 2)
    assert rx % xb == 0 and ry % yb == 0 and rz % zb == 0, (rz, ry, rx)
    flat = torch.as_tensor(_Raw(dens.value, rz * ry * rx * es), device="cuda")
    v = flat.view(rz // zb, zb, ry // yb, yb, rx // xb, xb * es)
    # a slab of block planes at a time: the temporary stays small beside an 8 GiB ring
    step = max(1, (1 << 28) // (zb * ry * rx * es))
    for b0 in range(0, rz // zb, step):
        part = v[b0:b0 + step]
        t = part.permute(0, 2, 4, 1, 3, 5).contiguous()
        part.reshape(-1).copy_(t.reshape(-1))          # (a leading-dimension slice of a contiguous view: reshape is a view)
    torch.cuda.synchronize()


def frames_and_times(variant):
    N.check(N.lib().svr_set_variant(vol._rings.handle, variant), "variant")
    got = {}
    for mode in MODES:
        vol.material.lmip_threshold = float("inf") if mode == "full" else float(spec.material["lmip_threshold"])
        res = vol.render(scene.camera, W, H)
        torch.cuda.synchronize()
        planes = {k: getattr(res, k).cpu().numpy().copy() for k in ("rgba", "depth", "label", "flags")}
        vol.prepare()
        cb, fb = vol.camera_block(scene.camera), vol.frame_block(W, H, None)
        ob = N.Outputs()
        ob.rgba, ob.depth, ob.label, ob.flags = res.rgba.data_ptr(), res.depth.data_ptr(), res.label.data_ptr(), res.flags.data_ptr()
        ms = C.c_float(0)
        vals = []
        for _ in range(3):
            N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), 10, C.byref(ms)), "time")
            vals.append(ms.value)
        got[mode] = (planes, sorted(vals)[1])
    return got


if phase == "ref":
    a = frames_and_times(0)
    b = frames_and_times(1 << 8)
    for mode in MODES:
        for k in a[mode][0]:
            assert a[mode][0][k].tobytes() == b[mode][0][k].tobytes(), (mode, k)
        np.savez(KEEP + f"_{mode}.npz", **a[mode][0])
    print(f"{config} {storage:8s} {cam:5s} linear rings, shipped routing: " + "  ".join(f"{m} {a[m][1]:.4f} ms" for m in MODES) + " | "
          "gathers only: " + "  ".join(f"{m} {b[m][1]:.4f} ms" for m in MODES), flush=True)
else:
    nl = len(vol._rings.ring_shapes)
    for lod in range(nl):
        if (TILED >> lod) & 1:
            retile(lod)
    linear = ~TILED & ((1 << nl) - 1)
    # bricks only from the rings that stayed linear (variant bits 24-31), or never (bit 8)
    t = frames_and_times((linear << 24) if linear else (1 << 8))
    same = True
    for mode in MODES:
        kept = np.load(KEEP + f"_{mode}.npz")
        for k in t[mode][0]:
            if kept[k].tobytes() != t[mode][0][k].tobytes():
                same = False
                print(f"  MISMATCH {mode} {k}: {int((kept[k] != t[mode][0][k]).sum())} elements differ")
    print(f"{config} {storage:8s} {cam:5s} micro-block rings on LODs {TILED:#x} (gathers there):  " + "  ".join(f"{m} {t[m][1]:.4f} ms" for m in MODES) + " | "
          f"frames identical to the linear rings': {same}", flush=True)
