#!/usr/bin/env python3
"""Minimal driver for rocprofv3 passes: N renders of ONE march mode of one workload (so every march dispatch in the
trace is the same workload).
usage: prof_driver.py [full|lmip] [n] [iters] [variant] [camera] [--config C2|C5] [--ring-storage native|float32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402
from sub_volume_renderer_amd.pyramid import build_pyramid  # noqa: E402


def build_scene(config="C2", n=None, camera="K1", ring_storage="native", W=1920, H=1080, blocked_twin="auto"):
    """The bench's scene for `config` (bench.py builds it the same way), camera K1 / K2 or one of the axis views."""
    n = n or {"C2": 1024, "C5": 2048}[config]
    dev = torch.device("cuda", 0)
    n_labels = 1000003 if config == "C5" else 4096
    d0, l0 = synth.volume(n, 0, n_labels, xp=torch, device=dev, slab=16 if n >= 512 else 64)
    pairs = build_pyramid(d0, l0, 3)
    del d0, l0
    torch.cuda.synchronize()
    spec = (bench.config5_spec if config == "C5" else bench.config2_spec)(n, W, H, camera if camera in ("K1", "K2") else "K1", pairs)
    dirs = {"-x": (-1, 0.02, 0.03), "-y": (0.02, -1, 0.03), "-z": (0.02, 0.03, -1), "diag": (-1, -1, -1)}
    if camera in dirs:
        c = (n - 1) / 2.0
        d = np.array(dirs[camera], float)
        d /= np.linalg.norm(d)
        spec.cam_position = tuple(np.array([c, c, c]) + 1.6 * n * d)
        spec.cam_target = (c, c, c)
    spec.ring_storage = ring_storage
    spec.blocked_twin = blocked_twin
    return testing.build(spec), spec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", nargs="?", default="full")
    ap.add_argument("n", nargs="?", type=int, default=None)
    ap.add_argument("iters", nargs="?", type=int, default=5)
    ap.add_argument("variant", nargs="?", default="0")
    ap.add_argument("camera", nargs="?", default="K1")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--ring-storage", default="native")
    ap.add_argument("--no-blocked-twin", action="store_true", help="rings in rows only (no micro-block copy of LOD 0)")
    a = ap.parse_args()
    scene, spec = build_scene(a.config, a.n, a.camera, a.ring_storage, blocked_twin=False if a.no_blocked_twin else "auto")
    vol = scene.volume
    N.check(N.lib().svr_set_variant(vol._rings.handle, int(a.variant, 0)), "variant")
    vol.material.lmip_threshold = float("inf") if a.mode == "full" else float(spec.material["lmip_threshold"])
    for _ in range(a.iters):
        vol.render(scene.camera, 1920, 1080)
    torch.cuda.synchronize()
    print("done", a.mode, a.config, a.n, a.iters, a.variant, a.camera, a.ring_storage)
