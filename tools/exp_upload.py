#!/usr/bin/env python3
"""Host -> ring upload throughput (PCIe-inclusive): the initial ring fill of config 2 from host-resident numpy
arrays, for several numbers of packing threads, with the time split the library itself reports
(svr_upload_stats: bytes through the pinned staging slots / wall time inside svr_upload_region), and the cost
per frame of blocking chunk-slab reloads.  usage: exp_upload.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sub_volume_renderer_amd import _native as N, synth, testing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = 1920, 1080
PCIE = 63.0   # gen5 x16 nominal; tools/exp_pcie.py measures 56 GB/s for plain pinned copies on the box
t = time.time()
dev = torch.device("cuda", 0)
pairs_dev = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
pairs = [(d.cpu().numpy(), l.cpu().numpy().view(np.uint32)) for d, l in pairs_dev]      # host-resident backing arrays
del pairs_dev
print(f"host volume ready in {time.time() - t:.1f}s, cpus {len(os.sched_getaffinity(0))}", flush=True)


def stats(vol, reset=False):
    b, s = C.c_uint64(0), C.c_double(0.0)
    N.check(N.lib().svr_upload_stats(vol._rings.handle, C.byref(b), C.byref(s), 1 if reset else 0), "stats")
    return b.value, s.value


for threads, block in (("1", ""), ("4", ""), ("8", ""), ("12", ""), ("", "4"), ("", "16"), ("", "48"), ("", "")):
    for key, v in (("SVR_PACK_THREADS", threads), ("SVR_UPLOAD_BLOCK_MIB", block)):
        if v:
            os.environ[key] = v
        else:
            os.environ.pop(key, None)
    spec = bench.config2_spec(n, W, H, "K2", pairs)
    spec.centers = []                                           # build first, time the fill alone
    scene = testing.build(spec)
    vol = scene.volume
    vol.prepare()
    centers = bench.config2_spec(n, W, H, "K2", pairs).centers
    vol.synchronize()
    stats(vol, reset=True)
    t = time.perf_counter()
    for position, sizes in centers:
        vol.center_on_position(position, sizes)
    vol.synchronize()
    dt = time.perf_counter() - t
    b, s = stats(vol)
    print(f"pack threads {threads or 'default':>7}, block {block or 'auto':>4} MiB: initial fill {b / 1e9:.2f} GB staged, wall {dt * 1e3:7.1f} ms = {b / dt / 1e9:5.1f} GB/s "
          f"({b / dt / 1e9 / PCIE:.2f} of PCIe gen5 x16); inside svr_upload_region {s * 1e3:7.1f} ms = {b / s / 1e9:5.1f} GB/s", flush=True)
    if threads or block:
        del scene, vol

# fly-through with blocking reloads: 2 voxels per frame along the view direction
spec = bench.config2_spec(n, W, H, "K2", pairs)
eye = np.array(spec.cam_position)
d = np.array(spec.cam_target) - eye
d /= np.linalg.norm(d)
for block in ("", "48"):                                        # "48": one block per staging slot, as before round 2's block split
    if block:
        os.environ["SVR_UPLOAD_BLOCK_MIB"] = block
    vol.center_on_position(tuple(eye))
    vol.synchronize()
    times = []
    stats(vol, reset=True)
    for k in range(120):
        p = eye + d * 2.0 * (k + 1)
        t = time.perf_counter()
        vol.center_on_position(tuple(p))
        vol.synchronize()
        times.append(time.perf_counter() - t)
    times = np.array(times) * 1e3
    b, s = stats(vol)
    print(f"block {block or 'auto'} MiB: blocking center_on_position per frame: median {np.median(times):.2f} ms, max {times.max():.2f} ms, "
          f"sum over frames with uploads {times[times > 1].sum():.1f} ms ({(times > 1).sum()}/120 frames); {b / 1e6:.0f} MB staged, "
          f"{b / max(s, 1e-9) / 1e9:.1f} GB/s inside the upload calls", flush=True)
