"""Run by tests/test_gpu_collective.py in a process of its own (the process group and the communicator stay out of the
pytest process): a ONE-rank RCCL group drives the whole N > 1 pipeline of `sub_volume_renderer_amd.distributed` on one GPU.
usage: collective_worker.py <scenario>   prints one JSON line.

scenarios
  streams      `gather_async` on one stream, `finish` on ANOTHER, region buffers rewritten right after `finish`:
               the gathered frame must be the frame that was rendered (an event orders finish() behind the
               transfers, on root and non-root alike; the advisor's round-2 finding)
  probe_fails  `init_comm`'s self-check is made to fail (the probe gather reports a wrong pixel): every rank must fall
               back to torch.distributed.gather, and that transport must reassemble the same frame
  no_rccl      `svr_comm_unique_id` fails (as on a machine without librccl): same fallback
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sub_volume_renderer_amd import FrameRegion, _native as N, testing  # noqa: E402
from sub_volume_renderer_amd.distributed import TiledFrame  # noqa: E402


def main(scenario):
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    W, H = 320, 200
    scene = testing.build(testing.synthetic_spec(64, W, H, threshold=0.4))
    vol, cam = scene.volume, scene.camera
    whole = vol.render(cam, W, H, region=FrameRegion.full(W, H))
    torch.cuda.synchronize()
    want = {k: getattr(whole, k).clone() for k in ("rgba", "depth", "label")}
    tiled = TiledFrame(W, H, 0, 1, band_h=16, force_collective=True)
    out = {"scenario": scenario}
    if scenario == "probe_fails":
        real = TiledFrame.gather
        calls = []

        def corrupt(self, local, dst=0, volume=None):          # the probe comes back with a wrong pixel
            got = real(self, local, dst=dst, volume=volume)
            calls.append(1)
            if len(calls) == 1 and got is not None:
                got[0, 0] += 1.0
            return got

        TiledFrame.gather = corrupt
        ok = tiled.init_comm(vol)
        TiledFrame.gather = real
        out["init_comm"] = bool(ok)
    elif scenario == "no_rccl":
        lib = N.lib()
        real = lib.svr_comm_unique_id
        lib.svr_comm_unique_id = lambda buf: -2                  # SVR_ERR_HIP, as when librccl cannot be loaded
        try:
            out["init_comm"] = bool(tiled.init_comm(vol))
        finally:
            lib.svr_comm_unique_id = real
    else:
        out["init_comm"] = bool(tiled.init_comm(vol))
    out["transport"] = tiled.transport
    s_render, s_finish = torch.cuda.Stream(), torch.cuda.Stream()
    res = vol._outputs(tiled.region.out_h, tiled.region.out_w, False)
    frames_ok = []
    for k in range(6):
        with torch.cuda.stream(s_render):
            vol.render(cam, W, H, region=tiled.region, out=res)
            tiled.gather_async((res.rgba, res.depth, res.label), slot=0, dst=0, volume=vol)
        with torch.cuda.stream(s_finish):                      # ANOTHER stream un-tiles ...
            got = tiled.finish(0, dst=0)
            for t in (res.rgba, res.depth, res.label):        # ... and the region buffers are rewritten right behind it
                t.fill_(-7)
        s_finish.synchronize()
        frames_ok.append(all(torch.equal(g, want[n]) for g, n in zip(got, ("rgba", "depth", "label"))))
        s_render.wait_stream(s_finish)
    out["frames_equal_single_gpu_render"] = frames_ok
    print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
