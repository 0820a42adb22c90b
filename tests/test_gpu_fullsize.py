"""GPU: committed golden vectors, and BASELINE config 2 at FULL size (1024^3, 3 LODs, 1920x1080)
through size-independent properties: tiling invariance, ring addressing buf[pos % ring] == data[pos],
kernel-variant equivalence, and oracle parity on a sub-sampled set of rows of the real frame."""
import os
import sys

import numpy as np
import pytest

from oracle import lmip
from sub_volume_renderer_amd import FrameRegion, Roi, testing

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import make_golden  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_golden.npz"))


@pytest.mark.parametrize("name", ["demo", "k1", "k2"])
def test_hip_matches_committed_golden(name):
    import torch

    scene = testing.build(make_golden.specs()[name])
    r = scene.volume.render(scene.camera, scene.width, scene.height, count_steps=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(r.flags.cpu().numpy(), GOLD[f"{name}_flags"])
    np.testing.assert_array_equal(r.label_numpy(), GOLD[f"{name}_label"])
    np.testing.assert_array_equal(r.steps.cpu().numpy().view(np.uint32), GOLD[f"{name}_steps"])
    np.testing.assert_allclose(r.rgba.cpu().numpy(), GOLD[f"{name}_rgba"], rtol=0, atol=1e-4)   # north_star tolerance
    np.testing.assert_allclose(r.depth.cpu().numpy(), GOLD[f"{name}_depth"], rtol=0, atol=1e-4)


@pytest.fixture(scope="module")
def c2():
    """BASELINE config 2, generated on the device exactly as bench.py does."""
    import torch

    import bench
    from sub_volume_renderer_amd import synth

    dev = torch.device("cuda", 0)
    n, W, H = 1024, 1920, 1080
    pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16) for k in range(3)]
    torch.cuda.synchronize()
    spec = bench.config2_spec(n, W, H, "K1", pairs)
    scene = testing.build(spec)
    yield scene
    del scene


def test_c2_ring_addressing_property(c2):
    """buf[pos % ring] == data[pos] (tests/wrapping_buffer/test_boundary_loading.py:133-160) on random
    blocks of every LOD's current ROI, against the closed-form synthetic volume."""
    from sub_volume_renderer_amd import synth

    rng = np.random.default_rng(0)
    for lod, buf in enumerate(c2.volume.wrapping_buffers):
        roi = buf._current_logical_roi_in_pixels
        assert roi is not None
        # the snapped ROI may extend past the data end (1024 is not a multiple of 48); nothing is loaded there
        roi = roi.intersect(Roi((0, 0, 0), tuple(buf.backing_data.shape)))
        ring = np.array(buf.shape_in_pixels)
        for _ in range(6):
            shape = np.minimum(np.array([5, 7, 40]), np.array(roi.shape))
            off = np.array(roi.offset) + rng.integers(0, np.array(roi.shape) - shape + 1)
            want_d, want_l = synth.block(1024, lod, off.tolist(), shape.tolist())
            # read voxel rows through the ring (may wrap): one logical row at a time along a2
            for a0 in range(shape[0]):
                for a1 in range(0, shape[1], 3):
                    p = off + np.array([a0, a1, 0])
                    q = p % ring
                    run = int(min(shape[2], ring[2] - q[2]))
                    d, l = buf.read_ring(Roi(tuple(int(v) for v in q), (1, 1, run)))
                    np.testing.assert_array_equal(d[0, 0], want_d[a0, a1, :run].astype(np.float32))
                    np.testing.assert_array_equal(l[0, 0], want_l[a0, a1, :run])


def test_c2_tiles_stripes_and_variants_equal_full_frame(c2):
    import ctypes as C

    import torch

    from sub_volume_renderer_amd import _native as N

    vol, cam, W, H = c2.volume, c2.camera, 1920, 1080
    vol.material.lmip_threshold = 0.5 * 255.0
    N.check(N.lib().svr_set_variant(vol.prepare(), 0), "variant")
    full = vol.render(cam, W, H, count_steps=True)
    torch.cuda.synchronize()
    ref = {k: getattr(full, k).clone() for k in ("rgba", "depth", "label", "flags", "steps")}
    assert int((ref["flags"] == 2).sum()) > 100000
    # 2 x 4 tile grid of BASELINE config 3 (960 x 270 tiles)
    for ty in range(4):
        for tx in range(2):
            r = vol.render(cam, W, H, region=FrameRegion.tile(tx * 960, ty * 270, 960, 270), count_steps=True)
            torch.cuda.synchronize()
            for k in ref:
                assert torch.equal(getattr(r, k), ref[k][ty * 270:(ty + 1) * 270, tx * 960:(tx + 1) * 960]), (k, tx, ty)
    # interleaved 16-row bands for 8 ranks
    for rank in (0, 3, 7):
        reg = FrameRegion.stripes(W, H, rank, 8, 16)
        r = vol.render(cam, W, H, region=reg)
        torch.cuda.synchronize()
        rows = [reg.y0 + (q // 16) * reg.band_pitch + q % 16 for q in range(reg.out_h)]
        keep = [q for q, y in enumerate(rows) if y < H]
        assert torch.equal(r.rgba[keep], ref["rgba"][[rows[q] for q in keep]])
        assert torch.equal(r.label[keep], ref["label"][[rows[q] for q in keep]])
    # every kernel variant gives the same frame, bit for bit
    # (bricks never / always, simple march, tile shape, 2x2-wave workgroups, placement policies)
    for variant in (0x100, 0x200, 0x001, 0x250, 0x002, 0x2000, 0x4000, 0xA202):
        N.check(N.lib().svr_set_variant(vol.prepare(), variant), "variant")
        r = vol.render(cam, W, H, count_steps=True)
        torch.cuda.synchronize()
        for k in ref:
            assert torch.equal(getattr(r, k), ref[k]), (k, hex(variant))
    N.check(N.lib().svr_set_variant(vol.prepare(), 0), "variant")


@pytest.mark.parametrize("mode", ["lmip", "full"])
def test_c2_oracle_parity_on_sampled_rows(c2, mode):
    """Every 24th row of the real 1920x1080 frame against the oracle fed with the rings read back from HBM."""
    import torch

    vol, cam, W, H = c2.volume, c2.camera, 1920, 1080
    thr = 0.5 * 255.0 if mode == "lmip" else float("inf")
    vol.material.lmip_threshold = thr
    rings = []
    for b in vol.wrapping_buffers:
        d, l = b.read_ring(Roi((0, 0, 0), b.shape_in_pixels))
        u = b.uniform_buffer.data
        rings.append(dict(density=d, labels=l, offset=tuple(int(v) for v in u["current_logical_offset_in_pixels"]),
                          shape=tuple(int(v) for v in u["current_logical_shape_in_pixels"]),
                          scale=tuple(float(v) for v in u["scale_factor"])))
    sample = FrameRegion(0, 0, W, 45, 1, 24)
    m = dict(c2.spec.material)
    m["lmip_threshold"] = thr
    ref = lmip.render(rings, c2.spec.matrices(), tuple(float(v) for v in vol._volume_dimensions), m, W, H, region=sample)
    res = vol.render(cam, W, H, region=sample, count_steps=True)
    torch.cuda.synchronize()
    rep = testing.compare(res, ref)
    assert rep["flags_equal"] and rep["labels_equal"] and rep["steps_equal"], rep
    assert rep["rgba_max_rel"] <= 1e-4 and rep["depth_max_abs"] <= 1e-4, rep
    assert rep["total_steps"] > 30_000_000


def test_one_rank_rccl_pipeline_equals_single_gpu_frame():
    """bench.py's N > 1 pipeline on real RCCL in a one-rank process group (own process: the group and the
    library state stay out of this one): row bands, asynchronous gather of device tensors on four streams,
    un-tile kernel; the gathered frame must equal the single-GPU render."""
    import json
    import subprocess
    import sys

    import socket

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:                        # a port nobody holds right now (no fixed rendezvous port)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-collective", "--check",
                          "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--modes", "full",
                          "--volume-n", "256", "--width", "640", "--height", "360"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["config"]["parallelism"].startswith("frame row-bands x1")
    assert res["check_gathered"]["gathered_frame_equals_single_gpu_render"] is True
    assert res["check_gathered"]["alpha1_gathered"] > 1000
