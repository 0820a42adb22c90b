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
    scene = testing.build(make_golden.specs()[name])
    prod, inst = testing.render_both(scene.volume, scene.camera, scene.width, scene.height)
    for r in (prod, inst):                              # the production kernel and the instrumented one
        np.testing.assert_array_equal(r.flags.cpu().numpy(), GOLD[f"{name}_flags"])
        np.testing.assert_array_equal(r.label_numpy(), GOLD[f"{name}_label"])
        np.testing.assert_allclose(r.rgba.cpu().numpy(), GOLD[f"{name}_rgba"], rtol=0, atol=1e-4)   # north_star tolerance
        np.testing.assert_allclose(r.depth.cpu().numpy(), GOLD[f"{name}_depth"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(inst.steps.cpu().numpy().view(np.uint32), GOLD[f"{name}_steps"])


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
    import torch

    from sub_volume_renderer_amd import _native as N

    vol, cam, W, H = c2.volume, c2.camera, 1920, 1080
    vol.material.lmip_threshold = 0.5 * 255.0
    N.check(N.lib().svr_set_variant(vol.prepare(), 0), "variant")
    _, full = testing.render_both(vol, cam, W, H)         # production == instrumented on the whole 1080p frame
    ref = {k: getattr(full, k).clone() for k in ("rgba", "depth", "label", "flags", "steps")}
    assert int((ref["flags"] == 2).sum()) > 100000
    # 2 x 4 tile grid of BASELINE config 3 (960 x 270 tiles), both kernels
    for ty in range(4):
        for tx in range(2):
            for r in testing.render_both(vol, cam, W, H, region=FrameRegion.tile(tx * 960, ty * 270, 960, 270)):
                for k in ref:
                    if getattr(r, k) is not None:
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
        for r in testing.render_both(vol, cam, W, H):
            for k in ref:
                if getattr(r, k) is not None:
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
    rep = testing.hold_both_to(ref, vol, cam, W, H, region=sample)     # production and instrumented kernels
    assert rep["total_steps"] > 30_000_000


@pytest.mark.parametrize("view", ["K2", "-y", "diag"])
def test_c2_other_views_oracle_parity_on_sampled_rows(c2, view):
    """The same full-size volume from inside (K2), along the y axis and along the space diagonal — the views whose waves
    take other paths than K1's (gathers only, long brick slabs, wrap events): every 60th row of the 1080p frame, both
    kernels, LMIP mode, against the oracle fed with the rings read back from HBM."""
    vol, W, H = c2.volume, 1920, 1080
    vol.material.lmip_threshold = 0.5 * 255.0
    n = 1024
    spec = c2.spec
    keep = (spec.cam_position, spec.cam_target)
    try:
        if view == "K2":
            centre = np.array([(n - 1) / 2.0] * 3)
            eye = centre + np.array([0.1 * n, 0.05 * n, -0.2 * n])
            spec.cam_position, spec.cam_target = tuple(eye), tuple(eye + np.array([0.6, 0.3, 0.74]))
        else:
            d = np.array({"-y": (0.02, -1, 0.03), "diag": (-1, -1, -1)}[view], float)
            d /= np.linalg.norm(d)
            c = (n - 1) / 2.0
            spec.cam_position, spec.cam_target = tuple(np.array([c, c, c]) + 1.6 * n * d), (c, c, c)
        cam = spec.camera()
        if "rings" not in c2.__dict__:
            c2.__dict__["rings"] = _rings_from_device(vol)
        sample = FrameRegion(0, 0, W, 18, 1, 60)
        m = dict(spec.material)
        m["lmip_threshold"] = 0.5 * 255.0
        ref = lmip.render(c2.rings, spec.matrices(), tuple(float(v) for v in vol._volume_dimensions), m, W, H, region=sample)
        rep = testing.hold_both_to(ref, vol, cam, W, H, region=sample)
        assert rep["n_hit"] > 1000 and rep["total_steps"] > 5_000_000
    finally:
        spec.cam_position, spec.cam_target = keep


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


# =====================================================================================================
# BASELINE config 1, literally: the arrays, material, rings and camera of scripts/multi_scale.py:31-85
# =====================================================================================================
def test_config1_literal_multi_scale_script_480x480():
    """(256,256,768)/(128,128,768)/(64,64,768) float arrays of tiled chunks, rings (2,2,2)/(4,4,4)/(8,8,8) chunks,
    480 x 480 canvas, camera at (-19.81, 7.5, 7.5): every pixel against the oracle."""
    import torch

    scene = testing.build(testing.multiscale_demo_spec(480, 480, tiles=16))
    assert [tuple(d.shape) for d, _ in scene.spec.pairs] == [(256, 256, 768), (128, 128, 768), (64, 64, 768)]
    ref = lmip.render_scene(scene)
    rep = testing.hold_both_to(ref, scene.volume, scene.camera, 480, 480)      # production and instrumented kernels
    assert set(np.unique(ref.label[ref.flags == 2])) == {0, 1, 2} and rep["n_hit"] > 10000


# =====================================================================================================
# BASELINE config 5 at its stated size: 2048^3, ~1 M labels, 256 hues, fog 0.05, threshold 0.3
# =====================================================================================================
def _rings_from_device(vol):
    rings = []
    for b in vol.wrapping_buffers:
        d, l = b.read_ring(Roi((0, 0, 0), b.shape_in_pixels))
        u = b.uniform_buffer.data
        rings.append(dict(density=d, labels=l, offset=tuple(int(v) for v in u["current_logical_offset_in_pixels"]),
                          shape=tuple(int(v) for v in u["current_logical_shape_in_pixels"]),
                          scale=tuple(float(v) for v in u["scale_factor"])))
    return rings


@pytest.fixture(scope="module")
def c5():
    import torch

    import bench
    from sub_volume_renderer_amd import synth
    from sub_volume_renderer_amd.pyramid import build_pyramid

    dev = torch.device("cuda", 0)
    n, W, H = 2048, 1920, 1080
    d0, l0 = synth.volume(n, 0, 1000003, xp=torch, device=dev, slab=16)
    pairs = build_pyramid(d0, l0, 3)
    torch.cuda.synchronize()
    scene = testing.build(bench.config5_spec(n, W, H, "K1", pairs))
    yield scene
    del scene, pairs, d0, l0
    torch.cuda.empty_cache()


def test_c5_geometry_and_ring_addressing_property(c5):
    from sub_volume_renderer_amd import synth

    vol = c5.volume
    assert [tuple(b.shape_in_pixels) for b in vol.wrapping_buffers] == [(1024, 1024, 1056), (1024, 1024, 1056), (512, 512, 576)]
    assert vol._rings.density_storage == "uint8" and len(vol.material.colors) == 256
    assert vol.material.fog_density == pytest.approx(0.05) and vol.material.lmip_threshold == pytest.approx(0.3 * 255)
    rng = np.random.default_rng(5)
    seen_labels = set()
    for lod, buf in enumerate(vol.wrapping_buffers):
        roi = buf._current_logical_roi_in_pixels.intersect(Roi((0, 0, 0), tuple(buf.backing_data.shape)))
        ring = np.array(buf.shape_in_pixels)
        for _ in range(5):
            shape = np.minimum(np.array([4, 6, 40]), np.array(roi.shape))
            off = np.array(roi.offset) + rng.integers(0, np.array(roi.shape) - shape + 1)
            want_d, want_l = synth.block(2048, lod, off.tolist(), shape.tolist(), 1000003)
            seen_labels.update(np.unique(want_l).tolist())
            for a0 in range(shape[0]):
                for a1 in range(0, shape[1], 2):
                    q = (off + np.array([a0, a1, 0])) % ring
                    run = int(min(shape[2], ring[2] - q[2]))
                    d, l = buf.read_ring(Roi(tuple(int(v) for v in q), (1, 1, run)))
                    np.testing.assert_array_equal(d[0, 0], want_d[a0, a1, :run].astype(np.float32))
                    np.testing.assert_array_equal(l[0, 0], want_l[a0, a1, :run])
    assert max(seen_labels) > 4096                         # the ~1 M label space is really in use


def test_c5_config3_tiles_equal_full_frame(c5):
    """The 2 x 4 grid of 960 x 270 tiles BASELINE configs 3 and 5 deal to 8 GPUs, rendered one after the other on
    this GPU, against the single full-frame render: every plane, bit for bit."""
    import torch

    vol, cam, W, H = c5.volume, c5.camera, 1920, 1080
    vol.material.lmip_threshold = 0.3 * 255.0
    _, full = testing.render_both(vol, cam, W, H)
    ref = {k: getattr(full, k).clone() for k in ("rgba", "depth", "label", "flags", "steps")}
    assert int((ref["flags"] == 2).sum()) > 100000
    assert len(torch.unique(ref["label"][ref["flags"] == 2])) > 1000       # many distinct labels reach the screen
    for ty in range(4):
        for tx in range(2):
            for r in testing.render_both(vol, cam, W, H, region=FrameRegion.tile(tx * 960, ty * 270, 960, 270)):
                for k in ref:
                    if getattr(r, k) is not None:
                        assert torch.equal(getattr(r, k), ref[k][ty * 270:(ty + 1) * 270, tx * 960:(tx + 1) * 960]), (k, tx, ty)


@pytest.mark.parametrize("mode", ["lmip", "full"])
def test_c5_oracle_parity_on_sampled_rows(c5, mode):
    """Every 40th row of the 1920x1080 frame of the 2048^3 scene against the oracle fed with the rings read back
    from HBM (1 M labels modulo a 256-entry HSV table, fog 0.05)."""
    import torch

    vol, cam, W, H = c5.volume, c5.camera, 1920, 1080
    thr = 0.3 * 255.0 if mode == "lmip" else float("inf")
    vol.material.lmip_threshold = thr
    if "rings" not in c5.__dict__:
        c5.__dict__["rings"] = _rings_from_device(vol)
    sample = FrameRegion(0, 0, W, 27, 1, 40)
    m = dict(c5.spec.material)
    m["lmip_threshold"] = thr
    ref = lmip.render(c5.rings, c5.spec.matrices(), tuple(float(v) for v in vol._volume_dimensions), m, W, H, region=sample)
    rep = testing.hold_both_to(ref, vol, cam, W, H, region=sample)     # production and instrumented kernels
    assert rep["total_steps"] > 30_000_000
    if mode == "full":
        del c5.__dict__["rings"]


# =====================================================================================================
# BASELINE config 4 at its stated size: a 4096^3 volume that is never resident, streamed by a fly-through
# =====================================================================================================
def test_c4_4096_streamed_flythrough_rings_and_pixels_match_oracle():
    """4096^3 behind lazy backing arrays (blocks generated on demand), C2's rings, K2 inside; 40 frames of the
    fly-through with center_on_position(asynchronous=True) after every render.  Checked: the windows really
    moved and reloaded; after the last load has landed the device rings equal the oracle's rings voxel for
    voxel (ring addressing buf[pos % ring] == data[pos] included) and the published ROIs are the synchronous
    ones; sampled rows of the last frame equal the oracle."""
    import torch

    import bench
    from oracle import ring_oracle
    from sub_volume_renderer_amd import synth

    n, W, H = 4096, 1920, 1080
    spec = bench.config4_spec(n, W, H)
    assert [p[0].shape for p in spec.pairs] == [(4096,) * 3, (2048,) * 3, (1024,) * 3]
    scene = testing.build(spec)
    vol = scene.volume
    start = [b._current_logical_roi_in_pixels for b in vol.wrapping_buffers]
    poses = bench.flythrough_poses(spec, 40, step=6.0)         # 240 voxels of travel: every level crosses chunks
    frames = 0
    for eye, target in poses:
        spec.cam_position, spec.cam_target = eye, target
        vol.render(spec.camera(), W, H)
        vol.center_on_position(eye, asynchronous=True)
        frames += 1
    vol.poll_uploads(wait=True)
    torch.cuda.synchronize()
    moved = [b._current_logical_roi_in_pixels != s for b, s in zip(vol.wrapping_buffers, start)]
    assert all(moved), moved
    # the oracle's host-side restatement, driven to the same final position (blocking loads)
    orac = ring_oracle.OracleSubVolume(list(spec.pairs), list(spec.ring_shapes), list(spec.chunk_shapes),
                                       world_inverse_matrix=np.linalg.inv(spec.world().matrix))
    orac.center_on_position(poses[-1][0], None)
    rings = []
    for lod, (b, ob) in enumerate(zip(vol.wrapping_buffers, orac.wrapping_buffers)):
        got = b._current_logical_roi_in_pixels
        assert (tuple(got.offset), tuple(got.shape)) == ob.current_logical_roi_in_pixels
        d, l = b.read_ring(Roi((0, 0, 0), b.shape_in_pixels))
        # inside the published window every ring slot holds its voxel (slots outside it may hold older chunks)
        roi = Roi(*ob.current_logical_roi_in_pixels).intersect(Roi((0, 0, 0), ob.backing_data.shape))
        ring = np.array(b.shape_in_pixels)
        idx = [np.arange(o, o + s) % r for o, s, r in zip(roi.offset, roi.shape, ring)]
        np.testing.assert_array_equal(d[np.ix_(*idx)], ob.texture[np.ix_(*idx)])
        np.testing.assert_array_equal(l[np.ix_(*idx)], ob.segmentations_texture[np.ix_(*idx)])
        # and against the closed form itself, at a few places of the window
        rng = np.random.default_rng(lod)
        for _ in range(3):
            shape = np.array([3, 5, 33])
            off = np.array(roi.offset) + rng.integers(0, np.array(roi.shape) - shape + 1)
            want_d, want_l = synth.block(n, lod, off.tolist(), shape.tolist())
            q = [np.arange(o, o + s) % r for o, s, r in zip(off, shape, ring)]
            np.testing.assert_array_equal(d[np.ix_(*q)], want_d.astype(np.float32))
            np.testing.assert_array_equal(l[np.ix_(*q)], want_l)
        u = b.uniform_buffer.data
        rings.append(dict(density=d, labels=l, offset=tuple(int(v) for v in u["current_logical_offset_in_pixels"]),
                          shape=tuple(int(v) for v in u["current_logical_shape_in_pixels"]),
                          scale=tuple(float(v) for v in u["scale_factor"])))
    sample = FrameRegion(0, 0, W, 27, 1, 40)
    cam = spec.camera()
    ref = lmip.render(rings, spec.matrices(), tuple(float(v) for v in vol._volume_dimensions), spec.material, W, H, region=sample)
    rep = testing.hold_both_to(ref, vol, cam, W, H, region=sample)     # production and instrumented kernels
    assert rep["n_hit"] > 1000 and frames == 40
