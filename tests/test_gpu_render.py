"""GPU: the HIP LMIP march through the C ABI against the CPU oracle on identical inputs.
Bar (BASELINE.json north_star): flags / labels / step counts bit-exact; RGBA and depth within 1e-4."""
import numpy as np
import pytest

from oracle import lmip
from sub_volume_renderer_amd import FrameRegion, testing

pytestmark = pytest.mark.gpu

RGBA_TOL = 1e-4      # absolute on values <= 1, relative above (raw-intensity scenes exceed 1)
DEPTH_TOL = 1e-4


def check(scene, region=None, ref=None, want_hits=True):
    """Both instantiations of the march against the oracle: the PRODUCTION kernel (`count_steps=False`, the code object
    bench.py times: `march_span<..., COUNT=false>`) on flags / labels / RGBA / depth, and the instrumented one
    (`COUNT=true`) on those planes plus the executed-iteration counts.  `render_both` also holds the two frames to
    each other bit for bit."""
    prod, res = testing.render_both(scene.volume, scene.camera, scene.width, scene.height, region=region)
    if ref is None:
        ref = lmip.render_spec(scene.spec, region=region)
    for which, r in (("production", prod), ("instrumented", res)):
        rep = testing.compare(r, ref)
        assert rep["flags_equal"], (which, rep)
        assert rep["labels_equal"], (which, rep)
        assert rep["rgba_max_rel"] <= RGBA_TOL, (which, rep)
        assert rep["depth_max_abs"] <= DEPTH_TOL, (which, rep)
    assert "steps_equal" not in testing.compare(prod, ref)             # the production kernel carries no counter
    assert rep["steps_equal"], rep
    if want_hits:
        assert rep["n_hit"] > 0, rep
    return res, ref, rep


def test_config1_multi_scale_demo():
    """BASELINE config 1: scripts/multi_scale.py arrays/material/camera (reduced tiling), 3 LODs visible."""
    scene = testing.build(testing.multiscale_demo_spec(240, 240, tiles=6))
    _, ref, rep = check(scene)
    assert set(np.unique(ref.label[ref.flags == 2])) == {0, 1, 2}      # every LOD contributes hits


@pytest.mark.parametrize("storage", ["native", "float32"], ids=["u8rings", "f32rings"])
@pytest.mark.parametrize("inside", [False, True], ids=["K1_outside", "K2_inside"])
@pytest.mark.parametrize("full", [False, True], ids=["lmip", "full"])
def test_synthetic_three_lods(inside, full, storage):
    spec = testing.synthetic_spec(64, 160, 96, inside=inside, full=full)
    spec.ring_storage = storage
    scene = testing.build(spec)
    assert scene.volume._rings.density_storage == ("uint8" if storage == "native" else "float32")
    _, ref, rep = check(scene, want_hits=not full)
    if full:
        assert rep["n_hit"] == 0 and rep["n_miss"] > 0                 # threshold = +inf: every ray runs all nsteps


@pytest.mark.parametrize("variant", [0x000, 0x200, 0x100, 0x001, 0x250, 0x230, 0x202, 0x2200, 0x4200, 0xE202],
                         ids=["auto", "brick", "nobrick", "simple", "brick16x4", "brick4x16", "brick-wg4",
                              "brick-contiguous", "brick-tilewise", "brick-wg4-static64"])
@pytest.mark.parametrize("cam", ["K1", "K2", "-x", "-y", "-z"])
def test_kernel_variants_bit_identical(variant, cam):
    """Every kernel variant (LDS bricks on/off, span vs simple march, tile shapes, workgroup size,
    block placement) must agree with the
    oracle bit for bit, for views entering through each face (rings 16-aligned so bricks are used)."""
    import ctypes as C

    from sub_volume_renderer_amd import _native as N

    spec = testing.synthetic_spec(128, 192, 128, inside=(cam == "K2"), threshold=0.45,
                                  chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)],
                                  ring_shapes=[(6, 6, 3), (12, 12, 3), (16, 16, 2)])
    if cam.startswith("-"):
        d = {"-x": (-1, 0.05, 0.08), "-y": (0.06, -1, 0.04), "-z": (0.03, 0.07, -1)}[cam]
        d = np.array(d) / np.linalg.norm(d)
        c = 63.5
        spec.cam_position = tuple(np.array([c, c, c]) + 1.6 * 128 * d)
        spec.cam_target = (c, c, c)
    scene = testing.build(spec)
    N.check(N.lib().svr_set_variant(scene.volume.prepare(), variant), "svr_set_variant")
    check(scene)


def test_synthetic_128_srgb_off_and_many_colors():
    spec = testing.synthetic_spec(128, 200, 120, threshold=0.3, fog_density=0.05, ncolors=256, n_labels=100003)
    spec.colorspace = "linear"
    spec.material.update(gamma=0.7, opacity=0.8, lmip_max_samples=3, lmip_fall_off=0.9)
    scene = testing.build(spec)
    check(scene)


def test_single_lod_and_empty_rois():
    from sub_volume_renderer_amd import synth

    pairs = [synth.volume(32, 0)]
    spec = testing.synthetic_spec(32, 96, 64, pairs=pairs, chunk_shapes=[(8, 8, 8)], ring_shapes=[(4, 4, 4)],
                                  threshold=0.2)
    scene = testing.build(spec)
    check(scene)
    # nothing loaded at all: ROI None -> offset = shape = 0 -> every sample is 0 -> all MISS, black, alpha 1
    spec2 = testing.synthetic_spec(32, 64, 48, pairs=pairs, chunk_shapes=[(8, 8, 8)], ring_shapes=[(4, 4, 4)])
    spec2.centers = []
    scene2 = testing.build(spec2)
    res, ref, rep = check(scene2, want_hits=False)
    assert rep["n_hit"] == 0
    rgba = res.rgba.cpu().numpy()
    miss = ref.flags == 1
    assert np.all(rgba[miss] == np.array([0, 0, 0, 1], np.float32))


def test_world_transform_anisotropic_scale():
    spec = testing.synthetic_spec(64, 128, 96)
    spec.world_scale = (1.0, 1.0, 3.0)          # scripts/mouse.py:90-91 style (world.scale_z)
    spec.world_position = (5.0, -3.0, 2.0)
    c = 31.5
    spec.cam_target = (c + 5.0, c - 3.0, 3 * c + 2.0)
    spec.cam_position = (c - 90.0, c + 40.0, 3 * c + 60.0)
    spec.centers = [((c + 5.0, c - 3.0, 3 * c + 2.0), None)]
    check(testing.build(spec))


def test_fly_through_sequence_of_centers():
    """center_on_position per frame along a path: diff-loads + ring wrap-around, render after each."""
    spec = testing.synthetic_spec(64, 96, 64, inside=True)
    scene = testing.build(spec)
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    for k in range(1, 6):
        p = eye + d * 5.0 * k
        spec.cam_position = tuple(p)
        spec.cam_target = tuple(p + d)
        spec.centers.append((tuple(p), None))
        scene.volume.center_on_position(tuple(p))
        scene.camera = spec.camera()
        check(scene, want_hits=False)


def test_tiles_and_stripes_equal_full_frame():
    scene = testing.build(testing.synthetic_spec(64, 200, 130))
    _, full, _ = check(scene)
    W, H = 200, 130
    # 2 x 4 tile grid (BASELINE config 3 geometry), odd sizes on purpose
    for ty in range(4):
        for tx in range(2):
            x0, y0 = tx * 100, ty * 33
            w, h = 100, min(33, H - y0) if ty < 3 else H - y0
            reg = FrameRegion.tile(x0, y0, w, h)
            res, ref, _ = check(scene, region=reg, want_hits=False)
            np.testing.assert_array_equal(ref.label, full.label[y0:y0 + h, x0:x0 + w])
            np.testing.assert_array_equal(res.rgba.cpu().numpy(), _render_full(scene)[y0:y0 + h, x0:x0 + w])
    # interleaved stripes for 3 ranks, 8-row bands, padded
    whole = _render_full(scene)
    for rank in range(3):
        reg = FrameRegion.stripes(W, H, rank, 3, band_h=8)
        res, ref, _ = check(scene, region=reg, want_hits=False)
        got = res.rgba.cpu().numpy()
        for r in range(reg.out_h):
            y = reg.y0 + (r // 8) * reg.band_pitch + r % 8
            if y < H:
                np.testing.assert_array_equal(got[r], whole[y])
            else:
                assert np.all(got[r] == 0)           # padding rows are "discarded"


_full_cache = {}


def _render_full(scene):
    import torch

    key = id(scene)
    if key not in _full_cache:
        r = scene.volume.render(scene.camera, scene.width, scene.height)
        torch.cuda.synchronize()
        _full_cache[key] = r.rgba.cpu().numpy().copy()
    return _full_cache[key]


def test_untile_stripes_roundtrip():
    import ctypes as C

    import torch

    from sub_volume_renderer_amd import _native as N

    scene = testing.build(testing.synthetic_spec(64, 120, 70))
    whole = _render_full(scene)
    W, H, nr, bh = 120, 70, 4, 8
    parts = []
    for rank in range(nr):
        reg = FrameRegion.stripes(W, H, rank, nr, band_h=bh)
        r = scene.volume.render(scene.camera, W, H, region=reg)
        torch.cuda.synchronize()
        parts.append(r.rgba.clone())
    gathered = torch.stack(parts)                     # [nranks, out_h, W, 4] == what the RCCL gather delivers
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    N.check(N.lib().svr_untile_stripes(scene.volume._rings.handle, C.c_void_p(gathered.data_ptr()),
                                       C.c_void_p(out.data_ptr()), W, H, bh, nr, parts[0].shape[0], 16,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "untile")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), whole)


def test_material_update_takes_effect():
    scene = testing.build(testing.synthetic_spec(64, 96, 64))
    check(scene)
    scene.spec.material.update(lmip_threshold=0.25 * 255, fog_color=(0.1, 0.2, 0.3), fog_density=0.4)
    m = scene.volume.material
    m.lmip_threshold = 0.25 * 255
    m.fog_color = (0.1, 0.2, 0.3)
    m.fog_density = 0.4
    check(scene)
    scene.spec.material["colors"] = [(0.9, 0.5, 1.0), (0.1, 0.0, 1.0)]       # s == 0 branch of hsv_to_rgb
    m.colors = scene.spec.material["colors"]
    check(scene)


def test_async_streaming_protocol_never_tears():
    """center_on_position(asynchronous=True): after every call the frame must equal the oracle for the
    ROI state the product has PUBLISHED (shrunk = old & new while chunks stream in, then the full new
    ROI), whatever the timing; the final state equals the synchronous result."""
    import torch

    spec = testing.synthetic_spec(64, 96, 64, inside=True)
    scene = testing.build(spec)
    vol = scene.volume
    orac = lmip.oracle_volume(spec)                      # applies the same initial center_on_position
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    shrunk_seen = 0
    for k in range(1, 9):
        p = eye + d * 6.0 * k
        spec.cam_position = tuple(p)
        spec.cam_target = tuple(p + d)
        scene.camera = spec.camera()
        vol.center_on_position(tuple(p), asynchronous=True)
        orac.center_on_position(tuple(p))                # oracle textures now hold the new chunks too
        for attempt in range(2):                         # frame 0: maybe still streaming; frame 1: after landing
            if attempt == 1:
                vol.poll_uploads(wait=True)
            if attempt == 0:
                # chunks may land (and the window grow) between two draws: ONE draw per published state, alternating
                # between the production kernel and the instrumented one
                res = vol.render(scene.camera, spec.width, spec.height, count_steps=bool(k & 1))
                torch.cuda.synchronize()
            else:
                _, res = testing.render_both(vol, scene.camera, spec.width, spec.height)    # landed: the state is stable
            rings = lmip.rings_of(orac)
            for ring, b, ob in zip(rings, vol.wrapping_buffers, orac.wrapping_buffers):
                u = b.uniform_buffer.data                # what the product published for THIS frame
                ring["offset"] = tuple(int(v) for v in u["current_logical_offset_in_pixels"])
                ring["shape"] = tuple(int(v) for v in u["current_logical_shape_in_pixels"])
                if ob.current_logical_roi_in_pixels is not None and tuple(ring["shape"]) != tuple(ob.uniform()["shape"]):
                    shrunk_seen += 1
            ref = lmip.render(rings, spec.matrices(), orac.volume_dimensions_shader, spec.material,
                              spec.width, spec.height)
            rep = testing.compare(res, ref)
            assert rep["flags_equal"] and rep["labels_equal"] and rep.get("steps_equal", True), (k, attempt, rep)
            assert rep["rgba_max_rel"] <= RGBA_TOL and rep["depth_max_abs"] <= DEPTH_TOL, (k, attempt, rep)
        # after landing, the published state is exactly the synchronous one
        for b, ob in zip(vol.wrapping_buffers, orac.wrapping_buffers):
            got = b._current_logical_roi_in_pixels
            assert (None if got is None else (tuple(got.offset), tuple(got.shape))) == ob.current_logical_roi_in_pixels
            np.testing.assert_array_equal(b.texture.data, ob.texture)


@pytest.mark.parametrize("world,band_h", [(2, 16), (8, 16), (3, 8)])
def test_tiledframe_untile_paths_equal_full_render(world, band_h):
    """The product's TiledFrame (band mapping + un-tile, GPU kernel and CPU path) reassembles the
    per-rank band renders into exactly the single-GPU frame (one process plays every rank)."""
    import torch

    from sub_volume_renderer_amd.distributed import TiledFrame

    W, H = 200, 135
    scene = testing.build(testing.synthetic_spec(64, W, H))
    whole = scene.volume.render(scene.camera, W, H)
    torch.cuda.synchronize()
    whole_rgba = whole.rgba.clone()
    bands = []
    for rank in range(world):
        tf = TiledFrame(W, H, rank, world, band_h)
        r = scene.volume.render(scene.camera, W, H, region=tf.region)
        torch.cuda.synchronize()
        bands.append(r.rgba.clone())
    tf0 = TiledFrame(W, H, 0, world, band_h)
    gathered = torch.stack(bands)
    out_gpu = tf0.untile(gathered, torch.empty_like(whole_rgba), volume=scene.volume)
    torch.cuda.synchronize()
    assert torch.equal(out_gpu, whole_rgba)
    out_cpu = tf0.untile(gathered.cpu(), torch.empty_like(whole_rgba).cpu())
    assert torch.equal(out_cpu, whole_rgba.cpu())


@pytest.mark.parametrize("simple", [False, True], ids=["span", "simple"])
def test_pick_plane_matches_oracle(simple):
    """The `write_pick` variant of the fragment shader (fs_main.wgsl:89-92): 64-bit pick word per pixel,
    bit for bit (integer packing of the hit coordinate, which is IEEE-exact on both sides)."""
    import torch

    from sub_volume_renderer_amd import _native as N

    spec = testing.synthetic_spec(64, 160, 96, threshold=0.4)
    scene = testing.build(spec)
    vol = scene.volume
    N.check(N.lib().svr_set_variant(vol.prepare(), 1 if simple else 0), "svr_set_variant")
    res = vol.render(scene.camera, scene.width, scene.height, pick=True)
    torch.cuda.synchronize()
    ref = lmip.render_spec(spec, pick_id=vol.id)
    got = res.pick.cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(got, ref.pick)
    hit = ref.flags == 2
    assert hit.sum() > 100 and np.all(got[~hit] == 0)
    assert np.all((got[hit] & np.uint64(0xFFFFF)) == np.uint64(vol.id))
    # a render without the pick plane leaves the other planes unchanged
    res2 = vol.render(scene.camera, scene.width, scene.height)
    torch.cuda.synchronize()
    assert torch.equal(res2.label, res.label) and torch.equal(res2.rgba, res.rgba)


@pytest.mark.parametrize("offset", [(0, 5, 0), (0, 0, 5), (6, 4, -9)])
def test_single_voxel_lands_where_the_camera_conventions_say(offset):
    """The hand-derived end-to-end known answer of tests/test_oracle_lmip.py on the device: where one bright voxel must
    appear on screen follows from the camera conventions alone (no oracle, no shared matrices)."""
    spec = testing.single_voxel_spec(offset)
    scene = testing.build(spec)
    prod, inst = testing.render_both(scene.volume, scene.camera, scene.width, scene.height)
    want_row, want_col = testing.expected_single_voxel_pixel(offset)
    for r in (prod, inst):
        rows, cols = np.nonzero(r.flags.cpu().numpy() == 2)
        assert rows.size >= 1
        assert abs(rows.mean() - want_row) <= 0.75 and abs(cols.mean() - want_col) <= 0.75, (rows, cols, want_row, want_col)
        assert np.all(r.label_numpy()[r.flags.cpu().numpy() == 2] == 7)
    check(scene)                                                  # and the oracle agrees pixel for pixel
