"""GPU parity on inputs that leave the fast paths: LOD scales that are not powers of two (general batches
only), rings whose x extent is not a multiple of 16 (no bricks), anisotropic volumes, tiny and ragged frames,
degenerate material values (the integer pre-check of the byte rings must agree with the f32 comparison)."""
import numpy as np
import pytest

from sub_volume_renderer_amd import _native as N, testing

from test_gpu_render import check

pytestmark = pytest.mark.gpu


def _random_pairs(shapes, seed, smooth=True):
    rng = np.random.default_rng(seed)
    pairs = []
    for shp in shapes:
        zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=np.float32) / s for s in shp], indexing="ij")
        base = 110 + 90 * np.cos(7 * xx + 3 * yy) * np.cos(5 * zz - 2 * yy) if smooth else 0
        d = np.clip(base + rng.integers(0, 40, shp), 0, 255).astype(np.uint8)
        lab = rng.integers(0, 5000, shp).astype(np.uint32)
        pairs.append((d, lab))
    return pairs


@pytest.mark.parametrize("variant", [0x000, 0x200, 0x001], ids=["auto", "brick", "simple"])
def test_lod_scales_not_powers_of_two(variant):
    """96 -> 48 -> 32 voxels: scale factors 1, 1/2, 1/3.  The 1/3 LOD cannot use the fused size*scale factor."""
    pairs = _random_pairs([(96,) * 3, (48,) * 3, (32,) * 3], 1)
    spec = testing.synthetic_spec(96, 150, 100, pairs=pairs, threshold=0.55,
                                  chunk_shapes=[(8, 8, 16), (4, 4, 16), (4, 4, 8)],
                                  ring_shapes=[(5, 5, 2), (8, 8, 2), (8, 8, 4)],
                                  sizes=[(30, 30, 30), (60, 60, 60), (96, 96, 96)])
    scene = testing.build(spec)
    assert [tuple(np.round(b.scale_factor, 4)) for b in scene.volume.wrapping_buffers][2] == (0.3333, 0.3333, 0.3333)
    N.check(N.lib().svr_set_variant(scene.volume.prepare(), variant), "svr_set_variant")
    _, ref, rep = check(scene)
    assert len(np.unique(ref.label[ref.flags == 2])) > 50


@pytest.mark.parametrize("variant", [0x000, 0x200])
def test_ring_rows_not_multiple_of_16_and_anisotropic_volume(variant):
    """(a0, a1, a2) = (40, 60, 84) voxels, chunks of 12 along x: ring rows of 36 / 60 bytes, so no LOD can
    stage 16-byte brick groups; every sample goes through the gather path, whatever the variant asks for."""
    pairs = _random_pairs([(40, 60, 84), (20, 30, 42)], 2)
    spec = testing.synthetic_spec(84, 140, 90, pairs=pairs, threshold=0.5,
                                  chunk_shapes=[(5, 6, 12), (5, 6, 6)], ring_shapes=[(4, 5, 3), (4, 5, 7)],
                                  sizes=[(15, 24, 24), (40, 60, 84)])
    c = (41.5, 29.5, 19.5)                                     # shader xyz = (a2, a1, a0) centre
    spec.centers = [(c, spec.centers[0][1])]
    spec.cam_target = c
    spec.cam_position = (c[0] - 120.0, c[1] + 55.0, c[2] + 70.0)
    scene = testing.build(spec)
    N.check(N.lib().svr_set_variant(scene.volume.prepare(), variant), "svr_set_variant")
    check(scene)


@pytest.mark.parametrize("wh", [(1, 1), (7, 3), (13, 9), (65, 17)])
def test_tiny_and_ragged_frames(wh):
    spec = testing.synthetic_spec(64, wh[0], wh[1], threshold=0.45)
    check(testing.build(spec), want_hits=False)


@pytest.mark.parametrize("material", [
    dict(lmip_threshold=0.0), dict(lmip_threshold=-3.0), dict(lmip_threshold=float("nan")),
    dict(lmip_threshold=255.0), dict(lmip_threshold=255.5), dict(lmip_threshold=254.000001),
    dict(lmip_threshold=100.0, lmip_max_samples=0), dict(lmip_threshold=100.0, lmip_max_samples=1),
    dict(lmip_threshold=100.0, lmip_fall_off=1.5), dict(lmip_threshold=100.0, lmip_fall_off=0.0),
    dict(lmip_threshold=140.0, gamma=2.2, clim=(20.0, 200.0), opacity=0.25, fog_density=3.0),
], ids=lambda m: ",".join(f"{k}={v}" for k, v in m.items()))
@pytest.mark.parametrize("storage", ["native", "float32"])
def test_degenerate_material_values(material, storage):
    spec = testing.synthetic_spec(64, 120, 80)
    spec.material.update(material)
    spec.ring_storage = storage
    scene = testing.build(spec)
    res, ref, rep = check(scene, want_hits=False)
    t = material.get("lmip_threshold")
    if t is not None and t <= 0:
        assert rep["n_miss"] == 0 and rep["n_hit"] > 0          # the very first sample is "significant"
    if t is not None and (t != t or t > 255):
        assert rep["n_hit"] == 0


def test_max_byte_value_reaches_threshold_255():
    """A voxel of 255 must be found with lmip_threshold = 255.0 (byte compare against ceil(threshold))."""
    d = np.zeros((32, 32, 32), np.uint8)
    d[10:20, 10:20, 10:20] = 255
    lab = np.full((32, 32, 32), 7, np.uint32)
    spec = testing.synthetic_spec(32, 96, 64, pairs=[(d, lab)], chunk_shapes=[(8, 8, 16)], ring_shapes=[(4, 4, 2)])
    spec.material.update(lmip_threshold=255.0)
    _, ref, rep = check(testing.build(spec))
    assert rep["n_hit"] > 50


def test_c_abi_rejects_bad_arguments_with_messages():
    """Every entry point validates on the host and returns an error code + message; nothing reaches a kernel."""
    import ctypes as C

    lib = N.lib()
    descs = (N.LodDesc * 1)()
    descs[0].ring_dims[:] = (32, 16, 16)
    descs[0].density_storage = 0                                   # SVR_U8
    ctx = C.c_void_p()
    assert lib.svr_create(0, 1, descs, C.byref(ctx)) == 0

    def fails(rc, needle):
        assert rc != 0
        msg = lib.svr_last_error().decode()
        assert needle in msg, msg

    I3, L3 = C.c_int32 * 3, C.c_int64 * 3
    st = N.LodState()
    st.offset[:] = (0, 0, 0); st.shape[:] = (64, 16, 16); st.scale[:] = (1.0, 1.0, 1.0)
    fails(lib.svr_set_lod_state(ctx, 0, C.byref(st)), "larger than the ring")
    st.shape[:] = (16, 16, 16)
    fails(lib.svr_set_lod_state(ctx, 3, C.byref(st)), "out of range")
    assert lib.svr_set_lod_state(ctx, 0, C.byref(st)) == 0
    data = np.zeros((16, 16, 64), np.uint8)
    fails(lib.svr_upload_region(ctx, 0, I3(0, 0, 0), I3(64, 16, 16), C.c_void_p(data.ctypes.data), 0,
                                L3(1, 64, 1024), None, 0, L3(0, 0, 0)), "svr_upload_region")
    f32 = np.zeros((16, 16, 32), np.float32)
    fails(lib.svr_upload_region(ctx, 0, I3(0, 0, 0), I3(32, 16, 16), C.c_void_p(f32.ctypes.data), 8,
                                L3(4, 128, 2048), None, 0, L3(0, 0, 0)), "integer density storage")
    cam, fr, ob = N.Camera(), N.Frame(), N.Outputs()
    fr.frame_w, fr.frame_h, fr.out_w, fr.out_h, fr.band_h, fr.band_pitch = 8, 8, 8, 8, 8, 8
    fails(lib.svr_render(ctx, C.byref(cam), C.byref(fr), C.byref(ob), None), "null argument")
    import torch

    out = torch.zeros((8, 8, 4), dtype=torch.float32, device="cuda")
    ob.rgba = out.data_ptr()
    fails(lib.svr_render(ctx, C.byref(cam), C.byref(fr), C.byref(ob), None), "svr_set_material")
    m = N.Material()
    fails(lib.svr_set_material(ctx, C.byref(m)), "at least one colour")
    assert lib.svr_destroy(ctx) == 0


@pytest.mark.parametrize("W,H,world,tiling", [(1920, 1080, 8, "2x4"), (66, 45, 6, "3x2"), (64, 48, 4, "rows"), (65, 45, 3, "rows")])
def test_untile_kernels_equal_the_host_permutation(W, H, world, tiling):
    """svr_untile_grid / svr_untile_stripes on gathered buffers of every plane type (RGBA f32, depth f32, label
    i32, flags u8) against TiledFrame's own CPU un-tiling."""
    import torch

    from sub_volume_renderer_amd import WrappingBuffer
    from sub_volume_renderer_amd.distributed import TiledFrame

    class Holder:                                           # un-tiling only needs a device context
        def __init__(self):
            z = np.zeros((8, 8, 8), np.uint8)
            self._rings = WrappingBuffer(z, z, (2, 2, 2), (4, 4, 4)).rings
            self._rings.handle

    holder = Holder()
    tf = TiledFrame(W, H, 0, world, 8, tiling=tiling)
    g = torch.Generator().manual_seed(W + world)
    for dtype, tail in ((torch.float32, (4,)), (torch.float32, ()), (torch.int32, ()), (torch.uint8, ())):
        shape = (world, tf.rows_per_rank, tf.cols_per_rank, *tail)
        gathered = (torch.rand(shape, generator=g) * 200).to(dtype)
        want = tf.untile(gathered, torch.zeros((H, W, *tail), dtype=dtype))
        got = tf.untile(gathered.cuda(), torch.zeros((H, W, *tail), dtype=dtype, device="cuda"), holder)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want), (dtype, tail)


def test_draws_on_many_short_lived_streams_and_rejected_experiment_bits():
    """A caller that creates a HIP stream per frame and destroys it afterwards: the library keeps per-stream state (the
    cost-sorted placement table, render marks) for 64 streams and recycles the oldest entry — through the entry's own
    events, never through the stream handle, which is gone by then.  Every frame equals the first."""
    import ctypes as C

    import torch

    scene = testing.build(testing.synthetic_spec(64, 160, 96, threshold=0.4))
    vol, cam = scene.volume, scene.camera
    first = vol.render(cam, 160, 96)
    torch.cuda.synchronize()
    want = first.rgba.clone()
    hip = C.CDLL("libamdhip64.so")                             # the runtime torch already loaded
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    for k in range(150):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        if k % 3 == 0:                                          # a new camera now and then: the table is rebuilt on that stream
            scene.spec.cam_position = tuple(np.array(scene.spec.cam_position) + (0.5 if k % 6 == 0 else -0.5))
            cam_k = scene.spec.camera()
        r = vol.render(cam_k, 160, 96, stream=s.value)
        r = vol.render(cam, 160, 96, stream=s.value)
        assert hip.hipStreamSynchronize(s) == 0
        assert torch.equal(r.rgba, want), k
        assert hip.hipStreamDestroy(s) == 0
    # the shipped library knows no timing-experiment bits (they render wrong pixels): rejected, state unchanged
    for bits in (1 << 11, 1 << 12, 3 << 11):
        with pytest.raises(ValueError, match="SVR_EXPERIMENTS"):
            N.check(N.lib().svr_set_variant(vol.prepare(), bits), "svr_set_variant")
    r = vol.render(cam, 160, 96)
    torch.cuda.synchronize()
    assert torch.equal(r.rgba, want)
