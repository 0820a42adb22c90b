"""GPU: the material features beside the plain LMIP uniforms — clipping planes (fs_main.wgsl:8), the MIP
and weighted-average render modes (FUTURE.md:97-120) — and uint16 ring storage, each against the CPU oracle on identical inputs."""
import numpy as np
import pytest

from oracle import lmip
from sub_volume_renderer_amd import testing

from test_gpu_render import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["ANY", "ALL"])
@pytest.mark.parametrize("inside", [False, True], ids=["K1", "K2"])
def test_clipping_planes_match_oracle(mode, inside):
    spec = testing.synthetic_spec(64, 160, 96, inside=inside, threshold=0.4)
    # two planes through the volume's centre (31.5, 31.5, 31.5): each cuts the box faces — where rays exit — in half
    spec.material.update(clipping_planes=[(1.0, 0.2, 0.0, 37.8), (0.0, -1.0, 0.3, -22.05)], clipping_mode=mode)
    scene = testing.build(spec)
    _, ref, rep = check(scene)
    plain = lmip.render_spec(testing.synthetic_spec(64, 160, 96, inside=inside, threshold=0.4))
    if mode == "ANY":
        assert (ref.flags != plain.flags).sum() > 50          # the planes really remove rays
    else:
        assert (ref.flags != plain.flags).sum() < (ref.flags != 0).sum()      # ALL clips no more than ANY would
    # planes can be changed and removed again between draws
    scene.volume.material.clipping_planes = []
    scene.spec.material.update(clipping_planes=[])
    _, ref2, _ = check(scene)
    np.testing.assert_array_equal(ref2.flags, plain.flags)


def test_eight_clipping_planes_and_limit():
    spec = testing.synthetic_spec(64, 96, 64, threshold=0.4)
    planes = [(np.cos(k), np.sin(k), 0.1 * k, -40.0 + 3 * k) for k in range(8)]
    spec.material.update(clipping_planes=planes, clipping_mode="ANY")
    check(testing.build(spec), want_hits=False)
    with pytest.raises(ValueError):
        testing.build(spec).volume.material.clipping_planes = planes + [(1, 0, 0, 0)]


@pytest.mark.parametrize("storage", ["native", "float32"])
@pytest.mark.parametrize("inside", [False, True], ids=["K1", "K2"])
def test_mip_mode_matches_oracle(inside, storage):
    spec = testing.synthetic_spec(64, 160, 96, inside=inside)
    spec.ring_storage = storage
    spec.material.update(render_mode="mip")
    scene = testing.build(spec)
    _, ref, rep = check(scene)
    assert rep["n_miss"] == 0 and rep["n_hit"] > 1000
    # back to LMIP: the lmip_* properties were left alone
    scene.volume.material.render_mode = "lmip"
    scene.spec.material.update(render_mode="lmip")
    _, ref2, rep2 = check(scene)
    assert rep2["n_miss"] > 0 and rep2["total_steps"] < rep["total_steps"]


@pytest.mark.parametrize("full", [False, True], ids=["lmip", "full"])
@pytest.mark.parametrize("inside", [False, True], ids=["K1", "K2"])
def test_uint16_sources_get_uint16_rings_with_identical_results(inside, full):
    """Common microscopy dtype: u16 values are exact in f32, so byte-for-byte the same frame as the reference's
    r32float layout at half the memory."""
    from sub_volume_renderer_amd import synth

    pairs = []
    for k in range(3):
        d, l = synth.volume(64, k)
        pairs.append((d.astype(np.uint16) * 257, l))           # 0 .. 65535
    spec = testing.synthetic_spec(64, 160, 96, inside=inside, full=full, pairs=pairs)
    if not full:
        spec.material.update(lmip_threshold=0.5 * 65535, clim=(0.0, 65535.0))
    scene = testing.build(spec)
    assert scene.volume._rings.density_storage == "uint16"
    res, ref, rep = check(scene, want_hits=not full)
    spec.ring_storage = "float32"
    scene32 = testing.build(spec)
    assert scene32.volume._rings.density_storage == "float32"
    import torch

    for r32 in testing.render_both(scene32.volume, scene32.camera, scene32.width, scene32.height):
        for plane in ("rgba", "depth", "label", "flags", "steps"):
            if getattr(r32, plane) is not None:
                assert torch.equal(getattr(res, plane), getattr(r32, plane)), plane


def test_uint16_threshold_edges():
    """Integer pre-check against ceil(threshold): thresholds at, just above and beyond the value range."""
    d = np.full((16, 16, 16), 65535, np.uint16)
    d[::2] = 40000
    seg = np.ones(d.shape, np.uint32)
    for thr in (65535.0, 65534.5, 65535.5, 40000.0, 40000.25, 0.0, -1.0, float("inf"), float("nan")):
        spec = testing.synthetic_spec(16, 48, 32, pairs=[(d, seg)], chunk_shapes=[(4, 4, 4)], ring_shapes=[(4, 4, 4)])
        spec.material.update(lmip_threshold=thr, clim=(0.0, 65535.0))
        spec.centers = [((7.5, 7.5, 7.5), [(16, 16, 16)])]
        check(testing.build(spec), want_hits=False)


@pytest.mark.parametrize("inside", [False, True], ids=["K1", "K2"])
def test_volume_without_segmentation_renders_every_hit_with_label_zero(inside):
    """FUTURE.md:178-193 ("make this optional"): pairs of (density, None) — no label rings are allocated, every hit
    carries label 0 and the hue of colors[0], the density side is untouched."""
    import ctypes as C

    import torch

    from sub_volume_renderer_amd import _native as N

    spec = testing.synthetic_spec(64, 160, 96, inside=inside)
    labelled = lmip.render_spec(spec)
    spec.pairs = [(d, None) for d, _ in spec.pairs]
    scene = testing.build(spec)
    res, ref, rep = check(scene)
    assert int(res.label.abs().sum()) == 0 and rep["n_hit"] > 100
    np.testing.assert_array_equal(ref.flags, labelled.flags)          # same hits, same steps: only the hue changes
    np.testing.assert_array_equal(ref.steps, labelled.steps)
    np.testing.assert_array_equal(ref.depth, labelled.depth)
    lab_ptr = C.c_void_p(1)
    N.check(N.lib().svr_lod_device_ptrs(scene.volume._rings.handle, 0, None, C.byref(lab_ptr)), "ptrs")
    assert not lab_ptr.value                                           # no label ring exists
    b = scene.volume.wrapping_buffers[0]
    assert int(b.segmentations_texture.data.sum()) == 0
    # windows move, blocking and asynchronous, as for labelled volumes
    orac = lmip.oracle_volume(spec)
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    for k in (1, 2):
        p = eye + d * 9.0 * k
        scene.volume.center_on_position(tuple(p), asynchronous=(k == 2))
        scene.volume.poll_uploads(wait=True)
        orac.center_on_position(tuple(p))
    for bb, ob in zip(scene.volume.wrapping_buffers, orac.wrapping_buffers):
        np.testing.assert_array_equal(bb.texture.data, ob.texture)
    with pytest.raises(ValueError):                                    # a label source for a context without label rings
        scene.volume.wrapping_buffers[0]._upload(__import__("sub_volume_renderer_amd").Roi((0, 0, 0), (8, 8, 16)),
                                                 np.zeros((8, 8, 16), np.uint8), np.zeros((8, 8, 16), np.uint32))
    torch.cuda.synchronize()


# ---- weighted-average mode (SVR_MODE_WEIGHTED_AVERAGE, include/svr.h) ---------------------------------------------
@pytest.mark.parametrize("falloff", [0.0, 0.5, 4.0])
@pytest.mark.parametrize("storage", ["native", "float32"])
@pytest.mark.parametrize("inside", [False, True], ids=["K1", "K2"])
def test_weighted_average_mode_matches_oracle(inside, storage, falloff):
    spec = testing.synthetic_spec(64, 160, 96, inside=inside)
    spec.ring_storage = storage
    spec.material.update(render_mode="weighted_average", weight_falloff=falloff)
    scene = testing.build(spec)
    assert scene.volume.material.render_mode == "weighted_average" and scene.volume.material.weight_falloff == falloff
    _, ref, rep = check(scene)
    assert rep["n_hit"] > 1000
    # the mode is a material switch: back to LMIP on the same volume, with the lmip_* properties as they were
    scene.volume.material.render_mode = "lmip"
    scene.spec.material.update(render_mode="lmip")
    _, ref2, rep2 = check(scene)
    assert not np.array_equal(ref2.rgba, ref.rgba)


def test_weighted_average_regions_label_less_volume_and_uint16():
    from sub_volume_renderer_amd import FrameRegion, synth

    pairs = []
    for k in range(3):
        d, _ = synth.volume(64, k)
        pairs.append((d.astype(np.uint16) * 257, None))
    spec = testing.synthetic_spec(64, 150, 90, pairs=pairs)
    spec.material.update(render_mode="weighted_average", weight_falloff=1.0, clim=(0.0, 65535.0))
    scene = testing.build(spec)
    assert scene.volume._rings.density_storage == "uint16"
    res, ref, rep = check(scene)
    assert np.all(ref.label == 0)
    for region in (FrameRegion.tile(13, 7, 100, 50), FrameRegion.stripes(150, 90, 1, 3, 8)):
        check(scene, region=region, want_hits=False)
    # the one-fetch-per-step kernel (march_wavg: the fallback for rings the span march cannot address) is the same frame
    from sub_volume_renderer_amd import _native as N

    N.check(N.lib().svr_set_variant(scene.volume.prepare(), 1), "svr_set_variant")
    check(scene)
    check(scene, region=FrameRegion.stripes(150, 90, 2, 3, 8), want_hits=False)
