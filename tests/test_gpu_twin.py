"""GPU: the micro-block copy of the finest LOD's density ring (svr_lod_desc::blocked_twin).  A wave that would
touch many ring ROWS per gather takes the same texels from the copy instead: frames must be the oracle's, and
bit for bit those of a volume without the copy — for every ring storage, with windows that wrap around the ring,
after re-centring (both copies written by every upload), with the copy inside the one buffer resource and as a
resource of its own, in LMIP, full-length and MIP marches."""
import ctypes as C

import numpy as np
import pytest

from oracle import lmip
from sub_volume_renderer_amd import _native as N, testing

from test_gpu_render import check

pytestmark = pytest.mark.gpu

ALWAYS = 0x200          # svr_set_variant bit 9: every wave behaves as if its probe had found many rows per gather


def twin_batches(volume) -> int:
    """direct batches of the instrumented renders so far that gathered from a micro-block copy (svr_debug_timers [15])"""
    tm = (C.c_uint64 * 16)()
    N.check(N.lib().svr_debug_timers(volume._rings.handle, tm, 1), "svr_debug_timers")
    return int(tm[15])


def _spec(n, storage, cam, dtype=np.uint8, full=False, w=200, h=136):
    pairs = None
    if dtype != np.uint8:
        from sub_volume_renderer_amd import synth
        pairs = [(synth.volume(n, k, 4096)[0].astype(dtype) * (257 if dtype == np.uint16 else 1), synth.volume(n, k, 4096)[1]) for k in range(3)]
    spec = testing.synthetic_spec(n, w, h, inside=(cam == "K2"), full=full, pairs=pairs,
                                  chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)], ring_shapes=[(6, 5, 3), (10, 10, 3), (16, 16, 2)])
    spec.ring_storage = storage
    if dtype == np.uint16:
        spec.material.update(lmip_threshold=spec.material["lmip_threshold"] * 257.0, clim=(0.0, 65535.0))
    c = (n - 1) / 2.0
    if cam not in ("K1", "K2"):
        d = {"-x": (-1, 0.01, 0.02), "+y": (0.02, 1, 0.01), "-z": (0.01, 0.02, -1), "diag": (-1, -0.9, -0.8)}[cam]
        d = np.array(d, float) / np.linalg.norm(d)
        spec.cam_position = tuple(np.array([c, c, c]) + 1.7 * n * d)
        spec.cam_target = (c, c, c)
    return spec


@pytest.mark.parametrize("cam", ["K1", "K2", "-x", "+y", "diag"])
@pytest.mark.parametrize("storage,dtype", [("native", np.uint8), ("native", np.uint16), ("float32", np.uint8)],
                         ids=["u8rings", "u16rings", "f32rings"])
@pytest.mark.parametrize("full", [False, True], ids=["lmip", "full"])
def test_frames_with_the_micro_block_copy_are_the_oracles_and_those_without_it(cam, storage, dtype, full):
    spec = _spec(96, storage, cam, dtype, full)
    # a window that wraps around the ring on every axis, reached by re-centring twice (both copies rewritten in part)
    spec.centers = [((40.0, 44.0, 52.0), None), ((57.0, 49.0, 43.0), None)]
    ref = lmip.render_spec(spec)
    frames = {}
    for twin in (True, False):
        spec.blocked_twin = [twin, False, False]
        scene = testing.build(spec)
        assert scene.volume._rings.blocked_twin == [twin, False, False]
        N.check(N.lib().svr_set_variant(scene.volume.prepare(), ALWAYS), "svr_set_variant")
        twin_batches(scene.volume)
        res, _, rep = check(scene, ref=ref, want_hits=not full)
        used = twin_batches(scene.volume)
        assert (used > 0) == twin, (twin, used)                         # the copy really served gathers, and only where it exists
        frames[twin] = res
        # and by the wave's own probe (the default routing): the same frame
        N.check(N.lib().svr_set_variant(scene.volume.prepare(), 0), "svr_set_variant")
        check(scene, ref=ref, want_hits=not full)
    same = testing.planes_identical(frames[True], frames[False])
    assert same and all(same.values()), same


@pytest.mark.parametrize("which", ["instead_everywhere", "middle", "all", "fallback_only"])
@pytest.mark.parametrize("storage", ["native", "float32"])
def test_copies_of_several_lods_and_of_a_coarser_one_alone(which, storage):
    """`blocked_twin=True` (every LOD whose extents allow, instead of bricks), a copy of LOD 1 only, "all" (the finest LOD
    instead of bricks, the coarser ones for waves that stage none) and copies that are all of the second kind:
    frames stay the oracle's.  (float32 rings of this size stage no bricks from LOD 0 — their boxes do not fit — so the
    second kind is what serves it there.)"""
    spec = _spec(96, storage, "diag")
    spec.blocked_twin = {"instead_everywhere": True, "middle": [False, True, False], "all": "all", "fallback_only": [2, 2, "fallback"]}[which]
    scene = testing.build(spec)
    assert scene.volume._rings.blocked_twin == {"instead_everywhere": [1, 1, 1], "middle": [0, 1, 0], "all": [1, 2, 2], "fallback_only": [2, 2, 2]}[which]
    check(scene)                                                           # the waves' own probes first
    N.check(N.lib().svr_set_variant(scene.volume.prepare(), ALWAYS), "svr_set_variant")
    twin_batches(scene.volume)
    check(scene)
    used = twin_batches(scene.volume)
    assert used > 0 or which == "fallback_only"                            # (copies of the second kind serve only waves without bricks)
    if which in ("all", "fallback_only"):
        # no LOD of this volume may stage bricks (svr_set_variant bits 24-31: only LOD 6 may): every copy of the second kind serves
        N.check(N.lib().svr_set_variant(scene.volume.prepare(), ALWAYS | (0x40 << 24)), "svr_set_variant")
        twin_batches(scene.volume)
        check(scene)
        assert twin_batches(scene.volume) > 0


def test_mip_and_weighted_average_marches_read_the_copy_too():
    for mode, extra in (("mip", {}), ("weighted_average", {"weight_falloff": 0.4})):
        spec = _spec(96, "native", "K1")
        spec.material.update(render_mode=mode, **extra)
        scene = testing.build(spec)
        N.check(N.lib().svr_set_variant(scene.volume.prepare(), ALWAYS), "svr_set_variant")
        twin_batches(scene.volume)
        check(scene)
        assert twin_batches(scene.volume) > 0, mode


def test_extents_that_do_not_fit_are_refused_by_the_c_abi_and_skipped_by_auto():
    descs = (N.LodDesc * 1)()
    descs[0].ring_dims[:] = (36, 32, 32)                                  # x not a multiple of 8
    descs[0].density_storage = N.SVR_U8
    descs[0].blocked_twin = 1
    ctx = C.c_void_p()
    assert N.lib().svr_create(0, 1, descs, C.byref(ctx)) == -1            # SVR_ERR_INVALID
    assert b"multiples of (8, 4, 4)" in N.lib().svr_last_error()
    descs[0].blocked_twin = 2
    assert N.lib().svr_create(0, 1, descs, C.byref(ctx)) == -1
    spec = testing.synthetic_spec(60, 96, 64, chunk_shapes=[(5, 5, 12), (5, 5, 6), (5, 5, 3)], ring_shapes=[(4, 4, 3), (4, 4, 3), (3, 3, 3)])
    scene = testing.build(spec)
    assert scene.volume._rings.blocked_twin == [False, False, False]      # "auto": rings of 20 x 20 x 36 voxels keep rows only
    check(scene)
