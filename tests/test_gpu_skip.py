"""GPU: empty-space skipping of the LMIP march (macro-cell maxima maintained at upload time).  Skipping must
never change a pixel, a label or an executed-iteration count: scenes built to catch a skip that is too eager —
isolated bright voxels on cell corners and faces, thresholds exactly at their value, windows that wrap around the
ring, reloads that leave stale maxima behind — against the oracle, and skip on == skip off."""
import numpy as np
import pytest

from oracle import lmip
from sub_volume_renderer_amd import _native as N, testing

from test_gpu_render import check

pytestmark = pytest.mark.gpu


def _sparse_pairs(n, seed, value=200, count=40, dtype=np.uint8, noise=12):
    """Three LODs (2x mean / max pooled) of a dark volume with a few isolated bright voxels, many of them on the
    corners, edges and faces of the 8^3 macro cells."""
    rng = np.random.default_rng(seed)
    d0 = rng.integers(0, noise, (n, n, n)).astype(dtype)
    l0 = np.zeros((n, n, n), np.uint32)
    for k in range(count):
        p = rng.integers(1, n - 2, 3)
        if k % 2 == 0:
            p = (p // 8) * 8 + rng.choice([-1, 7], 3)         # 2^3 blobs straddling a cell corner (8 cells)
        elif k % 3 == 0:
            p[k % 3] = (p[k % 3] // 8) * 8 - 1                 # ... or a cell face
        p = np.clip(p, 0, n - 2)
        sl = tuple(slice(int(v), int(v) + 2) for v in p)
        d0[sl] = value
        l0[sl] = 1 + k
    pairs = [(d0, l0)]
    d, l = d0, l0
    for _ in range(2):
        m = d.shape[0] // 2
        d = (d.reshape(m, 2, m, 2, m, 2).astype(np.uint32).sum(axis=(1, 3, 5)) // 8).astype(dtype)
        l = l.reshape(m, 2, m, 2, m, 2).max(axis=(1, 3, 5))
        pairs.append((d, l))
    return pairs


def _scene(n, pairs, threshold, cam, storage="native", w=192, h=128):
    spec = testing.synthetic_spec(n, w, h, inside=(cam == "K2"), pairs=pairs, chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)],
                                  ring_shapes=[(6, 6, 3), (12, 12, 3), (16, 16, 2)])
    spec.ring_storage = storage
    spec.material.update(lmip_threshold=threshold, clim=(0.0, 255.0))
    c = (n - 1) / 2.0
    if cam not in ("K1", "K2"):
        d = {"-x": (-1, 0.01, 0.02), "+y": (0.02, 1, 0.01), "-z": (0.01, 0.02, -1), "diag": (-1, -1, -1)}[cam]
        d = np.array(d, float) / np.linalg.norm(d)
        spec.cam_position = tuple(np.array([c, c, c]) + 1.7 * n * d)
        spec.cam_target = (c, c, c)
    return spec


@pytest.mark.parametrize("cam", ["K1", "K2", "-x", "+y", "-z", "diag"])
@pytest.mark.parametrize("threshold", [200.0, 199.5, 200.5, 5.0])
def test_sparse_bright_voxels_on_cell_borders(cam, threshold):
    """threshold == the bright value (>= must hit), just below, just above (nothing reaches it: all rays run to
    their end through skipped space), and inside the background noise (every cell is 'occupied')."""
    spec = _scene(128, _sparse_pairs(128, 1), threshold, cam)
    scene = testing.build(spec)
    res, ref, rep = check(scene, want_hits=False)
    if threshold > 200.0 and threshold < 255.0:
        assert rep["n_hit"] == 0 and rep["n_miss"] > 1000       # nothing reaches it (single bright voxels may also be
    elif threshold == 5.0:                                      # stepped over at 0.8 voxels per sample: no hit is owed)
        assert rep["n_hit"] > 1000


@pytest.mark.parametrize("storage,dtype,scale", [("native", np.uint8, 1), ("float32", np.uint8, 1), ("native", np.uint16, 257)])
def test_skip_on_equals_skip_off_and_really_skips(storage, dtype, scale):
    import torch

    pairs = [(d.astype(dtype) * scale, l) for d, l in _sparse_pairs(128, 2)]
    spec = _scene(128, pairs, 150.0 * scale, "K1", storage)
    spec.material.update(clim=(0.0, 255.0 * scale))
    scene = testing.build(spec)
    vol = scene.volume
    _, on = testing.render_both(vol, scene.camera, spec.width, spec.height)        # production == instrumented, skipping on
    on = {k: getattr(on, k).clone() for k in ("rgba", "depth", "label", "flags", "steps")}
    import ctypes as C

    census = (C.c_uint32 * 8)()
    N.check(N.lib().svr_debug_counters(vol._rings.handle, census, 1), "svr_debug_counters")
    skipped = census[7]                                     # [7]: batches skipped as empty space
    N.check(N.lib().svr_set_variant(vol._rings.handle, 8), "svr_set_variant")      # bit 3: no skipping
    for off in testing.render_both(vol, scene.camera, spec.width, spec.height):    # ... and skipping off, both kernels
        for k, v in on.items():
            if getattr(off, k) is not None:
                assert torch.equal(getattr(off, k), v), k
    N.check(N.lib().svr_debug_counters(vol._rings.handle, census, 1), "svr_debug_counters")
    assert skipped > 1000 and census[7] == 0                # the default took skips, the A/B run none
    check(scene, want_hits=False)                           # and both are the oracle's frame


@pytest.mark.parametrize("cam", ["K1", "K2", "diag"])
@pytest.mark.parametrize("storage,dtype,scale", [("native", np.uint8, 1), ("float32", np.uint8, 1), ("native", np.uint16, 257)])
def test_mip_mode_skips_what_cannot_beat_the_running_maximum(storage, dtype, scale, cam):
    """MIP (no fall-off, no sample limit): once a ray holds a maximum, blocks whose maximum does not exceed it are
    passed like empty ones (raycast.wgsl:50 replaces only on a strict >).  Bright blobs of EQUAL value behind one
    another: the first one must stay the hit."""
    import ctypes as C

    import torch

    pairs = [(d.astype(dtype) * scale, l) for d, l in _sparse_pairs(128, 3, count=16, noise=1)]     # background 0: only the blobs beat a maximum
    spec = _scene(128, pairs, 150.0 * scale, cam, storage)
    spec.material.update(clim=(0.0, 255.0 * scale), render_mode="mip")
    scene = testing.build(spec)
    vol = scene.volume
    _, on = testing.render_both(vol, scene.camera, spec.width, spec.height)
    on = {k: getattr(on, k).clone() for k in ("rgba", "depth", "label", "flags", "steps")}
    census = (C.c_uint32 * 8)()
    N.check(N.lib().svr_debug_counters(vol._rings.handle, census, 1), "svr_debug_counters")
    skipped = census[7]
    N.check(N.lib().svr_set_variant(vol._rings.handle, 8), "svr_set_variant")      # bit 3: no skipping
    for off in testing.render_both(vol, scene.camera, spec.width, spec.height):
        for k, v in on.items():
            if getattr(off, k) is not None:
                assert torch.equal(getattr(off, k), v), k
    assert skipped > 50
    N.check(N.lib().svr_set_variant(vol._rings.handle, 0), "svr_set_variant")
    _, ref, rep = check(scene)
    assert rep["n_miss"] == 0                                # every fragment yields its maximum
    if cam != "K2":                                          # (the inside camera sees few of the 16 blobs)
        assert len(np.unique(ref.label[ref.flags == 2])) >= 3


def test_stale_maxima_after_window_moves_stay_conservative():
    """Windows move (ring slots are rewritten, some cells keep slots of chunks that left the ROI), blocking and
    asynchronous; after every move the frame equals the oracle's."""
    import torch

    pairs = _sparse_pairs(128, 3, count=120)
    spec = _scene(128, pairs, 180.0, "K2")
    scene = testing.build(spec)
    vol = scene.volume
    orac = lmip.oracle_volume(spec)
    eye = np.array(spec.cam_position)
    d = np.array(spec.cam_target) - eye
    d = d / np.linalg.norm(d)
    for k in range(1, 9):
        p = eye + d * 7.0 * k
        spec.cam_position, spec.cam_target = tuple(p), tuple(p + d)
        vol.center_on_position(tuple(p), asynchronous=bool(k & 1))
        vol.poll_uploads(wait=True)
        orac.center_on_position(tuple(p))
        ref = lmip.render(lmip.rings_of(orac), spec.matrices(), orac.volume_dimensions_shader, spec.material, spec.width, spec.height)
        testing.hold_both_to(ref, vol, spec.camera(), spec.width, spec.height)      # production and instrumented kernels


def test_cleared_and_partially_loaded_rings():
    """A LOD whose ring was never loaded (ROI None) and one loaded only in a corner: zeros everywhere else."""
    pairs = _sparse_pairs(64, 4, count=30)
    spec = testing.synthetic_spec(64, 128, 96, pairs=pairs, chunk_shapes=[(8, 8, 16), (4, 4, 16), (2, 2, 16)],
                                  ring_shapes=[(4, 4, 2), (8, 8, 2), (8, 8, 1)])
    spec.material.update(lmip_threshold=150.0, clim=(0.0, 255.0))
    spec.centers = [((10.0, 12.0, 9.0), None)]              # window hanging over the volume's corner
    check(testing.build(spec), want_hits=False)
