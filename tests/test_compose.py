"""Display-side compose: numpy restatement properties on CPU, HIP kernel vs the
restatement on the GPU.  Blending / encode are pygfx behaviour restated: parity unpinned."""
import numpy as np
import pytest

from oracle import compose_oracle, lmip
from sub_volume_renderer_amd import testing


def test_oracle_known_values():
    rgba = np.zeros((2, 3, 4), np.float32)
    rgba[0, 0] = (1.0, 0.5, 0.0, 1.0)          # opaque fragment
    rgba[0, 1] = (1.0, 1.0, 1.0, 0.5)          # half-transparent white
    flags = np.array([[2, 1, 0], [0, 0, 0]], np.uint8)
    q, _ = compose_oracle.compose(rgba, flags=flags, bg_bottom=(0, 0, 0, 1), bg_top=(0, 0, 0, 1))
    assert tuple(q[0, 0]) == (255, 188, 0, 255)                      # sRGB(0.5) = 0.7354 -> 188
    assert tuple(q[0, 1]) == (188, 188, 188, 255)                    # 0.5 over black, alpha 0.5 + 1*0.5
    assert tuple(q[0, 2]) == (0, 0, 0, 255) and tuple(q[1, 1]) == (0, 0, 0, 255)   # discarded: background
    # gradient: top row nearer bg_top
    q2, _ = compose_oracle.compose(np.zeros((4, 1, 4), np.float32), flags=np.zeros((4, 1), np.uint8),
                                   bg_bottom=(1, 1, 1, 1), bg_top=(0, 0, 0, 1), srgb=False)
    assert list(q2[:, 0, 0]) == [32, 96, 159, 223]
    # depth test keeps what is nearer
    z = np.array([[0.2, 0.9]], np.float32)
    d = np.array([[0.5, 0.5]], np.float32)
    px = np.ones((1, 2, 4), np.float32)
    q3, z3 = compose_oracle.compose(px, depth=d, zbuf=z, srgb=False)
    assert tuple(q3[0, 0]) == (0, 0, 0, 255) and tuple(q3[0, 1]) == (255, 255, 255, 255)
    np.testing.assert_array_equal(z3, np.array([[0.2, 0.5]], np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("with_depth", [False, True])
def test_compose_kernel_matches_restatement(with_depth):
    import torch

    from sub_volume_renderer_amd.compose import compose

    spec = testing.synthetic_spec(64, 150, 90, threshold=0.4)
    spec.material["opacity"] = 0.8
    scene = testing.build(spec)
    res = scene.volume.render(scene.camera, scene.width, scene.height)
    torch.cuda.synchronize()
    bottom, top = (100 / 255, 100 / 255, 100 / 255, 1.0), (169 / 255, 167 / 255, 168 / 255, 1.0)   # tests/conftest.py:18-19
    rng = np.random.default_rng(5)
    z0 = rng.random((scene.height, scene.width), dtype=np.float32) if with_depth else None
    zdev = torch.from_numpy(z0.copy()).to(res.rgba.device) if with_depth else None
    got = compose(scene.volume, res, background=(bottom, top), depth_buffer=zdev)
    torch.cuda.synchronize()
    want, zwant = compose_oracle.compose(res.rgba.cpu().numpy(), res.depth.cpu().numpy(), res.flags.cpu().numpy(),
                                         bottom, top, zbuf=z0)
    diff = np.abs(got.cpu().numpy().astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1, diff.max()                                 # powf: 1 LSB at most
    assert (diff > 0).mean() < 0.01
    assert (res.flags == 2).sum().item() > 100
    if with_depth:
        np.testing.assert_array_equal(zdev.cpu().numpy(), zwant)
        assert 0 < (zwant != z0).sum() < z0.size                       # some fragments passed, some did not
    # the render itself agrees with the oracle (so the composed image is the oracle's, composed)
    ref = lmip.render_spec(spec)
    assert np.array_equal(res.flags.cpu().numpy(), ref.flags)
