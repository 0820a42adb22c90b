"""GPU pyramid builder (svr_pool2x) against the restated pooling rules of the reference's offline scripts."""
import numpy as np
import pytest

from oracle import pool_oracle
from sub_volume_renderer_amd import synth


def test_pool_oracle_matches_reference_numpy_expression():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 255, (8, 12, 20), dtype=np.uint8)
    ref = a.reshape(4, 2, 6, 2, 10, 2).mean(axis=(1, 3, 5))          # create_mouse_multiscale.py:44-54
    np.testing.assert_array_equal(pool_oracle.mean_u8(a), np.floor(ref).astype(np.uint8))
    f = rng.random((8, 12, 20)).astype(np.float32)
    np.testing.assert_allclose(pool_oracle.mean_f32(f), f.reshape(4, 2, 6, 2, 10, 2).mean(axis=(1, 3, 5)), rtol=1e-6)
    lab = rng.integers(0, 2 ** 32 - 1, (8, 12, 20), dtype=np.uint32)
    ref = np.zeros((4, 6, 10), np.uint32)
    for i in range(4):
        for j in range(6):
            for k in range(10):
                ref[i, j, k] = lab[2 * i:2 * i + 2, 2 * j:2 * j + 2, 2 * k:2 * k + 2].max()   # create_platynereis:113-130
    np.testing.assert_array_equal(pool_oracle.max_u32(lab), ref)
    # and the synthetic volumes' LODs follow the same rules
    d0, l0 = synth.volume(32, 0)
    d1, l1 = synth.volume(32, 1)
    np.testing.assert_array_equal(pool_oracle.mean_u8(d0), d1)
    np.testing.assert_array_equal(pool_oracle.max_u32(l0), l1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 12, 20), (64, 64, 64), (2, 2, 2), (6, 10, 30)])
def test_pool2x_bit_exact(shape):
    import torch

    from sub_volume_renderer_amd.pyramid import pool2x

    rng = np.random.default_rng(3)
    a = rng.integers(0, 255, shape, dtype=np.uint8)
    np.testing.assert_array_equal(pool2x(torch.from_numpy(a).cuda(), "mean").cpu().numpy(), pool_oracle.mean_u8(a))
    f = (rng.random(shape) * 300 - 20).astype(np.float32)
    np.testing.assert_array_equal(pool2x(torch.from_numpy(f).cuda(), "mean").cpu().numpy(), pool_oracle.mean_f32(f))
    lab = rng.integers(0, 2 ** 32 - 1, shape, dtype=np.uint32)
    got = pool2x(torch.from_numpy(lab.view(np.int32)).cuda(), "max").cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, pool_oracle.max_u32(lab))
    with pytest.raises(ValueError):
        pool2x(torch.zeros((3, 4, 4), dtype=torch.uint8, device="cuda"), "mean")


@pytest.mark.gpu
def test_build_pyramid_equals_synthetic_lods():
    import torch

    from sub_volume_renderer_amd.pyramid import build_pyramid

    n = 256
    d0, l0 = synth.volume(n, 0, xp=torch, device=torch.device("cuda", 0))
    pairs = build_pyramid(d0, l0, 3)
    for k in (1, 2):
        dk, lk = synth.volume(n, k, xp=torch, device=torch.device("cuda", 0))
        assert torch.equal(pairs[k][0], dk) and torch.equal(pairs[k][1], lk)


@pytest.mark.gpu
def test_pool_kernels_stream_at_a_sane_fraction_of_hbm_rate(capsys):
    """One 2x pooling step of a 1024^3 level reads it once and writes an eighth: pure HBM streaming.  The rate is
    printed (tools/exp_kernels.py records it under profiles/); the floor asserted here only catches a
    pathological kernel (uncoalesced rows), not a slow box."""
    import torch

    from sub_volume_renderer_amd.pyramid import pool2x

    n = 1024
    for dtype, mode, es in ((torch.uint8, "mean", 1), (torch.int32, "max", 4)):
        t = torch.randint(0, 200, (n, n, n), dtype=dtype, device="cuda")
        pool2x(t, mode)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            pool2x(t, mode)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        gbs = n ** 3 * es * (1 + 1 / 8) / (ms * 1e-3) / 1e9
        with capsys.disabled():
            print(f"\npool2x {mode} {dtype}: {ms:.3f} ms, {gbs:.0f} GB/s ({gbs / 8000:.2f} of HBM peak)")
        assert gbs > 800.0, (mode, gbs)
        del t


@pytest.mark.gpu
def test_write_multiscale_store_equals_the_reference_pooling_rules(tmp_path):
    """The reference's offline builders in one call: level 0 streamed through the GPU pool kernels slab by slab, every
    level written as a sharded zstd zarr v3 array; what comes back from the stores is the closed-form volume's own
    LODs (mean pooling for density, max pooling for labels), and a SubVolume fed the stores renders like one fed numpy."""
    from sub_volume_renderer_amd import testing, zarr3
    from sub_volume_renderer_amd.pyramid import write_multiscale_store

    n = 256
    d0, l0 = synth.volume(n, 0)
    raw = write_multiscale_store(str(tmp_path / "raw.zarr"), d0, 3, "mean")
    lab = write_multiscale_store(str(tmp_path / "labels.zarr"), l0, 3, "max")
    assert sorted(zarr3.open_group(str(tmp_path / "raw.zarr")).keys()) == ["scale0", "scale1", "scale2"]
    for k in range(3):
        dk, lk = synth.volume(n, k)
        assert raw[k].shape == dk.shape and raw[k].chunks == (16, 16, 16) and raw[k].shards == (64, 64, 64)
        np.testing.assert_array_equal(raw[k][:, :, :], dk)
        np.testing.assert_array_equal(lab[k][:, :, :], lk)
    with pytest.raises(ValueError, match="divisible"):
        write_multiscale_store(str(tmp_path / "bad.zarr"), d0[:250], 3, "mean")
    # the stores as backing data == the numpy arrays as backing data
    import torch

    spec_np = testing.synthetic_spec(n, 200, 120, threshold=0.45, pairs=[synth.volume(n, k) for k in range(3)])
    spec_z = testing.synthetic_spec(n, 200, 120, threshold=0.45, pairs=list(zip(raw, lab)))
    a, b = testing.build(spec_np), testing.build(spec_z)
    ra = a.volume.render(a.camera, 200, 120)
    rb = b.volume.render(b.camera, 200, 120)
    torch.cuda.synchronize()
    assert a.volume._rings.density_storage == b.volume._rings.density_storage == "uint8"
    assert torch.equal(ra.flags, rb.flags), int((ra.flags != rb.flags).sum())
    assert torch.equal(ra.label, rb.label) and torch.equal(ra.rgba, rb.rgba)
    assert int((ra.flags == 2).sum()) > 100, int((ra.flags == 2).sum())
