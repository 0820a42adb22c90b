"""The fused host generator of the synthetic volumes (csrc/synth_host.c, the lazy 4096^3 backing array of config
C4) against the numpy closed form it restates: bit for bit, every LOD, including blocks at the volume's far corner."""
import numpy as np
import pytest

import __graft_entry__ as g
from sub_volume_renderer_amd import synth


@pytest.fixture(scope="module", autouse=True)
def _built():
    g.build_synth()
    synth._host_lib = None
    assert synth.host_lib() is not None


@pytest.mark.parametrize("n,n_labels", [(64, 4096), (1024, 4096), (4096, 1000003), (2048, 1000003)])
@pytest.mark.parametrize("lod", [0, 1, 2])
def test_host_generator_equals_numpy_closed_form(n, n_labels, lod):
    rng = np.random.default_rng(n + lod)
    m = n >> lod
    for _ in range(3):
        shape = [int(min(m, s)) for s in rng.integers(1, [12, 20, 70])]
        off = [int(rng.integers(0, m - s + 1)) for s in shape]
        d, l = synth.block(n, lod, off, shape, n_labels)
        d2, l2 = synth.block_host(n, lod, off, shape, n_labels)
        np.testing.assert_array_equal(d, d2)
        np.testing.assert_array_equal(l, l2)
    # the far corner (largest coordinates the closed form sees)
    shape = [min(m, 4), min(m, 6), min(m, 48)]
    off = [m - s for s in shape]
    d, l = synth.block(n, lod, off, shape, n_labels)
    d2, l2 = synth.block_host(n, lod, off, shape, n_labels, nthreads=3)
    np.testing.assert_array_equal(d, d2)
    np.testing.assert_array_equal(l, l2)


def test_lazy_lod_reads_through_the_host_generator_and_counts_its_time():
    lazy_d, lazy_l = synth.LazyLod(4096, 1, labels=False, n_labels=1000003), synth.LazyLod(4096, 1, labels=True, n_labels=1000003)
    before = synth.LazyLod.read_bytes
    sl = (slice(1000, 1004), slice(8, 16), slice(96, 144))
    d, l = synth.block(4096, 1, (1000, 8, 96), (4, 8, 48), 1000003)
    np.testing.assert_array_equal(lazy_d[sl], d)
    np.testing.assert_array_equal(lazy_l[sl], l)
    assert lazy_d.shape == (2048, 2048, 2048) and lazy_l.dtype == np.uint32
    assert synth.LazyLod.read_bytes - before == d.nbytes + l.nbytes and synth.LazyLod.read_seconds > 0


def test_host_generator_rejects_out_of_range_blocks():
    with pytest.raises(ValueError):
        synth.block_host(1 << 24, 0, (0, 0, 0), (1, 1, 1))
    with pytest.raises(ValueError):
        synth.block_host(64, 0, (-1, 0, 0), (2, 2, 2))
