"""Seeded differential fuzzing (tools/fuzz_parity.py): random volumes, ring windows, cameras, materials, frame
regions and kernel variants; every plane of the HIP render must agree with the oracle."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fuzz_parity  # noqa: E402

from oracle import lmip  # noqa: E402
from sub_volume_renderer_amd import _native as N, testing  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(6))
def test_seeded_random_scenes(block):
    hits = 0
    for seed in range(1000 + 20 * block, 1000 + 20 * (block + 1)):
        spec, region, variant = fuzz_parity.random_spec(seed)
        scene = testing.build(spec)
        N.check(N.lib().svr_set_variant(scene.volume.prepare(), variant), "svr_set_variant")
        ref = lmip.render_spec(spec, region=region, pick_id=scene.volume.id)
        try:                                                 # the production kernel and the instrumented one
            rep = testing.hold_both_to(ref, scene.volume, scene.camera, scene.width, scene.height, region=region, pick=True)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} variant {hex(variant)}: {e}") from e
        for r in (rep["production"], rep["instrumented"]):
            assert np.array_equal(r.pick.cpu().numpy().view(np.uint64), ref.pick), (seed, hex(variant))
        hits += rep["n_hit"] > 0
    assert hits >= 5
