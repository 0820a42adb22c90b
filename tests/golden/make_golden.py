#!/usr/bin/env python3
"""Generates tests/golden/render_golden.npz from the CPU oracle (oracle/lmip_oracle.c).

The reference cannot be imported or executed offline (pygfx / wgpu absent) and holds no rendered
fixture of its own, so these vectors come from the restatement, not from the reference: they pin
the oracle against drift and give the GPU tests a committed target.  Inputs are regenerated from
code (sub_volume_renderer_amd.testing / synth); only outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import lmip  # noqa: E402
from sub_volume_renderer_amd import testing  # noqa: E402


def specs():
    demo = testing.multiscale_demo_spec(96, 96, tiles=3)
    k1 = testing.synthetic_spec(48, 80, 48, threshold=0.35)
    k2 = testing.synthetic_spec(48, 80, 48, inside=True, threshold=0.35, fog_density=0.3, ncolors=7)
    return {"demo": demo, "k1": k1, "k2": k2}


def main():
    out = {}
    for name, spec in specs().items():
        r = lmip.render_spec(spec, nthreads=1)
        for plane in ("rgba", "depth", "label", "flags", "steps"):
            out[f"{name}_{plane}"] = getattr(r, plane)
        print(name, {k: int((r.flags == k).sum()) for k in (0, 1, 2)}, "steps", int(r.steps.sum()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_golden.npz"), **out)


if __name__ == "__main__":
    main()
