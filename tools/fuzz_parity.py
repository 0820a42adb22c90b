#!/usr/bin/env python3
"""Differential fuzzing of the HIP march against the CPU oracle: random volumes (1-4 LODs, random chunk
and ring shapes, anisotropic, u8 / u16 / f32 rings, with or without segmentation), ring windows, cameras (outside /
inside / grazing), materials (LMIP, MIP and weighted average, clipping planes), frame sizes, frame regions and kernel variants
(empty-space skipping on and off, bricks always / never / by probe, tile shapes, placements).  Integer planes must be identical, float planes within 1e-4.
Every case runs BOTH instantiations of the march (`production`: count_steps off, the code object bench.py times;
`instrumented`: the COUNT build with exact step counts) unless `kernel=` names one.
usage: fuzz_parity.py [cases] [first_seed] [brick] [kernel=both|production|instrumented]
(prints one line per failing case, then a summary)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lmip  # noqa: E402
from sub_volume_renderer_amd import FrameRegion, _native as N, testing  # noqa: E402

VARIANTS = [0x000, 0x000, 0x200, 0x100, 0x002, 0x2000, 0x4000, 0x250, 0x230, 0x001, 0x204, 0xA202, 0x008, 0x208]


BRICK_VARIANTS = [0x200, 0x200, 0x204, 0x230, 0x250, 0xA202, 0x2200, 0x000, 0x208]


def random_spec(seed, brick=False):
    """``brick=True`` biases the draw towards scenes that stage LDS bricks: 16-voxel chunks along x, byte
    data, larger volumes and frames, variants that always stage."""
    rng = np.random.default_rng(seed)
    nl = int(rng.integers(1, 5))
    chunk0 = [int(rng.choice([4, 8])), int(rng.choice([4, 8])), int(rng.choice([8, 16, 12]))]
    if brick:
        chunk0[2] = 16 << max(0, nl - 2)                    # rows stay multiples of 16 bytes on every LOD
    nch = [int(rng.integers(3, 8)) for _ in range(3)]
    shape0 = [c * k * (1 << (nl - 1)) for c, k in zip(chunk0, nch)]
    cap = 160 if brick else 96
    shape0 = [min(s, cap) // (c * (1 << (nl - 1))) * (c * (1 << (nl - 1))) or c * (1 << (nl - 1)) for s, c in zip(shape0, chunk0)]
    pairs, chunks, rings = [], [], []
    smooth = rng.random() < 0.7
    for l in range(nl):
        shp = tuple(s >> l for s in shape0)
        zz, yy, xx = np.meshgrid(*[np.arange(s, dtype=np.float32) / max(s, 1) for s in shp], indexing="ij")
        base = 120 + 100 * np.cos(6 * xx + 2 * yy + seed) * np.cos(4 * zz - 3 * yy) if smooth else 40
        d = np.clip(base + rng.integers(0, 60, shp), 0, 255)
        if rng.random() < 0.25:
            d[rng.random(shp) < 0.5] = 0
        d = d.astype(np.uint8) if (brick or rng.random() < 0.8) else (d / 255.0).astype(np.float32)
        lab = rng.integers(0, int(rng.choice([3, 1000, 2**31])), shp).astype(np.uint32)
        pairs.append((d, lab))
        ch = tuple(max(1, c >> min(l, 1)) if rng.random() < 0.5 else c for c in chunk0)
        ch = tuple(c for c in ch)
        ch = tuple(min(c, s) for c, s in zip(ch, shp))
        # chunk must divide nothing in particular; ring at least 2 chunks, at most covering the level + 1
        chunks.append(ch)
        rings.append(tuple(int(rng.integers(2, max(3, s // c + 2))) for s, c in zip(shp, ch)))
    if any(p[0].dtype != pairs[0][0].dtype for p in pairs):
        pairs = [(p[0].astype(np.float32) if p[0].dtype != np.float32 else p[0], p[1]) for p in pairs]
    W, H = (int(rng.integers(64, 260)), int(rng.integers(48, 180))) if brick else (int(rng.integers(9, 150)), int(rng.integers(7, 110)))
    size_xyz = np.array(shape0[::-1], float)
    centre = size_xyz * rng.uniform(0.2, 0.8, 3)
    mode = rng.integers(0, 3)
    if mode == 0:      # outside
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        eye = size_xyz / 2 + d * size_xyz.max() * rng.uniform(0.9, 2.5)
        target = centre
    elif mode == 1:    # inside
        eye = size_xyz * rng.uniform(0.1, 0.9, 3)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        target = eye + d
    else:              # axis-aligned-ish from outside (grazing faces, exact zeros in the direction)
        ax = int(rng.integers(0, 3)); d = np.zeros(3); d[ax] = rng.choice([-1.0, 1.0])
        d += rng.choice([0.0, 0.0, 0.03], 3) * rng.normal(size=3)
        eye = size_xyz / 2 - d * size_xyz.max() * rng.uniform(0.8, 2.0)
        target = eye + d
    f32data = pairs[0][0].dtype == np.float32
    scale = 1.0 / 255.0 if f32data else 1.0
    colors = [(float(rng.random()), float(rng.choice([0.0, 1.0, rng.random()])), 1.0) for _ in range(int(rng.integers(1, 17)))]
    material = dict(
        lmip_threshold=float(rng.choice([rng.uniform(0, 260), 127.5, 0.0, float("inf"), 255.0])) * scale,
        lmip_fall_off=float(rng.choice([0.5, rng.uniform(0, 1.2)])), lmip_max_samples=int(rng.integers(0, 21)),
        fog_density=float(rng.choice([0.0, 0.01, rng.uniform(0, 5)])), fog_color=tuple(float(v) for v in rng.random(3)),
        colors=colors, clim=(0.0, 255.0 * scale) if rng.random() < 0.7 else (float(rng.uniform(0, 50)) * scale, float(rng.uniform(100, 300)) * scale),
        gamma=float(rng.choice([1.0, rng.uniform(0.3, 3.0)])), opacity=float(rng.random()))
    spec = testing.SceneSpec(
        pairs=pairs, chunk_shapes=chunks, ring_shapes=rings, material=material, width=W, height=H,
        cam_position=tuple(eye), cam_target=tuple(target), fov=float(rng.uniform(15, 110)),
        depth_range=(float(size_xyz.max()) / 500.0, float(size_xyz.max()) * 20.0),
        centers=[(tuple(centre), None)],
    )
    if rng.random() < 0.3:
        spec.world_scale = tuple(float(v) for v in rng.uniform(0.5, 3.0, 3))
        spec.world_position = tuple(float(v) for v in rng.uniform(-10, 10, 3))
        spec.cam_position = tuple(np.array(spec.cam_position) * np.array(spec.world_scale) + np.array(spec.world_position))
        spec.cam_target = tuple(np.array(spec.cam_target) * np.array(spec.world_scale) + np.array(spec.world_position))
        spec.centers = [(tuple(np.array(centre) * np.array(spec.world_scale) + np.array(spec.world_position)), None)]
    if rng.random() < 0.3:
        spec.centers.append((tuple(np.array(spec.centers[0][0]) + rng.uniform(-9, 9, 3)), None))     # a second window move (ring wrap)
    if rng.random() < 0.2:
        spec.colorspace = "linear"
    spec.ring_storage = "native" if (brick or rng.random() < 0.8) else "float32"
    # round-2 features (drawn after everything else, so that earlier seeds keep their geometry)
    extra = rng.random(5)
    if extra[0] < 0.15 and not f32data:                   # uint16 sources -> uint16 rings
        spec.pairs = [(d.astype(np.uint16) * 257, l) for d, l in spec.pairs]
        m = spec.material
        m["lmip_threshold"] = m["lmip_threshold"] * 257.0
        m["clim"] = (m["clim"][0] * 257.0, m["clim"][1] * 257.0)
    if extra[1] < 0.12:
        spec.material["render_mode"] = "mip"
    if extra[2] < 0.15:                                   # 1-3 world-space planes through points of the volume
        planes = []
        for _ in range(int(rng.integers(1, 4))):
            nrm = rng.normal(size=3)
            nrm /= np.linalg.norm(nrm)
            pt = (size_xyz * rng.uniform(0.1, 0.9, 3)) * np.array(spec.world_scale) + np.array(spec.world_position)
            planes.append((float(nrm[0]), float(nrm[1]), float(nrm[2]), float(nrm @ pt)))
        spec.material["clipping_planes"] = planes
        spec.material["clipping_mode"] = "ALL" if extra[3] < 0.3 else "ANY"
    if extra[4] < 0.08:                                   # a volume without segmentation
        spec.pairs = [(d, None) for d, _ in spec.pairs]
    region = None
    r = rng.random()
    if r < 0.2:
        x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
        region = FrameRegion.tile(x0, y0, int(rng.integers(1, W - x0 + 1)), int(rng.integers(1, H - y0 + 1)))
    elif r < 0.35:
        world = int(rng.integers(2, 5))
        region = FrameRegion.stripes(W, H, int(rng.integers(0, world)), world, int(rng.choice([1, 3, 8, 16])))
    variant = int(rng.choice(BRICK_VARIANTS if brick else VARIANTS))
    last = rng.random(2)                                      # drawn after everything else (earlier seeds keep their scene)
    if last[0] < 0.08 and spec.material.get("render_mode", "lmip") == "lmip":
        spec.material["render_mode"] = "weighted_average"
        spec.material["weight_falloff"] = float(rng.choice([0.0, 0.5, last[1] * 6.0]))
    return spec, region, variant


def main():
    import torch

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    brick = "brick" in sys.argv[3:]
    kernel = ([a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("kernel=")] or ["both"])[0]
    assert kernel in ("both", "production", "instrumented"), kernel
    runs = {"both": (False, True), "production": (False,), "instrumented": (True,)}[kernel]
    bricks = twins = 0
    bad = skipped = hits = 0
    for seed in range(first, first + cases):
        if (seed - first) % 1000 == 999:                     # long campaigns: a sign of life once a minute or so
            print(f"... {seed - first + 1} of {cases} cases, {bad} mismatching so far", file=sys.stderr, flush=True)
        try:
            spec, region, variant = random_spec(seed, brick)
            ovol = lmip.oracle_volume(spec)
        except Exception as e:          # a configuration the reference's own assertions reject
            skipped += 1
            continue
        try:
            scene = testing.build(spec)
        except Exception as e:
            print(f"seed {seed}: product rejected a scene the oracle accepted: {type(e).__name__}: {e}", flush=True)
            bad += 1
            continue
        N.check(N.lib().svr_set_variant(scene.volume.prepare(), variant), "svr_set_variant")
        import ctypes as C
        dbg = (C.c_uint32 * 8)()
        N.lib().svr_debug_counters(scene.volume._rings.handle, dbg, 1)
        ref = lmip.render_spec(spec, region=region, vol=ovol, pick_id=scene.volume.id)
        ok = True
        for counted in runs:
            res = scene.volume.render(scene.camera, scene.width, scene.height, region=region, count_steps=counted, pick=True)
            torch.cuda.synchronize()
            if counted:
                N.lib().svr_debug_counters(scene.volume._rings.handle, dbg, 1)
                bricks += dbg[2] > 0
                tm = (C.c_uint64 * 16)()
                N.lib().svr_debug_timers(scene.volume._rings.handle, tm, 1)
                twins += tm[15] > 0                                   # batches gathered from the micro-block copy of a ring
            rep = testing.compare(res, ref)
            pick_ok = bool(np.array_equal(res.pick.cpu().numpy().view(np.uint64), ref.pick))
            this_ok = (rep["flags_equal"] and rep["labels_equal"] and rep.get("steps_equal", True) and pick_ok
                       and rep["rgba_max_rel"] <= 1e-4 and rep["depth_max_abs"] <= 1e-4)
            if not this_ok:
                print(f"seed {seed}: MISMATCH kernel={'instrumented' if counted else 'production'} variant={variant:#x} "
                      f"region={region} pick_ok={pick_ok} {rep}", flush=True)
            ok = ok and this_ok
        hits += rep["n_hit"] > 0
        bad += not ok
        del scene
    print(f"fuzz{' (brick-biased)' if brick else ''} kernel={kernel}: {cases} cases from seed {first}: {bad} mismatching, {skipped} rejected by both, "
          f"{hits} with hits, {bricks} staged LDS bricks, {twins} gathered from a micro-block copy", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
