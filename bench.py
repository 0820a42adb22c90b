#!/usr/bin/env python3
"""Benchmark of the LMIP sub-volume march on MI355X (BASELINE.json metric).

A *step* of this bench is one frame: one pass of the hot path (``svr_render``: vs_main + fs_main + raycast of
the reference's WGSL as one HIP kernel) over all rays of a 1920x1080 frame.  Ring buffers are resident in HBM
before the timed region (config C4 streams new chunks into them DURING it: that is its point).

``--config`` (BASELINE.json ``configs``; the default is the one the metric is quoted on):

* ``C2`` (default) 1024^3 uint8 density + uint32 labels, 3 LODs, chunk shapes (16,16,48)/(8,8,48)/(4,4,48), ring
  shapes (32,32,11)/(64,64,11)/(64,64,6) chunks (2.36 GB of ring textures in the reference's layout), camera K1.
* ``C5`` 2048^3, ~1 M labels (1,000,003), 256-entry HSV table, fog_density 0.05, lmip_threshold 0.3, rings
  scaled x2 per axis; generated on the device.
* ``C4`` 4096^3 generated chunk-wise behind a lazy backing array (never resident), K2 fly-through: every frame is
  one render + ``center_on_position(asynchronous=True)``; reports the frame-time distribution and the host ->
  ring upload rate.  (``--volume-n`` shrinks any config for rehearsals.)

``value`` = ray-steps of the frame(s) / time in *full* march mode (threshold = +inf: every ray runs all its
nsteps; the deterministic roofline number).  The realistic early-out LMIP mode is in the ``lmip`` block.
The timed region of K steps is repeated (``spread``) and also run with one frame at a time (``sequential``).

N > 1 (launched by torch.distributed.run): rings replicated per GPU, the frame dealt to ranks in interleaved
row bands (``--tiling rows``) or as config 3's literal grid (``--tiling 2x4``), one gather of every rank's
planes to rank 0 per frame over RCCL (``svr_gather_tiles``) + an un-tile kernel; strong scaling.
"""

from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0      # PCIe Gen5 x16 spec (same guide, chip-level parameters)
PROFILE_ROUND = "r03"     # profiles/<round>/traffic.json: PMC-derived figures of the default workloads (one entry per ring storage)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames (default 20; C4: 240)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prime-s", type=float, default=0.3, help="seconds of untimed frames during setup, before the warm-up "
                    "frames (the first ~50 ms of frames after an idle GPU run 3-4 %% slower: clocks ramping)")
    ap.add_argument("--config", choices=["C2", "C4", "C5"], default="C2")
    ap.add_argument("--volume-n", dest="n", type=int, default=None, help="volume edge (C2 1024, C5 2048, C4 4096)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--camera", choices=["K1", "K2"], default=None, help="default: K1 (C4: the K2 fly-through)")
    ap.add_argument("--band-h", type=int, default=16)
    ap.add_argument("--tiling", default="rows", help="N > 1: 'rows' (interleaved bands), 'grid', or '<gx>x<gy>' e.g. 2x4")
    ap.add_argument("--planes", choices=["rgba", "all"], default="rgba", help="N > 1: gather RGBA only, or depth + label too")
    ap.add_argument("--gather", choices=["svr", "torch"], default="svr",
                    help="N > 1 transport: svr_gather_tiles (RCCL behind the C ABI) or torch.distributed.gather")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--in-flight", type=int, default=4,
                    help="frames kept in flight on separate HIP streams for `value` (the `sequential` block is always 1)")
    ap.add_argument("--repeats", type=int, default=5, help="how often the timed region is repeated for `spread`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--ring-storage", choices=["native", "float32"], default="native",
                    help="native: byte rings for the uint8 volume (identical results); float32: reference layout")
    ap.add_argument("--blocked-twin-all", action="store_true",
                    help="also keep micro-block copies of the coarser LODs, for waves that stage no bricks (blocked_twin='all')")
    ap.add_argument("--no-blocked-twin", action="store_true",
                    help="density rings in rows only: no micro-block copy of the finest LOD (svr_lod_desc::blocked_twin; A/B)")
    ap.add_argument("--float32-block", action="store_true", help="C5: measure the float32-ring block too (C2 does by default)")
    ap.add_argument("--no-float32-block", action="store_true",
                    help="C2 / C5 at N = 1 with native rings: skip the second measurement of the same workload on float32 rings "
                         "(the reference's own layout, _wrapping_buffer.py:50-53), reported as `float32_rings`")
    ap.add_argument("--source-dtype", choices=["uint8", "uint16"], default="uint8",
                    help="C2 / C5: dtype of the density arrays handed to SubVolume (uint16: values x 257, threshold and clim "
                         "scaled alike -> uint16 rings)")
    ap.add_argument("--modes", default="full,lmip", help="march modes to time (full must be included)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo: rehearsal of the N>1 path when several ranks must "
                         "share one GPU (regions are gathered through host memory; not a performance number)")
    ap.add_argument("--check", action="store_true", help="also compare sampled rows with the oracle / the gathered frame with a single-GPU render")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: run the N > 1 pipeline (tiling, RCCL gather, un-tile) in a one-rank process group")
    ap.add_argument("--blocking-too", action="store_true", help="C4: also time the fly-through with blocking reloads")
    ap.add_argument("--source", choices=["zarr3", "synth"], default="zarr3",
                    help="C4: where the backing arrays come from.  zarr3 (default, what BASELINE config 4 says): a sparse, sharded, "
                         "zstd zarr v3 store of the fly-through's corridor (16^3 chunks in 64^3 shards, the layout of the reference's "
                         "builders), written to local disk during setup and read back through sub_volume_renderer_amd.zarr3; "
                         "synth: lazy arrays that generate every block on demand from the closed form (no store)")
    ap.add_argument("--pace-hz", type=float, default=120.0,
                    help="C4: also run the fly-through with frames released at this rate, like a display (0: skip), reported as `paced`")
    ap.add_argument("--store-dir", default=None, help="C4 --source zarr3: where to write the store (default: a fresh directory under $TMPDIR)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# scene descriptions
# ---------------------------------------------------------------------------------------------------
def _ring_shapes(scale):
    r = lambda v: max(2, round(v * scale))  # noqa: E731
    return [(r(32), r(32), r(11)), (r(64), r(64), r(11)), (r(64), r(64), r(6))]


def config2_spec(n, width, height, camera, pairs):
    """BASELINE config 2 (and, shrunk by --volume-n, its rehearsal sizes): SURVEY.md 8d."""
    from sub_volume_renderer_amd import testing

    ring_shapes = _ring_shapes(n / 1024.0)
    spec = testing.synthetic_spec(
        n, width, height, inside=(camera == "K2"), threshold=0.5, fog_density=0.01, ncolors=4,
        chunk_shapes=[(16, 16, 48), (8, 8, 48), (4, 4, 48)], ring_shapes=ring_shapes,
        sizes=[None, (n // 2,) * 3, (n // 4,) * 3], pairs=pairs)
    # LOD0: default window (N-1)*C around the centre; LOD1/2: their whole level
    size0 = tuple((a - 1) * b for a, b in zip(ring_shapes[0], spec.chunk_shapes[0]))
    spec.centers = [(spec.centers[0][0], [size0, (n // 2,) * 3, (n // 4,) * 3])]
    return spec


def config5_spec(n, width, height, camera, pairs):
    """BASELINE config 5: 2048^3, ~1 M labels, 256 hues, fog 0.05, threshold 0.3; rings x2 per axis, so every
    level's window has the same share of its level as in C2 (LOD1/2 again hold their whole level)."""
    from sub_volume_renderer_amd import testing

    ring_shapes = _ring_shapes(n / 1024.0)
    spec = testing.synthetic_spec(
        n, width, height, inside=(camera == "K2"), threshold=0.3, fog_density=0.05, ncolors=256,
        chunk_shapes=[(16, 16, 48), (8, 8, 48), (4, 4, 48)], ring_shapes=ring_shapes,
        sizes=[None, (n // 2,) * 3, (n // 4,) * 3], pairs=pairs)
    size0 = tuple((a - 1) * b for a, b in zip(ring_shapes[0], spec.chunk_shapes[0]))
    spec.centers = [(spec.centers[0][0], [size0, (n // 2,) * 3, (n // 4,) * 3])]
    return spec


def config4_spec(n, width, height):
    """BASELINE config 4: an n^3 volume that is never resident (lazy backing arrays generate blocks on demand),
    C2's chunk and ring shapes, camera K2 inside the volume; every level gets its default window (N-1)*C."""
    from sub_volume_renderer_amd import synth, testing

    pairs = [(synth.LazyLod(n, k, labels=False), synth.LazyLod(n, k, labels=True)) for k in range(3)]
    spec = testing.synthetic_spec(
        n, width, height, inside=True, threshold=0.5, fog_density=0.01, ncolors=4,
        chunk_shapes=[(16, 16, 48), (8, 8, 48), (4, 4, 48)], ring_shapes=_ring_shapes(1.0), sizes=None, pairs=pairs)
    spec.centers = [(spec.cam_position, None)]                 # windows centred on the camera, as the demo scripts do
    return spec


def flythrough_poses(spec, frames, step=2.0):
    """the K2 eye moves `step` voxels per frame along its view direction."""
    import numpy as np

    eye = np.array(spec.cam_position, float)
    d = np.array(spec.cam_target, float) - eye
    d /= np.linalg.norm(d)
    return [(tuple(eye + d * step * k), tuple(eye + d * step * k + d)) for k in range(frames)]


def corridor_shards(spec, positions, shard=64):
    """Per LOD, the set of shard indices (numpy order) the ring windows of `center_on_position(p)` touch for p in
    `positions`: exactly the windows `_wobject.center_on_position` would request (default sizes), computed by the
    product's own host logic on a volume that never touches the device."""
    from itertools import product

    import numpy as np

    from sub_volume_renderer_amd import Roi, SubVolume, SubVolumeMaterial

    plan = SubVolume(SubVolumeMaterial(lmip_threshold=1.0), list(spec.pairs), list(spec.ring_shapes), list(spec.chunk_shapes))
    touched = [set() for _ in plan.wrapping_buffers]
    for pos in positions:
        p = (plan.world.inverse_matrix @ np.array([*pos, 1.0]))[:3][::-1]
        for lod, b in enumerate(plan.wrapping_buffers):
            size = tuple((n - 1) * c for n, c in zip(b.shape_in_chunks, b.chunk_shape_in_pixels))
            offset = tuple(int(c * f - s // 2) for c, s, f in zip(p, size, b.scale_factor))
            roi = b.get_snapped_roi_in_pixels(Roi(offset, size))
            if roi.empty:
                continue
            lo = [max(0, int(v)) for v in roi.begin]
            hi = [min(int(d), int(v)) for d, v in zip(b.backing_data.shape, roi.end)]
            if any(h <= l for l, h in zip(lo, hi)):
                continue
            touched[lod].update(product(*[range(l // shard, (h - 1) // shard + 1) for l, h in zip(lo, hi)]))
    return touched


def write_corridor_store(root, n, n_labels, touched, shard=64, chunk=16):
    """The zarr v3 store of config 4: group `raw.zarr` / `labels.zarr` with arrays `scale0..2` (uint8 / uint32), 16^3
    chunks in 64^3 shards, `bytes` + `zstd` — the layout of scripts/create_mouse_multiscale.py:102-131 — holding ONLY
    the shards in `touched` (every other shard is missing = fill value 0; the fly-through never reads them).  The
    voxels come from the closed form (`synth.block_host`), a row of shards per call."""
    import time

    import numpy as np

    from sub_volume_renderer_amd import synth, zarr3

    t0 = last = time.time()
    stats = {"shards": 0, "bytes_on_disk": 0, "raw_bytes": 0}
    groups = {"raw": zarr3.create_group(os.path.join(root, "raw.zarr")), "labels": zarr3.create_group(os.path.join(root, "labels.zarr"))}
    arrays = []
    for lod, shards in enumerate(touched):
        m = n >> lod
        dens = zarr3.create_array(os.path.join(groups["raw"].path, f"scale{lod}"), (m, m, m), np.uint8, (chunk,) * 3, (shard,) * 3)
        labs = zarr3.create_array(os.path.join(groups["labels"].path, f"scale{lod}"), (m, m, m), np.uint32, (chunk,) * 3, (shard,) * 3)
        rows = {}
        for i0, i1, i2 in sorted(shards):
            rows.setdefault((i0, i1), []).append(i2)
        for (i0, i1), cols in rows.items():
            cols.sort()
            k = 0
            while k < len(cols):                         # runs of consecutive shards along the contiguous axis: one generator call
                j = k
                while j + 1 < len(cols) and cols[j + 1] == cols[j] + 1:
                    j += 1
                off = (i0 * shard, i1 * shard, cols[k] * shard)
                shape = tuple(min(s, m - o) for s, o in zip((shard, shard, (cols[j] - cols[k] + 1) * shard), off))
                d, l = synth.block_host(n, lod, off, shape, n_labels, nthreads=0)
                for q in range(k, j + 1):
                    a2 = (cols[q] - cols[k]) * shard
                    stats["bytes_on_disk"] += zarr3.write_block(dens, (i0, i1, cols[q]), d[:, :, a2:a2 + shard])
                    stats["bytes_on_disk"] += zarr3.write_block(labs, (i0, i1, cols[q]), l[:, :, a2:a2 + shard])
                    stats["shards"] += 2
                stats["raw_bytes"] += d.nbytes + l.nbytes
                k = j + 1
                if time.time() - last > 30.0:            # a sign of life on long setups
                    last = time.time()
                    print(f"[bench] writing the corridor store: level {lod}, {stats['shards']} shard files, "
                          f"{stats['raw_bytes'] / 1e9:.2f} GB of voxels so far ({last - t0:.0f} s)", file=sys.stderr, flush=True)
        arrays.append((zarr3.open_zarr(dens.path), zarr3.open_zarr(labs.path)))
    stats["write_seconds"] = round(time.time() - t0, 2)
    return arrays, stats


def kernel_source_hash():
    """What `roofline.traffic` was measured on: the march kernel's sources (a PMC figure taken on another kernel
    must not be carried along)."""
    h = hashlib.sha256()
    for name in ("march_kernel.hip", "svr_internal.h", "march_dispatch.hip"):
        with open(os.path.join(ROOT, "sub_volume_renderer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def summarise(ms):
    import numpy as np

    a = np.sort(np.asarray(ms, float))
    return {"n": int(a.size), "median_ms": float(np.median(a)), "min_ms": float(a[0]), "max_ms": float(a[-1])}


def main():
    args = parse()
    # OpenMP threads of the host-side helpers (the lazy volume's block generator, the oracle) sleep between parallel
    # regions instead of spinning: a spinning team beside the render loop burns the process's CPU quota
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as g

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if rank == 0:
        g.build_hip()
        g.build_synth()
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":
        local_rank = local_rank % max(1, ndev)           # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        dist.barrier()

    from sub_volume_renderer_amd import FrameRegion, Roi, _native as N, synth, testing
    from sub_volume_renderer_amd.distributed import TiledFrame

    dev = torch.device("cuda", local_rank)
    cfg = args.config
    n = args.n or {"C2": 1024, "C5": 2048, "C4": 4096}[cfg]
    W, H = args.width, args.height
    camera = args.camera or ("K2" if cfg == "C4" else "K1")
    steps = args.steps if args.steps is not None else (240 if cfg == "C4" else 20)
    n_labels = 1000003 if cfg == "C5" else 4096

    # ---- the volume and its rings -------------------------------------------------------------------
    t0 = time.time()
    store = None
    if cfg == "C4":
        spec = config4_spec(n, W, H)
        if args.source == "zarr3":
            # the store of the fly-through's corridor, written now (setup), read during the timed region
            import tempfile

            g.build_host_codecs()
            positions = [spec.cam_position] + [eye for eye, _ in flythrough_poses(spec, args.warmup + steps)]
            touched = corridor_shards(spec, positions)
            root = args.store_dir or tempfile.mkdtemp(prefix="svr_c4_store_")
            if not args.store_dir:                               # a store of our own making (0.5 GB): gone when the bench ends
                import atexit
                import shutil

                atexit.register(shutil.rmtree, root, ignore_errors=True)
            arrays, store = write_corridor_store(root, n, n_labels, touched)
            store["path"] = root
            spec.pairs = arrays                              # zarr arrays as backing data, as the reference is fed (README.md:18)
    else:
        # LOD 0 from the closed form; the coarser levels with the GPU pyramid builder (svr_pool2x, the reference's
        # 2x mean / max pooling rules) — bit-identical to synthesising every level (tests/test_pyramid.py)
        from sub_volume_renderer_amd.pyramid import build_pyramid

        d0, l0 = synth.volume(n, 0, n_labels, xp=torch, device=dev, slab=16 if n >= 512 else 64)
        pairs = build_pyramid(d0, l0, 3)
        del d0, l0
        scale16 = 257 if args.source_dtype == "uint16" else 1
        if scale16 != 1:
            pairs = [((d.to(torch.int32) * scale16).to(torch.uint16), l) for d, l in pairs]
        torch.cuda.synchronize()
        spec = (config5_spec if cfg == "C5" else config2_spec)(n, W, H, camera, pairs)
        if scale16 != 1:
            spec.material.update(lmip_threshold=spec.material["lmip_threshold"] * scale16, clim=(0.0, 255.0 * scale16))
    t_gen = time.time() - t0
    spec.ring_storage = args.ring_storage
    if args.no_blocked_twin:
        spec.blocked_twin = False
    elif args.blocked_twin_all:
        spec.blocked_twin = "all"

    def source_stats(reset=False):
        """(seconds inside the backing arrays' reads, decoded bytes handed out, stored bytes read) since the last reset:
        the lazy generator keeps them on its class, a zarr array on each instance."""
        sec, dec, sto = synth.LazyLod.read_seconds, synth.LazyLod.read_bytes, 0
        for pair in spec.pairs:
            for a in pair:
                if hasattr(a, "stored_bytes"):
                    sec, dec, sto = sec + a.read_seconds, dec + a.read_bytes, sto + a.stored_bytes
                    if reset:
                        a.read_seconds, a.read_bytes, a.stored_bytes = 0.0, 0, 0
        if reset:
            synth.LazyLod.read_seconds, synth.LazyLod.read_bytes = 0.0, 0
        return sec, dec, sto

    source_stats(reset=True)
    t0 = time.time()
    scene = testing.build(spec, device=local_rank)
    scene.volume.synchronize()
    t_load = time.time() - t0
    vol, cam = scene.volume, scene.camera
    handle = vol._rings.handle
    N.check(N.lib().svr_set_variant(handle, args.variant), "svr_set_variant")

    def upload_stats(reset=False):
        b, s = C.c_uint64(0), C.c_double(0.0)
        N.check(N.lib().svr_upload_stats(handle, C.byref(b), C.byref(s), 1 if reset else 0), "svr_upload_stats")
        return int(b.value), float(s.value)

    fill_bytes, fill_seconds = upload_stats(reset=True)
    fill_read_s = source_stats()[0]

    tiled = TiledFrame(W, H, rank, world, args.band_h, force_collective=args.force_collective, tiling=args.tiling)
    transport = "none"
    if collective:
        transport = "torch.distributed.gather"
        if args.backend == "nccl" and args.gather == "svr" and tiled.init_comm(vol):
            transport = tiled.transport
    region = tiled.region
    full_frame = FrameRegion.full(W, H)
    modes = [m for m in args.modes.split(",") if m]
    assert "full" in modes, "--modes must include full (the headline number)"
    lmip_threshold = float(spec.material["lmip_threshold"])
    want_all = args.planes == "all"

    class Harness:
        """The timing loops bound to ONE volume (the default line measures two: native rings and float32 rings)."""

        def __init__(self, vol):
            self.vol = vol
            self.handle = vol._rings.handle

        def set_mode(self, full):
            self.vol.material.lmip_threshold = float("inf") if full else lmip_threshold

        def set_variant(self, v):
            N.check(N.lib().svr_set_variant(self.handle, v), "svr_set_variant")

        def instrumented(self, camera_obj):
            r = self.vol.render(camera_obj, W, H, region=full_frame, count_steps=True)
            torch.cuda.synchronize()
            return dict(steps=int(r.steps.to(torch.int64).sum().item()), hits=int((r.flags == 2).sum().item()),
                        frags=int((r.flags != 0).sum().item()))

        def loop(self, F, prime_s=0.0):
            return FrameLoop(self, F, prime_s)

        def timed(self, loop, mode, nframes, warmup):
            """W untimed frames, then EXACTLY nframes bracketed by barrier + synchronize; max over ranks."""
            self.set_mode(mode == "full")
            # block -> tile placement: with several frames in flight the next frame's head fills this frame's tail anyway,
            # and the camera-independent table (policy 7) is ~1 % faster; one frame at a time wants the cost-sorted one
            # (policy 0, the library's default).  An explicit --variant placement is left alone.
            placement = (7 << 13) if (loop.F > 1 and not (args.variant >> 13) & 7) else 0
            self.set_variant(args.variant | placement)
            for _ in range(warmup):
                loop.frame()
            loop.drain()
            if collective:
                dist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(nframes):
                loop.frame()
            loop.drain()                                     # the K-th frame's gather + un-tile are inside the timed region
            torch.cuda.synchronize()
            if collective:
                dist.barrier()
            dt = time.perf_counter() - t
            if collective:
                tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt

        def kernel_ms(self, mode, out, iters=10):
            """The dominant kernel alone: HIP events on the stream the kernel runs on (svr_time_render)."""
            self.set_mode(mode == "full")
            self.set_variant(args.variant)
            self.vol.prepare()
            cb, fb = self.vol.camera_block(cam), self.vol.frame_block(W, H, region)
            ob = N.Outputs()
            ob.rgba, ob.depth, ob.label, ob.flags, ob.steps = (out.rgba.data_ptr(), out.depth.data_ptr(),
                                                               out.label.data_ptr(), out.flags.data_ptr(), None)
            ms = C.c_float(0)
            N.check(N.lib().svr_time_render(self.handle, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(ms)),
                    "svr_time_render")
            return float(ms.value)

        def measure(self, in_flight, prime_s):
            """Exact step / hit / pixel counts (instrumented kernel, untimed), the contract's K-step region (+ repeats,
            + one frame at a time) per mode, and the kernel alone."""
            counts = {}
            for mode in modes:
                self.set_mode(mode == "full")
                counts[mode] = self.instrumented(cam)
            loop = self.loop(max(1, in_flight), prime_s=prime_s)
            seq = loop if loop.F == 1 else self.loop(1)
            dts = {}
            for mode in modes:
                first = self.timed(loop, mode, steps, args.warmup)                       # the contract's K steps
                more = [self.timed(loop, mode, steps, 0) for _ in range(max(0, args.repeats - 1))]
                one = [self.timed(seq, mode, steps, 1 if i == 0 else 0) for i in range(max(1, args.repeats))]
                dts[mode] = (first, [first] + more, one)
            torch.cuda.synchronize()
            kms = {mode: sorted(self.kernel_ms(mode, loop.outs[0]) for _ in range(3))[1] for mode in modes}   # median of 3 x 10 launches
            return dict(counts=counts, dts=dts, kms=kms, loop=loop,
                        es={"uint8": 1, "uint16": 2}.get(self.vol._rings.density_storage, 4))

    # ---- frames in flight: frame k runs on stream k % F with its own output buffers; on N > 1 each stream
    # carries render -> gather -> un-tile of its frames, so frame k's collective and its tail of long rays overlap
    # frame k+1's march.
    class FrameLoop:
        def __init__(self, h, F, prime_s=0.0):
            self.h, self.F = h, F
            vol = h.vol
            self.outs = []
            for _ in range(F):
                vol._out_cache = {}
                self.outs.append(vol._outputs(region.out_h, region.out_w, False))
            vol._out_cache = {}
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(F)] if F > 1 else [torch.cuda.current_stream(dev)]
            self.k = 0
            self.last = None
            # per-stream resources of the library (placement table + its pinned staging buffer, render marks) are
            # created at a stream's first render: do that here, during setup (a short warm-up never reaches the 4th
            # stream); `prime_s` seconds of untimed frames on top let the GPU's clocks ramp up before the W warm-up frames
            h.set_mode(True)
            t_end = time.perf_counter() + prime_s
            i = 0
            while i < 2 * F or time.perf_counter() < t_end:
                with torch.cuda.stream(self.streams[i % F]):
                    vol.render(cam, W, H, region=region, out=self.outs[i % F])
                i += 1
                if i % 16 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()

        def planes_of(self, res):
            return (res.rgba, res.depth, res.label) if want_all else res.rgba

        def frame(self, camera_obj=None):
            vol = self.h.vol
            slot = self.k % self.F
            self.k += 1
            with torch.cuda.stream(self.streams[slot]):
                if collective:
                    f = tiled.finish(slot, dst=0)               # frame k - F: wait for its gather, un-tile (rank 0)
                    if f is not None:
                        self.last = f
                res = vol.render(camera_obj or cam, W, H, region=region, out=self.outs[slot])
                if collective and args.backend == "nccl":
                    tiled.gather_async(self.planes_of(res), slot, dst=0, volume=vol)
                elif collective:
                    p = self.planes_of(res)
                    tiled.gather_async(tuple(t.cpu() for t in p) if want_all else p.cpu(), slot, dst=0)   # gloo rehearsal
            return res

        def drain(self):
            if collective:
                for k in range(self.k - self.F, self.k):        # oldest first
                    if k >= 0:
                        with torch.cuda.stream(self.streams[k % self.F]):
                            f = tiled.finish(k % self.F, dst=0)
                            if f is not None:
                                self.last = f

    harness = Harness(vol)
    set_mode, instrumented = harness.set_mode, harness.instrumented

    def algo_bytes(c, npix):
        # 4 B per ray-step (r32float texel) + 4 B per hit ray (r32uint label)
        # + per written pixel: 16 B RGBA + 4 B depth + 4 B label + 1 B flags
        return 4 * c["steps"] + 4 * c["hits"] + 25 * npix

    def pmc_entry(ring_storage):
        """The PMC-derived figures of THIS workload on THIS kernel (profiles/<round>/traffic.json, written by
        tools/profile_bench.sh: rocprofv3 cannot run from inside the process it profiles), or a note why there are none."""
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic.json")
        here = dict(config=cfg, n=n, width=W, height=H, camera=camera, variant=args.variant,
                    ring_storage=ring_storage, blocked_twin=not (args.no_blocked_twin or args.blocked_twin_all),
                    kernel_source_sha16=kernel_source_hash())
        if not os.path.exists(tpath):
            return None, None
        with open(tpath) as f:
            tj = json.load(f)
        for e in tj.get("entries", [tj]):
            if all(e.get("workload", {}).get(k) == v for k, v in here.items()):
                return e, {"file": f"profiles/{PROFILE_ROUND}/traffic.json", "command": e.get("command"),
                           "kernel_source_sha16": e["workload"]["kernel_source_sha16"]}
        return None, {"file": f"profiles/{PROFILE_ROUND}/traffic.json",
                      "stale": "taken on another kernel build or workload: not carried over"}

    def roofline_block(m, ring_storage, npix):
        """SURVEY.md 8d's figure for the dominant kernel, and beside it what the counters say binds it."""
        c, k_ms, es = m["counts"]["full"], m["kms"]["full"], m["es"]
        a_full = algo_bytes(c, npix) / (k_ms * 1e-3) / 1e9
        nb = es * c["steps"] + 4 * c["hits"] + 25 * npix               # what the ring's element type really needs
        entry, source = pmc_entry(ring_storage)
        traffic = entry["traffic_bytes_per_launch"] if entry else None
        block = {
            "bound": "hbm", "achieved": a_full, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": a_full / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source,
            "frac_is": "algorithmic-bytes throughput by SURVEY.md 8d's definition (4 B charged per ray-step, the reference's "
                       "r32float texel) over the HBM peak: how the march compares with a texture-unit implementation on the "
                       "reference's layout, NOT the share of HBM bandwidth in use (that is traffic_frac)",
            "kernel": "march_span (full mode)", "kernel_ms": k_ms,
            "algorithmic_bytes": algo_bytes(c, npix),
            "algorithmic_bytes_def": "4 B/ray-step (reference r32float texel) + 4 B/hit + 25 B/pixel (SURVEY.md 8d)",
            "native_layout": {"bytes": nb, "achieved": nb / (k_ms * 1e-3) / 1e9, "frac": nb / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "note": f"{es} B/ray-step: what the {ring_storage} rings really need"},
        }
        if traffic:
            block["traffic_frac"] = traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if entry and entry.get("binding"):
            block["binding"] = entry["binding"]                          # the resource the counters show busiest
        return block

    def result_blocks(m, ring_storage, with_roofline):
        """`value` & co. of one measured volume (top level of the line for the native rings, `float32_rings` block beside)."""
        counts, dts = m["counts"], m["dts"]
        first, rep, one = dts["full"]
        blk = {
            "value": counts["full"]["steps"] / (first / steps) / 1e6, "unit": "Mray-steps/s",
            "frames_per_s": steps / first, "ms_per_step": first / steps * 1e3,
            # the same K-step region repeated, and with ONE frame at a time (no overlap of tails and heads):
            # per-frame kernel time <= sequential ms_per_step must hold inside this record
            "spread": dict(summarise([d / steps * 1e3 for d in rep]), what=f"ms per step over repeats of the {steps}-step region, {m['loop'].F} frames in flight"),
            "sequential": dict(summarise([d / steps * 1e3 for d in one]), frames_in_flight=1,
                               value=counts["full"]["steps"] / (float(np.median(one)) / steps) / 1e6),
        }
        if "lmip" in dts:
            lf, lrep, lone = dts["lmip"]
            blk["lmip"] = {
                "march_mode": f"lmip threshold={lmip_threshold:g} fall_off=0.5 max_samples=10",
                "ray_steps_per_frame": counts["lmip"]["steps"], "hit_rays": counts["lmip"]["hits"],
                "value": counts["lmip"]["steps"] / (lf / steps) / 1e6, "unit": "Mray-steps/s",
                "value_is": "executed iterations of raycast.wgsl:29-62 per second (what the reference's loop would run): "
                            "empty-space skipping executes most of them without fetching a texel",
                "frames_per_s": steps / lf, "ms_per_step": lf / steps * 1e3,
                "spread": summarise([d / steps * 1e3 for d in lrep]),
                "sequential": dict(summarise([d / steps * 1e3 for d in lone]), frames_in_flight=1),
            }
        if with_roofline:
            npix = W * H
            blk["roofline"] = roofline_block(m, ring_storage, npix)
            if "lmip" in m["kms"]:
                a_lmip = algo_bytes(counts["lmip"], npix) / (m["kms"]["lmip"] * 1e-3) / 1e9
                # NOT a roofline fraction: skipped iterations are charged 4 B each without moving a byte
                lmip_entry = (pmc_entry(ring_storage)[0] or {}).get("lmip")
                blk["lmip"]["kernel"] = {"kernel_ms": m["kms"]["lmip"], "ref_equiv_GBps": a_lmip,
                                         "ref_equiv_frac": a_lmip / HBM_PEAK_GBS,
                                         "note": "reference-equivalent bytes (4 B per executed iteration, SURVEY.md 8d) per second "
                                                 "over the HBM peak; the kernel skips most iterations without a fetch, so this is no "
                                                 "measure of memory use and can exceed 1: compare kernel_ms"}
                if lmip_entry:                                  # what the counters of LMIP-only frames show (same stamp as `traffic`)
                    blk["lmip"]["kernel"]["traffic"] = lmip_entry.get("traffic_bytes_per_launch")
                    blk["lmip"]["kernel"]["binding"] = lmip_entry.get("binding")
        return blk

    result = None
    if cfg != "C4":
        m = harness.measure(args.in_flight, args.prime_s)
        counts, loop = m["counts"], m["loop"]
        if rank == 0:
            dens = "u8" if args.source_dtype == "uint8" else "u16"
            workload = {"C2": f"C2: {n}^3 {dens} density + u32 labels, 3 LODs",
                        "C5": f"C5: {n}^3 {dens} density + u32 labels ({n_labels} labels), 3 LODs, 256 hues, fog 0.05, threshold 0.3 of the range"}[cfg]
            blk = result_blocks(m, vol._rings.density_storage, world == 1)
            result = {
                "metric": "Mray-steps/sec (+ frames/sec) of the LMIP sub-volume march at 1920x1080, 3-LOD 1024^3 volume",
                "value": blk["value"],
                "unit": "Mray-steps/s",
                "frames_per_s": blk["frames_per_s"],
                "n_gpus": world, "steps": steps, "warmup": args.warmup,
                "ms_per_step": blk["ms_per_step"],
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": "f32",
                "data": "synthetic",
                "config": {
                    "workload": f"{workload}, chunks (16,16,48)/(8,8,48)/(4,4,48), rings {spec.ring_shapes} chunks, "
                                f"{W}x{H}, camera {camera}, march_mode=full",
                    "ray_steps_per_frame": counts["full"]["steps"],
                    "rays_with_fragment": counts["full"]["frags"],
                    "parallelism": "single" if not collective else
                                   f"frame {'row-bands' if args.tiling == 'rows' else 'tiles ' + args.tiling} x{world}"
                                   f"{' (band_h=%d)' % args.band_h if args.tiling == 'rows' else ''} + gather of "
                                   f"{'rgba+depth+label' if want_all else 'rgba'} via {transport}",
                    "kernel_variant": args.variant,
                    "frames_in_flight": loop.F,
                    "setup_prime_s": args.prime_s,
                    "placement": "64x64-pixel chunks in raster order (svr_set_variant policy 7) for the pipelined loop; "
                                 "cost-sorted per camera (policy 0, default) for `sequential` and `roofline`"
                                 if (loop.F > 1 and not (args.variant >> 13) & 7) else "as --variant says (0: cost-sorted per camera)",
                    "ring_storage": vol._rings.density_storage,
                    "blocked_twin": list(vol._rings.blocked_twin),      # LODs whose density ring is also kept in 128-byte micro-blocks
                },
            }
            for k in ("spread", "sequential", "lmip", "roofline"):
                if k in blk:
                    result[k] = blk[k]
            result["setup_s"] = {"synthesize": round(t_gen, 2), "ring_upload": round(t_load, 2)}
        # ---- the same workload on float32 rings: the layout the reference itself stores (r32float textures,
        # _wrapping_buffer.py:50-53; its pyramid builders write float32 sources, create_mouse_multiscale.py:82), where the
        # 4 B per ray-step of SURVEY.md 8d are the bytes really stored
        if (world == 1 and not collective and not args.no_float32_block and args.ring_storage == "native"
                and (cfg == "C2" or args.float32_block)      # C5 on float32 rings (19 GB, the 4-GiB build of the kernel) only on request
                and args.source_dtype == "uint8" and vol._rings.density_storage != "float32"):
            spec32 = (config5_spec if cfg == "C5" else config2_spec)(n, W, H, camera, pairs)
            spec32.ring_storage = "float32"
            spec32.blocked_twin = spec.blocked_twin
            t0 = time.time()
            scene32 = testing.build(spec32, device=local_rank)
            scene32.volume.synchronize()
            h32 = Harness(scene32.volume)
            h32.set_variant(args.variant)
            m32 = h32.measure(args.in_flight, 0.1)
            assert m32["counts"] == counts, "float32 rings must execute the same iterations and hit the same rays"
            b32 = result_blocks(m32, "float32", True)
            b32["what"] = ("the same volume, camera, material and frame on float32 rings — the reference's own ring layout "
                           "(r32float, _wrapping_buffer.py:50-53): every pixel, label and step count is identical to the native-ring "
                           "frame; 4 B/ray-step is what these rings really store")
            b32["ring_upload_s"] = round(time.time() - t0, 2)
            result["float32_rings"] = b32
            del h32, m32, scene32
            torch.cuda.empty_cache()
    else:
        # ---- C4: the fly-through.  A step = one frame = render + center_on_position(asynchronous=True).
        poses = flythrough_poses(spec, args.warmup + steps)
        cams = []
        for eye, target in poses:
            spec.cam_position, spec.cam_target = eye, target
            cams.append(spec.camera())
        # full mode: every ray runs all its nsteps whatever the rings hold, so the exact step count of the path
        # needs no streaming: one instrumented render per pose
        set_mode(True)
        path_steps = [instrumented(c)["steps"] for c in cams[args.warmup:]]
        loop = harness.loop(1)

        def fly(mode, asynchronous, pace_hz=0.0):
            """Back to the start, W untimed frames, then K timed ones; per-frame wall times (frame k = enqueue the render,
            move the ring windows while it runs, wait for it as a display would).  `pace_hz` > 0: frames are released at
            that rate, like a display's refresh (the wait for the next slot is not part of a frame's time)."""
            set_mode(mode == "full")
            vol.poll_uploads(wait=True)
            vol.center_on_position(poses[0][0])                            # blocking: rings as at the start
            vol.synchronize()
            upload_stats(reset=True)
            source_stats(reset=True)
            for b in vol.wrapping_buffers:
                b.superseded_requests = 0
            times = []
            if collective:
                dist.barrier()
            torch.cuda.synchronize()
            t_start = None
            for k, (eye, _) in enumerate(poses):
                if k == args.warmup:
                    if collective:
                        dist.barrier()
                    torch.cuda.synchronize()
                    t_start = time.perf_counter()
                t = time.perf_counter()
                loop.frame(cams[k])                                            # enqueue the draw ...
                loop.drain()
                vol.center_on_position(eye, asynchronous=asynchronous)         # ... plan / start the window moves beside it
                torch.cuda.current_stream(dev).synchronize()                   # (the order of the reference's do_draw:
                if k >= args.warmup:                                           #  scripts/multi_scale.py:76-80), then wait for the frame
                    times.append((time.perf_counter() - t) * 1e3)
                if pace_hz > 0.0:
                    time.sleep(max(0.0, t + 1.0 / pace_hz - time.perf_counter()))
            torch.cuda.synchronize()
            if collective:
                dist.barrier()
            dt = time.perf_counter() - t_start
            if pace_hz > 0.0:
                time.sleep(1.0 / pace_hz)                                      # the last frame's own requests get one slot too
            landed = vol.poll_uploads(wait=False)
            vol.poll_uploads(wait=True)
            ub, us = upload_stats()
            src = source_stats()
            if collective:
                tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            a = np.sort(np.asarray(times))
            need = (ub / max(1, len(times) + args.warmup))                    # bytes the windows asked for per frame
            return dict(dt=dt, pace_hz=pace_hz, bytes_per_frame=need, frame_ms={"median": float(np.median(a)), "p99": float(a[min(len(a) - 1, int(0.99 * len(a)))]),
                                         "max": float(a[-1]), "mean": float(a.mean()), "over_5ms": int((a > 5.0).sum())},
                        upload={"staged_bytes": ub, "seconds_in_upload_calls": round(us, 4),
                                "GBps": (ub / us / 1e9) if us > 0 else None,
                                "frac_of_pcie_gen5_x16": (ub / us / 1e9 / PCIE_PEAK_GBS) if us > 0 else None,
                                "source": "zarr v3 store (sharded, zstd) read through sub_volume_renderer_amd.zarr3" if store else "lazy closed-form generator",
                                "source_read_seconds": round(src[0], 3),
                                "source_read_GBps": (src[1] / src[0] / 1e9) if src[0] else None,          # decoded bytes handed to the rings
                                "source_stored_GBps": (src[2] / src[0] / 1e9) if (src[0] and src[2]) else None,   # compressed bytes read from disk
                                "all_loads_landed_at_last_frame": bool(landed),
                                # chunk windows that were asked for while a level was still loading and were replaced by a request
                                # for a DIFFERENT window before they started (the latest wins): 0 = every window asked for was loaded
                                "windows_superseded": int(sum(b.superseded_requests for b in vol.wrapping_buffers))})

        runs = {"full": fly("full", True)}
        if "lmip" in modes:
            runs["lmip"] = fly("lmip", True)
        if args.pace_hz > 0:
            # the same path released at a display's rate: what the streaming front-end is for.  Unpaced, the march outruns
            # any source (several hundred frames per second ask for GB/s of new chunks) and superseded requests are
            # dropped; paced, every load should land before the next window move asks for more
            runs["paced"] = fly("lmip" if "lmip" in modes else "full", True, pace_hz=args.pace_hz)
        if args.blocking_too:
            runs["full_blocking"] = fly("full", False)
        if rank == 0:
            r = runs["full"]
            result = {
                "metric": "Mray-steps/sec (+ frames/sec) of the LMIP sub-volume march at 1920x1080, 3-LOD 1024^3 volume",
                "value": sum(path_steps) / r["dt"] / 1e6, "unit": "Mray-steps/s", "frames_per_s": steps / r["dt"],
                "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": r["dt"] / steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {
                    "workload": f"C4: {n}^3 volume " + ("behind a zarr v3 store (16^3 chunks in 64^3 shards, zstd; only the fly-through's corridor is stored), "
                                                         if store else "generated chunk-wise behind a lazy array, ") + "never resident, 3 LODs, chunks "
                                f"(16,16,48)/(8,8,48)/(4,4,48), rings {spec.ring_shapes} chunks, {W}x{H}, K2 fly-through "
                                f"2 voxels/frame, center_on_position(asynchronous=True) every frame, march_mode=full",
                    "ray_steps_total": int(sum(path_steps)),
                    "parallelism": "single" if not collective else f"{args.tiling} x{world} via {transport}",
                    "kernel_variant": args.variant, "frames_in_flight": 1, "ring_storage": vol._rings.density_storage,
                    "blocked_twin": list(vol._rings.blocked_twin),
                },
                "frame_ms": r["frame_ms"], "upload": r["upload"],
                "initial_fill": {"staged_bytes": fill_bytes, "seconds_in_upload_calls": round(fill_seconds, 3),
                                 "GBps": fill_bytes / fill_seconds / 1e9 if fill_seconds else None,
                                 "frac_of_pcie_gen5_x16": fill_bytes / fill_seconds / 1e9 / PCIE_PEAK_GBS if fill_seconds else None,
                                 "source_read_seconds": round(fill_read_s, 2), "wall_seconds": round(t_load, 2)},
                "setup_s": {"describe": round(t_gen, 2), "initial_fill": round(t_load, 2)},
            }
            if store:
                result["store"] = store
            for name in ("lmip", "full_blocking", "paced"):
                if name in runs:
                    q = runs[name]
                    result[name] = {"frames_per_s": steps / q["dt"], "ms_per_step": q["dt"] / steps * 1e3,
                                    "frame_ms": q["frame_ms"], "upload": q["upload"]}
            if "paced" in runs:
                q = runs["paced"]
                result["paced"].update(
                    march_mode="lmip" if "lmip" in modes else "full", pace_hz=q["pace_hz"],
                    what="the same fly-through with frames released at a display's refresh rate (the wait for the next slot is not "
                         "part of frame_ms): the load the asynchronous streaming path is built for")
            # what the unpaced path asks of its source: new ring bytes per frame x the frame rate it reaches
            src_rate = r["upload"]["source_read_GBps"]
            result["streaming_demand"] = {
                "ring_bytes_per_frame": r["bytes_per_frame"],
                "needed_source_GBps_at_the_unpaced_rate": r["bytes_per_frame"] * (steps / r["dt"]) / 1e9,
                "source_GBps": src_rate,
                "frames_per_s_the_source_can_feed": (src_rate * 1e9 / r["bytes_per_frame"]) if (src_rate and r["bytes_per_frame"]) else None,
                "note": "staged bytes only count loads that were not superseded: the true demand of the unpaced path is higher"}
        counts = {"full": {"steps": int(np.mean(path_steps)), "hits": 0, "frags": 0}}

    # ---- CPU baseline: the oracle (a port: the reference itself cannot run offline) on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import lmip as oracle_lmip

        # threads = the CPUs this process may really use: affinity mask capped by the cgroup CPU quota (the
        # GPU box shows 256 hardware threads but grants 16 CPUs' worth of time to a one-GPU job)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        quota_note = ""
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, per = f.read().split()[:2]
            if q != "max":
                allowed = max(1, int(-(-int(q) // int(per))))
                if allowed < cores:
                    quota_note = f" (cgroup CPU quota: {allowed} of {cores} hardware threads)"
                    cores = allowed
        except (OSError, ValueError):
            pass
        if cfg == "C4":                                      # the pose of the first timed frame, rings as they are now
            spec.cam_position, spec.cam_target = poses[args.warmup]
            vol.center_on_position(poses[args.warmup][0])
            vol.synchronize()
        rings = []
        for b in vol.wrapping_buffers:
            d, l = b.read_ring(Roi((0, 0, 0), b.shape_in_pixels))
            u = b.uniform_buffer.data
            rings.append(dict(density=d, labels=l,
                              offset=tuple(int(v) for v in u["current_logical_offset_in_pixels"]),
                              shape=tuple(int(v) for v in u["current_logical_shape_in_pixels"]),
                              scale=tuple(float(v) for v in u["scale_factor"])))
        mats = spec.matrices()
        vdim = tuple(float(v) for v in vol._volume_dimensions)
        m_full = dict(spec.material)
        m_full["lmip_threshold"] = float("inf")
        # calibrate on every 32nd row, then size the sample to ~cpu_seconds of oracle time
        cal = FrameRegion(0, 0, W, -(-H // 32), 1, 32)
        t = time.perf_counter()
        ref = oracle_lmip.render(rings, mats, vdim, m_full, W, H, region=cal, nthreads=cores)
        t_cal = time.perf_counter() - t
        rate = max(1.0, ref.steps.astype(np.int64).sum() / t_cal)
        frame_steps = counts["full"]["steps"]
        frac = min(1.0, args.cpu_seconds * rate / max(1, frame_steps))
        stride = max(1, int(round(1.0 / frac)))
        nrows = -(-H // stride)
        sample = FrameRegion(0, 0, W, nrows, 1, stride)
        est = frac * frame_steps / rate                                 # seconds for one pass over the sample
        reps = max(1, min(16, int(round(args.cpu_seconds / max(est, 1e-3))))) if stride == 1 else 1
        base = {}
        for mode in modes:
            m = dict(spec.material)
            m["lmip_threshold"] = float("inf") if mode == "full" else lmip_threshold
            t = time.perf_counter()
            for _ in range(reps if mode == "full" else 1):
                ref = oracle_lmip.render(rings, mats, vdim, m, W, H, region=sample, nthreads=cores)
            dtc = time.perf_counter() - t
            k = reps if mode == "full" else 1
            base[mode] = (k * int(ref.steps.astype(np.int64).sum()), dtc, ref)
        st, dtc, ref = base["full"]
        result["cpu_baseline"] = {
            "value": st / dtc / 1e6, "unit": "Mray-steps/s", "cores": cores, "kind": "port",
            "sample": f"{reps} pass(es) over every {stride}th row of the same frame ({nrows} rows; {st} ray-steps, {dtc:.1f} s of oracle time), "
                      f"full mode, {cores} OpenMP threads{quota_note}; CPU restatement of the reference shader "
                      "(the reference itself cannot run offline: pygfx/wgpu absent)",
        }
        if "lmip" in base:
            result["cpu_baseline"]["lmip_value"] = base["lmip"][0] / base["lmip"][1] / 1e6
        if args.check:
            set_mode(True)
            res = vol.render(spec.camera(), W, H, region=sample, count_steps=True)
            torch.cuda.synchronize()
            result["check"] = testing.compare(res, ref)

    if collective and args.check and cfg != "C4":
        set_mode(True)
        frame_full = vol.render(cam, W, H, region=full_frame) if rank == 0 else None
        torch.cuda.synchronize()
        loop.frame()
        loop.drain()
        torch.cuda.synchronize()
        if rank == 0:
            got = loop.last
            names = ("rgba", "depth", "label") if want_all else ("rgba",)
            got = got if want_all else (got,)
            rep = {}
            for name, gt in zip(names, got):
                want = getattr(frame_full, name)
                gt = gt.to(want.device)
                bad = (gt != want)
                bad = bad.any(dim=-1) if bad.dim() == 3 else bad
                rep[name] = {"equal": bool(torch.equal(gt, want)), "mismatched_pixels": int(bad.sum().item())}
            rep["gathered_frame_equals_single_gpu_render"] = all(v["equal"] for v in rep.values())
            rep["alpha1_gathered"] = int((got[0][..., 3] == 1).sum().item())
            rep["transport"] = transport
            result["check" if world > 1 else "check_gathered"] = rep
    if rank == 0:
        print(json.dumps(result), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
