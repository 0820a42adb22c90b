#!/usr/bin/env python3
"""Benchmark of the LMIP sub-volume march on MI355X (BASELINE.json metric).

A *step* of this bench is one frame: one pass of the hot path (``svr_render``:
vs_main + fs_main + raycast of the reference's WGSL as one HIP kernel) over all
rays of a 1920x1080 frame of BASELINE config 2 — 1024^3 uint8 density + uint32
labels, 3 LODs, chunk shapes (16,16,48)/(8,8,48)/(4,4,48), ring shapes
(32,32,11)/(64,64,11)/(64,64,6) chunks (2.36 GB of ring textures), camera K1
(SURVEY.md §8d).  Ring buffers are resident in HBM before the timed region.

``value`` = ray-steps of the frame / frame time in *full* march mode (threshold =
+inf: every ray runs all its nsteps; the deterministic roofline number).  The
realistic early-out LMIP mode (threshold 0.5) is reported in the ``lmip`` block.

N > 1 (launched by torch.distributed.run): ring buffers replicated per GPU, the
frame dealt to ranks in interleaved row bands, one RCCL gather of the RGBA bands
to rank 0 + an un-tile kernel per frame; strong scaling (one frame, N GPUs).
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--volume-n", dest="n", type=int, default=1024, help="volume edge (config 2: 1024)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--camera", choices=["K1", "K2"], default="K1")
    ap.add_argument("--band-h", type=int, default=16)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--in-flight", type=int, default=4,
                    help="frames kept in flight on separate HIP streams (1: strictly one after the other)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--ring-storage", choices=["native", "float32"], default="native",
                    help="native: byte rings for the uint8 volume (identical results); float32: reference layout")
    ap.add_argument("--modes", default="full,lmip", help="march modes to time (full must be included)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo: rehearsal of the N>1 path when several ranks must "
                         "share one GPU (bands are gathered through host memory; not a performance number)")
    ap.add_argument("--check", action="store_true", help="also compare the sampled rows with the oracle")
    ap.add_argument("--force-collective", action="store_true",
                    help="with --gpus 1: run the N > 1 pipeline (bands, RCCL gather, un-tile) in a one-rank process group")
    return ap.parse_args()


def config2_spec(n, width, height, camera, pairs):
    from sub_volume_renderer_amd import testing

    s = n / 1024.0
    ring_shapes = [(max(2, round(32 * s)), max(2, round(32 * s)), max(2, round(11 * s))),
                   (max(2, round(64 * s)), max(2, round(64 * s)), max(2, round(11 * s))),
                   (max(2, round(64 * s)), max(2, round(64 * s)), max(2, round(6 * s)))]
    spec = testing.synthetic_spec(
        n, width, height, inside=(camera == "K2"), threshold=0.5, fog_density=0.01, ncolors=4,
        chunk_shapes=[(16, 16, 48), (8, 8, 48), (4, 4, 48)], ring_shapes=ring_shapes,
        sizes=[None, (n // 2,) * 3, (n // 4,) * 3], pairs=pairs)
    # LOD0: default window (N-1)*C around the centre; LOD1/2: their whole level
    r0 = ring_shapes[0]
    c0 = spec.chunk_shapes[0]
    size0 = tuple((a - 1) * b for a, b in zip(r0, c0))
    spec.centers = [(spec.centers[0][0], [size0, (n // 2,) * 3, (n // 4,) * 3])]
    return spec


def main():
    args = parse()
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
    n, W, H = args.n, args.width, args.height
    t0 = time.time()
    pairs = [synth.volume(n, k, 4096, xp=torch, device=dev, slab=16 if n >= 512 else 64) for k in range(3)]
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    spec = config2_spec(n, W, H, args.camera, pairs)
    spec.ring_storage = args.ring_storage
    t0 = time.time()
    scene = testing.build(spec, device=local_rank)
    scene.volume.synchronize()
    t_load = time.time() - t0
    vol, cam = scene.volume, scene.camera
    N.check(N.lib().svr_set_variant(vol._rings.handle, args.variant), "svr_set_variant")

    tiled = TiledFrame(W, H, rank, world, args.band_h, force_collective=args.force_collective)
    region = tiled.region
    full_frame = FrameRegion.full(W, H)
    modes = [m for m in args.modes.split(",") if m]
    assert "full" in modes, "--modes must include full (the headline number)"

    def set_mode(full):
        vol.material.lmip_threshold = float("inf") if full else 0.5 * 255.0

    # ---- exact step / hit / pixel counts of the whole frame (instrumented kernel, untimed)
    counts = {}
    for mode in modes:
        set_mode(mode == "full")
        r = vol.render(cam, W, H, region=full_frame, count_steps=True)
        torch.cuda.synchronize()
        counts[mode] = dict(steps=int(r.steps.to(torch.int64).sum().item()),
                            hits=int((r.flags == 2).sum().item()),
                            frags=int((r.flags != 0).sum().item()))
    vol._out_cache = {}

    # ---- outputs for the timed loop.  Successive frames are independent (camera poses are known ahead
    # in a fly-through), so `in_flight` frames are kept in flight: frame k runs on stream k % in_flight
    # with its own band buffer; on N > 1 each stream carries render -> RCCL gather -> un-tile of its
    # frames, so frame k's collective and its tail of long rays overlap frame k+1's march.
    F = max(1, args.in_flight)
    vol._out_cache = {}
    outs = []
    for _ in range(F):
        outs.append(vol._outputs(region.out_h, region.out_w, False))
        vol._out_cache = {}
    out = outs[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)] if F > 1 else [torch.cuda.current_stream(dev)]
    last_frame = [None]
    frame_no = [0]
    pipelined = collective

    def frame():
        slot = frame_no[0] % F
        frame_no[0] += 1
        with torch.cuda.stream(streams[slot]):
            if pipelined:
                f = tiled.finish(slot, dst=0)               # frame k - F: wait for its gather, un-tile (rank 0)
                if f is not None:
                    last_frame[0] = f
            res = vol.render(cam, W, H, region=region, out=outs[slot])
            if pipelined and args.backend == "nccl":
                tiled.gather_async(res.rgba, slot, dst=0, volume=vol)      # RCCL gather of this frame's RGBA bands
            elif pipelined:
                tiled.gather_async(res.rgba.cpu(), slot, dst=0)             # gloo rehearsal through host memory

    def drain():
        if pipelined:
            for k in range(frame_no[0] - F, frame_no[0]):                   # oldest first
                if k >= 0:
                    with torch.cuda.stream(streams[k % F]):
                        f = tiled.finish(k % F, dst=0)
                        if f is not None:
                            last_frame[0] = f

    def timed(mode, steps, warmup):
        set_mode(mode == "full")
        for _ in range(warmup):
            frame()
        drain()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            frame()
        drain()                                          # the K-th frame's gather + un-tile are inside the timed region
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        dt = time.perf_counter() - t
        if collective:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    dt_full = timed("full", args.steps, args.warmup)
    dt_lmip = timed("lmip", args.steps, args.warmup) if "lmip" in modes else None

    # ---- roofline of the dominant kernel: HIP events on the stream the kernel runs on
    def kernel_ms(mode, iters=10):
        set_mode(mode == "full")
        vol.prepare()
        cb, fb = vol.camera_block(cam), vol.frame_block(W, H, region)
        ob = N.Outputs()
        ob.rgba, ob.depth, ob.label, ob.flags, ob.steps = (out.rgba.data_ptr(), out.depth.data_ptr(),
                                                           out.label.data_ptr(), out.flags.data_ptr(), None)
        ms = C.c_float(0)
        N.check(N.lib().svr_time_render(vol._rings.handle, C.byref(cb), C.byref(fb), C.byref(ob), iters, C.byref(ms)),
                "svr_time_render")
        return float(ms.value)

    torch.cuda.synchronize()
    k_full = kernel_ms("full")
    k_lmip = kernel_ms("lmip") if "lmip" in modes else None

    def algo_bytes(c, npix):
        # 4 B per ray-step (r32float texel) + 4 B per hit ray (r32uint label)
        # + per written pixel: 16 B RGBA + 4 B depth + 4 B label + 1 B flags
        return 4 * c["steps"] + 4 * c["hits"] + 25 * npix

    u8 = vol._rings.density_storage == "uint8"

    def native_bytes(c, npix):
        # what the kernel actually has to fetch with byte rings: 1 B per ray-step
        return (1 if u8 else 4) * c["steps"] + 4 * c["hits"] + 25 * npix

    result = None
    if world == 1:
        npix = W * H
        a_full = algo_bytes(counts["full"], npix) / (k_full * 1e-3) / 1e9
        a_lmip = algo_bytes(counts["lmip"], npix) / (k_lmip * 1e-3) / 1e9 if k_lmip else None
    if rank == 0:
        ms_full = dt_full / args.steps * 1e3
        ms_lmip = dt_lmip / args.steps * 1e3 if dt_lmip else None
        result = {
            "metric": "Mray-steps/sec (+ frames/sec) of the LMIP sub-volume march at 1920x1080, 3-LOD 1024^3 volume",
            "value": counts["full"]["steps"] / (dt_full / args.steps) / 1e6,
            "unit": "Mray-steps/s",
            "frames_per_s": args.steps / dt_full,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_full,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"C2: {n}^3 u8 density + u32 labels, 3 LODs, chunks (16,16,48)/(8,8,48)/(4,4,48), "
                            f"rings {spec.ring_shapes} chunks, {W}x{H}, camera {args.camera}, march_mode=full",
                "ray_steps_per_frame": counts["full"]["steps"],
                "rays_with_fragment": counts["full"]["frags"],
                "parallelism": "single" if not collective else f"frame row-bands x{world} (band_h={args.band_h}) + RCCL gather",
                "kernel_variant": args.variant,
                "frames_in_flight": F,
                "ring_storage": vol._rings.density_storage,
            },
        }
        if dt_lmip:
            result["lmip"] = {
                "march_mode": "lmip threshold=0.5*255 fall_off=0.5 max_samples=10",
                "ray_steps_per_frame": counts["lmip"]["steps"], "hit_rays": counts["lmip"]["hits"],
                "value": counts["lmip"]["steps"] / (dt_lmip / args.steps) / 1e6, "unit": "Mray-steps/s",
                "frames_per_s": args.steps / dt_lmip, "ms_per_step": ms_lmip,
            }
        result["setup_s"] = {"synthesize": round(t_gen, 2), "ring_upload": round(t_load, 2)}
        if world == 1:
            # HBM bytes per launch from the PMC passes of this same command (rocprofv3 cannot be run from
            # inside the process it profiles): profiles/r01/traffic.json, valid for the default workload only
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01", "traffic.json")
            if os.path.exists(tpath) and (n, W, H, args.camera, args.variant, u8) == (1024, 1920, 1080, "K1", 0, True):
                with open(tpath) as f:
                    traffic = json.load(f)["traffic_bytes_per_launch"]
            result["roofline"] = {
                "bound": "hbm", "achieved": a_full, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": a_full / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "march_span (full mode)", "kernel_ms": k_full,
                "algorithmic_bytes": algo_bytes(counts["full"], W * H),
                "algorithmic_bytes_def": "4 B/ray-step (reference r32float texel) + 4 B/hit + 25 B/pixel (SURVEY.md 8d)",
                "native_layout": {"bytes": native_bytes(counts["full"], W * H),
                                  "achieved": native_bytes(counts["full"], W * H) / (k_full * 1e-3) / 1e9,
                                  "note": "byte rings: 1 B/ray-step actually needed" if u8 else "same as reference layout"},
            }
            if k_lmip:
                result["lmip"]["roofline"] = {"achieved": a_lmip, "frac": a_lmip / HBM_PEAK_GBS, "kernel_ms": k_lmip}

    # ---- CPU baseline: the oracle (a port: the reference itself cannot run offline) on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import lmip as oracle_lmip

        rings = []
        for b in vol.wrapping_buffers:
            d, l = b.read_ring(Roi((0, 0, 0), b.shape_in_pixels))
            u = b.uniform_buffer.data
            rings.append(dict(density=d, labels=l,
                              offset=tuple(int(v) for v in u["current_logical_offset_in_pixels"]),
                              shape=tuple(int(v) for v in u["current_logical_shape_in_pixels"]),
                              scale=tuple(float(v) for v in u["scale_factor"])))
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
        frac = min(1.0, args.cpu_seconds * rate / max(1, counts["full"]["steps"]))
        stride = max(1, int(round(1.0 / frac)))
        nrows = -(-H // stride)
        sample = FrameRegion(0, 0, W, nrows, 1, stride)
        est = frac * counts["full"]["steps"] / rate                     # seconds for one pass over the sample
        reps = max(1, min(16, int(round(args.cpu_seconds / max(est, 1e-3))))) if stride == 1 else 1
        base = {}
        for mode in modes:
            m = dict(spec.material)
            m["lmip_threshold"] = float("inf") if mode == "full" else 0.5 * 255.0
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
            res = vol.render(cam, W, H, region=sample, count_steps=True)
            torch.cuda.synchronize()
            result["check"] = testing.compare(res, ref)

    if rank == 0 and collective and args.check:
        set_mode(True)
        frame_full = vol.render(cam, W, H, region=full_frame)
        torch.cuda.synchronize()
        set_mode(True)
        frame()
        drain()
        torch.cuda.synchronize()
        got = last_frame[0]
        got = got.to(frame_full.rgba.device)
        bad = (got != frame_full.rgba).any(dim=-1)
        result["check" if world > 1 else "check_gathered"] = {"gathered_frame_equals_single_gpu_render": bool(torch.equal(got, frame_full.rgba)),
                           "mismatched_pixels": int(bad.sum().item()),
                           "coloured_pixels_gathered": int((got[..., :3].abs().sum(-1) > 0).sum().item()),
                           "coloured_pixels_single": int((frame_full.rgba[..., :3].abs().sum(-1) > 0).sum().item()),
                           "alpha1_gathered": int((got[..., 3] == 1).sum().item()),
                           "alpha1_single": int((frame_full.rgba[..., 3] == 1).sum().item()),
                           "mismatched_rows": [int(v) for v in torch.nonzero(bad.any(dim=1)).flatten()[:12].tolist()]}
    elif collective and args.check:
        set_mode(True)
        frame()
        drain()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
