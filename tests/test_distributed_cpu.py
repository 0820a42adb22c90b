"""N > 1 path on CPUs: world_size-2 and -3 `gloo` process groups.  Each rank renders ITS row
bands (with the oracle standing in for the GPU kernel — the GPU tests prove kernel == oracle
on exactly such regions), the bands are gathered on rank 0 through the product's TiledFrame,
un-tiled, and must equal the full frame bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lmip
from sub_volume_renderer_amd import FrameRegion, testing
from sub_volume_renderer_amd.distributed import TiledFrame


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, band_h, W, H, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = testing.synthetic_spec(32, W, H, threshold=0.3)
        tf = TiledFrame(W, H, rank, world, band_h)
        ref = lmip.render_spec(spec, region=tf.region, nthreads=2)
        local = torch.from_numpy(ref.rgba)
        frame = tf.gather(local, dst=0)
        labels = tf.gather(torch.from_numpy(ref.label.astype(np.int64))[..., None], dst=0)
        if rank == 0:
            np.savez(out_path, rgba=frame.numpy(), label=labels.numpy()[..., 0])
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,band_h", [(2, 16), (3, 8), (2, 7)])
def test_gather_of_row_bands_equals_full_frame(tmp_path, world, band_h):
    W, H = 64, 45                      # height not a multiple of the band: padding rows exist
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), band_h, W, H, out), nprocs=world, join=True)
    got = np.load(out)
    full = lmip.render_spec(testing.synthetic_spec(32, W, H, threshold=0.3), nthreads=2)
    np.testing.assert_array_equal(got["rgba"], full.rgba)
    np.testing.assert_array_equal(got["label"], full.label.astype(np.int64))
    assert np.count_nonzero(full.flags == 2) > 0


def test_band_partition_is_exact():
    for (W, H, world, bh) in [(1920, 1080, 8, 16), (1920, 1080, 4, 8), (64, 45, 3, 8), (10, 5, 2, 16)]:
        seen = np.zeros(H, int)
        for rank in range(world):
            tf = TiledFrame(W, H, rank, world, bh)
            assert tf.rows_per_rank == TiledFrame(W, H, 0, world, bh).rows_per_rank     # equal counts for gather
            reg = tf.region
            for r, y in enumerate(tf.frame_rows_of(rank)):
                yy = reg.y0 + (r // reg.band_h) * reg.band_pitch + r % reg.band_h        # svr_frame mapping
                assert (y == yy) or (y == -1 and yy >= H)
                if y >= 0:
                    seen[y] += 1
        assert np.all(seen == 1)
    assert TiledFrame(8, 8, 0, 1).region == FrameRegion.full(8, 8)


def _pipe_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, bh = 24, 21, 4
        tf = TiledFrame(W, H, rank, world, bh)
        rows = tf.frame_rows_of(rank)
        frames = []
        for k in range(4):                     # frame k: pixel value = 1000*k + 24*row + col (padding rows: -1)
            band = torch.full((tf.rows_per_rank, W, 1), -1.0)
            for r, y in enumerate(rows):
                if y >= 0:
                    band[r, :, 0] = 1000.0 * k + 24.0 * y + torch.arange(W, dtype=torch.float32)
            f = tf.gather_pipelined(band, dst=0)
            if rank == 0:
                frames.append(None if f is None else f.clone())
        last = tf.flush(dst=0)
        if rank == 0:
            frames.append(last.clone())
            np.savez(out_path, **{f"f{i}": (np.zeros(0) if f is None else f.numpy()) for i, f in enumerate(frames)})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_returns_previous_frame(tmp_path):
    """gather_pipelined overlaps frame k's collective with frame k+1's render: call k returns frame k-1."""
    out = str(tmp_path / "pipe.npz")
    mp.spawn(_pipe_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    assert got["f0"].size == 0                                   # nothing finished yet on the first call
    yy, xx = np.meshgrid(np.arange(21), np.arange(24), indexing="ij")
    for i in range(1, 5):                                        # call i returns frame i-1; flush returns frame 3
        np.testing.assert_array_equal(got[f"f{i}"][..., 0], 1000.0 * (i - 1) + 24.0 * yy + xx)


def _slots_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H, bh, depth = 16, 13, 2, 2
        tf = TiledFrame(W, H, rank, world, bh)
        rows = tf.frame_rows_of(rank)
        done = {}
        bands = [torch.empty((tf.rows_per_rank, W, 1)) for _ in range(depth)]   # one band buffer per slot
        for k in range(5):                     # bench.py's loop: finish frame k - depth, render k, start its gather
            slot = k % depth
            f = tf.finish(slot, dst=0)
            if rank == 0 and k >= depth:
                done[k - depth] = f.clone()
            else:
                assert f is None
            band = bands[slot]
            band.fill_(-1.0)
            for r, y in enumerate(rows):
                if y >= 0:
                    band[r, :, 0] = 1000.0 * k + 16.0 * y + torch.arange(W, dtype=torch.float32)
            tf.gather_async(band, slot, dst=0)
            with pytest.raises(RuntimeError):
                tf.gather_async(band, slot, dst=0)                 # the slot is busy until finish()
        for k in (3, 4):                       # drain, oldest first
            f = tf.finish(k % depth, dst=0)
            if rank == 0:
                done[k] = f.clone()
        assert tf.finish(0) is None and tf.finish(1) is None
        if rank == 0:
            np.savez(out_path, **{f"f{k}": v.numpy() for k, v in done.items()})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_frames_in_flight_slots(tmp_path):
    """gather_async / finish: two gathers in flight in separate slots (the multi-GPU bench loop)."""
    out = str(tmp_path / "slots.npz")
    mp.spawn(_slots_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    yy, xx = np.meshgrid(np.arange(13), np.arange(16), indexing="ij")
    for k in range(5):
        np.testing.assert_array_equal(got[f"f{k}"][..., 0], 1000.0 * k + 16.0 * yy + xx)


# ---- config 3's grid tiling ("2x4") and several planes per gather ---------------------------------------------------
def test_grid_partition_is_exact():
    for (W, H, world, tiling, want) in [(1920, 1080, 8, "2x4", (960, 270)), (1920, 1080, 8, "grid", (960, 270)),
                                        (1920, 1080, 4, "2x2", (960, 540)), (65, 45, 6, "3x2", (22, 23)),
                                        (10, 7, 2, "1x2", (10, 4))]:
        seen = np.zeros((H, W), int)
        for rank in range(world):
            tf = TiledFrame(W, H, rank, world, tiling=tiling)
            assert (tf.cols_per_rank, tf.rows_per_rank) == want
            reg = tf.region
            assert (reg.out_w, reg.out_h, reg.band_h) == (want[0], want[1], want[1])
            rows, cols = tf.frame_rows_of(rank), tf.frame_cols_of(rank)
            assert rows[0] == reg.y0 and cols[0] == reg.x0           # svr_frame mapping of a plain tile
            ys = [y for y in rows if y >= 0]
            xs = [x for x in cols if x >= 0]
            seen[np.ix_(ys, xs)] += 1
        assert np.all(seen == 1)
    # BASELINE config 3: rank ty * 2 + tx renders the 960 x 270 tile (tx, ty)
    tf = TiledFrame(1920, 1080, 5, 8, tiling="2x4")
    assert (tf.region.x0, tf.region.y0) == (960, 540)
    with pytest.raises(ValueError):
        TiledFrame(64, 64, 0, 8, tiling="3x3")
    with pytest.raises(ValueError):
        TiledFrame(64, 64, 0, 8, tiling="diagonal")


def _grid_worker(rank, world, port, tiling, W, H, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = testing.synthetic_spec(32, W, H, threshold=0.3)
        tf = TiledFrame(W, H, rank, world, tiling=tiling)
        ref = lmip.render_spec(spec, region=tf.region, nthreads=2)
        planes = (torch.from_numpy(ref.rgba), torch.from_numpy(ref.depth), torch.from_numpy(ref.label.astype(np.int64)),
                  torch.from_numpy(ref.flags))
        frames = tf.gather(planes, dst=0)                 # RGBA + depth + label + flags in one call
        if rank == 0:
            np.savez(out_path, rgba=frames[0].numpy(), depth=frames[1].numpy(), label=frames[2].numpy(), flags=frames[3].numpy())
        else:
            assert frames is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tiling", [(2, "2x1"), (4, "2x2"), (3, "grid")])
def test_gather_of_grid_tiles_and_all_planes_equals_full_frame(tmp_path, world, tiling):
    W, H = 66, 45                      # neither extent divides evenly: edge tiles hang over the frame
    out = str(tmp_path / "grid.npz")
    mp.spawn(_grid_worker, args=(world, _free_port(), tiling, W, H, out), nprocs=world, join=True)
    got = np.load(out)
    full = lmip.render_spec(testing.synthetic_spec(32, W, H, threshold=0.3), nthreads=2)
    np.testing.assert_array_equal(got["rgba"], full.rgba)
    np.testing.assert_array_equal(got["depth"], full.depth)
    np.testing.assert_array_equal(got["label"], full.label.astype(np.int64))
    np.testing.assert_array_equal(got["flags"], full.flags)
