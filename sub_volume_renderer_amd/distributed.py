"""Frame tiling across the GPUs of one node + gather of the per-rank regions.

The reference has no multi-GPU support at all.  The path shards by pixels only: rays are
independent (no inter-ray communication in fs_main / raycast), while any ray may cross the whole volume, so the
ring buffers are replicated on every GPU and the frame is dealt to the ranks either in interleaved row bands
(``tiling="rows"``, the default: every rank sees every part of the image, so the ranks' loads are even) or as
the literal grid of BASELINE config 3 (``tiling="2x4"``: rank ``ty * 2 + tx`` renders tile (tx, ty)).  One
process per GPU; the only exchange is "every rank's rendered planes -> rank 0" once per frame:

* device tensors: ``svr_gather_tiles`` (include/svr.h) — grouped ncclSend / ncclRecv on RCCL over xGMI, enqueued
  on the stream that carried the render, after ``init_comm``; without a communicator,
  ``torch.distributed.gather`` (backend "nccl" is the same RCCL);
* CPU tensors (the gloo tests): ``torch.distributed.gather``.

Root has a direct xGMI link to every peer, so the gather is one hop per peer — no ring algorithm is involved
or wanted.
"""

from __future__ import annotations

import ctypes as C

from . import _native as N
from ._wobject import FrameRegion


def _grid_for(world: int, tiling: str) -> tuple[int, int]:
    if tiling == "grid":                                  # the most square grid with gx <= gy
        gx = max(d for d in range(1, int(world ** 0.5) + 1) if world % d == 0)
        return gx, world // gx
    try:
        gx, gy = (int(v) for v in tiling.lower().split("x"))
    except ValueError:
        raise ValueError(f"tiling must be 'rows', 'grid' or '<gx>x<gy>', not {tiling!r}") from None
    if gx < 1 or gy < 1 or gx * gy != world:
        raise ValueError(f"a {gx}x{gy} grid needs {gx * gy} ranks, the world has {world}")
    return gx, gy


class _StreamEvent:
    """`wait()` like a torch.distributed work handle: the CURRENT stream waits for the event (no host block)."""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        import torch

        torch.cuda.current_stream(self.device).wait_event(self.event)


class TiledFrame:
    """Decomposition of a ``width x height`` frame over ``world`` ranks and the gather of its pieces."""

    def __init__(self, width: int, height: int, rank: int, world: int, band_h: int = 16, force_collective: bool = False,
                 tiling: str = "rows"):
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        if band_h <= 0:
            raise ValueError("band_h must be positive")
        self.width, self.height, self.rank, self.world, self.band_h = width, height, rank, world, band_h
        # a world of one normally skips tiling and the collective; `force_collective` keeps both (a
        # one-rank RCCL group exercises the whole N > 1 code path on a single GPU)
        self.collective = world > 1 or force_collective
        self.tiling = tiling
        self.grid = None
        if not self.collective:
            self.region = FrameRegion.full(width, height)
        elif tiling == "rows":
            self.region = FrameRegion.stripes(width, height, rank, world, band_h)
        else:
            gx, gy = _grid_for(world, tiling)
            tw, th = -(-width // gx), -(-height // gy)
            self.grid = (gx, gy, tw, th)
            self.region = FrameRegion.tile((rank % gx) * tw, (rank // gx) * th, tw, th)
        self.rows_per_rank = self.region.out_h
        self.cols_per_rank = self.region.out_w
        self._gathered = None
        self._frame = None
        self._comm_volume = None

    # ---- geometry -------------------------------------------------------------------------------
    def frame_rows_of(self, rank: int):
        """Frame row of every output row of ``rank`` (-1 for padding rows)."""
        rows = []
        for r in range(self.rows_per_rank):
            if self.grid is None:
                y = rank * self.band_h + (r // self.band_h) * self.band_h * self.world + r % self.band_h
            else:
                y = (rank // self.grid[0]) * self.grid[3] + r
            rows.append(y if y < self.height else -1)
        return rows

    def frame_cols_of(self, rank: int):
        """Frame column of every output column of ``rank`` (-1 for padding columns)."""
        x0 = 0 if self.grid is None else (rank % self.grid[0]) * self.grid[2]
        return [x0 + c if x0 + c < self.width else -1 for c in range(self.cols_per_rank)]

    def _check(self, t):
        if tuple(t.shape[:2]) != (self.rows_per_rank, self.cols_per_rank):
            raise ValueError(f"region buffer has shape {tuple(t.shape)}, expected "
                             f"({self.rows_per_rank}, {self.cols_per_rank}, ...)")

    # ---- RCCL communicator behind the C ABI ----------------------------------------------------------
    def init_comm(self, volume) -> bool:
        """Create the RCCL communicator of ``volume``'s device context (``svr_comm_init``) over the ranks of
        the default torch process group, which only carries the 128-byte id.  Then one self-check gather; if
        anything fails on any rank, EVERY rank falls back to ``torch.distributed.gather`` (returns False)."""
        import torch
        import torch.distributed as dist

        lib, handle = N.lib(), volume._rings.handle
        dev = torch.device("cuda", volume._rings.device)

        def all_ok(ok: int) -> bool:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        # the id: rank 0 makes it; a rank-0 failure travels as None, so that no rank enters ncclCommInitRank alone
        buf = C.create_string_buffer(128)
        made = self.rank != 0 or lib.svr_comm_unique_id(C.cast(buf, C.POINTER(C.c_char * 128)).contents) == 0
        box = [buf.raw if (self.rank == 0 and made) else None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            return False
        ident = (C.c_char * 128).from_buffer_copy(box[0])
        joined = lib.svr_comm_init(handle, ident, self.rank, self.world) == 0
        if not all_ok(int(joined)):                        # some rank has no communicator: nobody starts a send / recv on it
            if joined:
                lib.svr_comm_destroy(handle)
            return False
        ok = 1
        try:
            self._comm_volume = volume
            probe = torch.full((self.rows_per_rank, self.cols_per_rank, 1), float(self.rank + 1), device=dev)
            got = self.gather(probe, dst=0, volume=volume)
            torch.cuda.synchronize(dev)
            if self.rank == 0:
                for k in range(self.world):               # rank k's pixels hold k + 1
                    rows = [y for y in self.frame_rows_of(k) if y >= 0]
                    cols = [x for x in self.frame_cols_of(k) if x >= 0]
                    if not bool((got[rows][:, cols] == float(k + 1)).all()):
                        ok = 0
        except Exception:  # noqa: BLE001 - any failure here means "use the torch transport"
            ok = 0
        if not all_ok(ok):
            self._comm_volume = None
            return False
        return True

    @property
    def transport(self) -> str:
        return "svr_gather_tiles (RCCL send/recv behind the C ABI)" if self._comm_volume is not None else "torch.distributed.gather"

    # ---- one gather, blocking on the host only as far as the transport does -----------------------------
    def gather(self, local, dst: int = 0, volume=None):
        """Gather every rank's region buffer(s) on ``dst`` and un-tile them into full-frame tensor(s)
        (returned on ``dst``, ``None`` elsewhere).  ``local``: one ``[rows, cols, ...]`` tensor or a tuple of
        planes (RGBA, depth, label ...), all gathered in one collective."""
        self.gather_async(local, slot="_sync", dst=dst, volume=volume)
        return self.finish("_sync", dst=dst)

    # ---- frames in flight: slot-wise asynchronous gathers ------------------------------------------
    def gather_async(self, local, slot=0, dst: int = 0, volume=None):
        """Start the gather of one frame's region(s) into buffer set ``slot`` and return at once.  The
        collective is ordered after the work already enqueued on the current stream (the render that
        wrote ``local``).  ``local`` must stay untouched until :meth:`finish` of the same slot."""
        import torch
        import torch.distributed as dist

        single = not isinstance(local, (tuple, list))
        planes = [local] if single else list(local)
        slots = self._slots()
        if not self.collective:
            slots[slot] = ("local", planes, single, volume)
            return
        for t in planes:
            self._check(t)
        if slots.get(slot) is not None:
            raise RuntimeError(f"slot {slot} still holds an unfinished gather")
        bufs = self.__dict__.setdefault("_slot_bufs", {})
        b = bufs.get(slot)
        if self.rank == dst:
            stale = b is None or len(b) != len(planes) or any(
                g.shape[1:] != t.shape or g.dtype != t.dtype or g.device != t.device for (g, _), t in zip(b, planes))
            if stale:
                b = [(torch.empty((self.world, *t.shape), dtype=t.dtype, device=t.device),
                      torch.empty((self.height, self.width, *t.shape[2:]), dtype=t.dtype, device=t.device)) for t in planes]
                bufs[slot] = b
        comm_volume = self._comm_volume if planes[0].is_cuda else None
        if comm_volume is not None:
            n = len(planes)
            loc = (C.c_void_p * n)(*[t.data_ptr() for t in planes])
            gat = (C.c_void_p * n)(*[(b[i][0].data_ptr() if self.rank == dst else 0) for i in range(n)])
            nbytes = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in planes])
            stream = torch.cuda.current_stream(planes[0].device).cuda_stream
            N.check(N.lib().svr_gather_tiles(comm_volume._rings.handle, n, loc, gat, nbytes, dst, C.c_void_p(stream)),
                    "svr_gather_tiles")
            # the transfers are ordered on THIS stream only: an event behind them lets finish() run on any other stream
            # (root: before it un-tiles `gathered`; every rank: before the caller renders into `local` again)
            events = self.__dict__.setdefault("_slot_events", {})
            ev = events.get(slot)
            if ev is None:
                ev = events[slot] = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(planes[0].device))
            works = [_StreamEvent(ev, planes[0].device)]
        else:
            works = [dist.gather(t, list(b[i][0].unbind(0)) if self.rank == dst else None, dst=dst, async_op=True)
                     for i, t in enumerate(planes)]
        slots[slot] = (works, planes, single, volume)

    def finish(self, slot=0, dst: int = 0):
        """Complete the gather started in ``slot``: the current stream waits for that collective, then the
        regions are un-tiled.  Returns the frame(s) on ``dst`` (``None`` elsewhere or if the slot is idle)."""
        slots = self._slots()
        pending, slots[slot] = slots.get(slot), None
        if pending is None:
            return None
        works, planes, single, volume = pending
        if isinstance(works, str):                    # no collective: the local buffers are the frame
            return planes[0] if single else tuple(planes)
        for w in works:
            w.wait()                                  # the current stream waits for that collective only (both transports,
        if self.rank != dst:                          # root and non-root alike: `local` may be rendered into again after this)
            return None
        frames = [self.untile(g, out, volume or self._comm_volume) for g, out in self._slot_bufs[slot]]
        return frames[0] if single else tuple(frames)

    def _slots(self):
        return self.__dict__.setdefault("_slot_pending", {})

    def gather_pipelined(self, local, dst: int = 0, volume=None):
        """Like :meth:`gather`, but frame k's collective runs beside frame k+1's render: the gather is
        started asynchronously into one of two buffer sets and finished (wait + un-tile) one call later.
        Returns the PREVIOUS frame on ``dst`` (``None`` on the first call and on other ranks).  The
        caller must render successive frames into alternating region buffers.  Call :meth:`flush` at the end."""
        if not self.collective:
            return local
        k = self._pipe_k = getattr(self, "_pipe_k", -1) + 1
        self.gather_async(local, slot=k & 1, dst=dst, volume=volume)
        return self.finish(slot=(k - 1) & 1, dst=dst) if k > 0 else None

    def flush(self, dst: int = 0):
        """Finish the last pipelined gather; returns the last frame on ``dst``."""
        k = getattr(self, "_pipe_k", -1)
        return self.finish(slot=k & 1, dst=dst) if k >= 0 else None

    def untile(self, gathered, out, volume=None):
        """``gathered[rank, r, c]`` -> ``out[frame_row, frame_col]``.  On the GPU this is the
        ``svr_untile_stripes`` / ``svr_untile_grid`` kernel; CPU tensors (gloo tests) are permuted with torch
        indexing."""
        import torch

        if gathered.is_cuda:
            if volume is None:
                raise ValueError("un-tiling on the GPU needs the SubVolume that owns the device context")
            elem = gathered.element_size() * (gathered.shape[3] if gathered.dim() > 3 else 1)
            stream = C.c_void_p(torch.cuda.current_stream(gathered.device).cuda_stream)
            if self.grid is None:
                N.check(
                    N.lib().svr_untile_stripes(
                        volume._rings.handle, C.c_void_p(gathered.data_ptr()), C.c_void_p(out.data_ptr()),
                        self.width, self.height, self.band_h, self.world, self.rows_per_rank, int(elem), stream),
                    "svr_untile_stripes")
            else:
                gx, gy, tw, th = self.grid
                N.check(
                    N.lib().svr_untile_grid(
                        volume._rings.handle, C.c_void_p(gathered.data_ptr()), C.c_void_p(out.data_ptr()),
                        self.width, self.height, tw, th, gx, gy, int(elem), stream),
                    "svr_untile_grid")
            return out
        for rank in range(self.world):
            rows, cols = self.frame_rows_of(rank), self.frame_cols_of(rank)
            src_r = torch.tensor([r for r, y in enumerate(rows) if y >= 0], dtype=torch.long)
            dst_r = torch.tensor([y for y in rows if y >= 0], dtype=torch.long)
            src_c = torch.tensor([c for c, x in enumerate(cols) if x >= 0], dtype=torch.long)
            dst_c = torch.tensor([x for x in cols if x >= 0], dtype=torch.long)
            out[dst_r[:, None], dst_c[None, :]] = gathered[rank][src_r[:, None], src_c[None, :]]
        return out
