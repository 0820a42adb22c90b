"""Frame tiling across the GPUs of one node + gather of the per-rank RGBA bands.

The reference has no multi-GPU support at all.  The path
shards by pixels only: rays are independent (no inter-ray communication in
fs_main / raycast), while any ray may cross the whole volume, so the ring buffers
are replicated on every GPU and the frame is dealt to the ranks in interleaved
row bands (load balance: every rank sees every part of the image).  One process
per GPU; the only collective is one ``gather`` of the rendered bands to rank 0 per
frame (backend "nccl" == RCCL over xGMI on the GPU box; "gloo" on CPUs in the
tests).  Root has a direct xGMI link to every peer, so a plain gather is one hop
per peer — no ring algorithm is involved or wanted.
"""

from __future__ import annotations

import ctypes as C

from . import _native as N
from ._wobject import FrameRegion


class TiledFrame:
    """Row-band decomposition of a ``width x height`` frame over ``world`` ranks."""

    def __init__(self, width: int, height: int, rank: int, world: int, band_h: int = 16, force_collective: bool = False):
        if not (0 <= rank < world):
            raise ValueError("rank out of range")
        if band_h <= 0:
            raise ValueError("band_h must be positive")
        self.width, self.height, self.rank, self.world, self.band_h = width, height, rank, world, band_h
        # a world of one normally skips banding and the collective; `force_collective` keeps both (a
        # one-rank RCCL group exercises the whole N > 1 code path on a single GPU)
        self.collective = world > 1 or force_collective
        self.region = FrameRegion.stripes(width, height, rank, world, band_h) if self.collective else FrameRegion.full(width, height)
        self.rows_per_rank = self.region.out_h
        self._gathered = None
        self._frame = None

    def frame_rows_of(self, rank: int):
        """Frame row of every output row of ``rank`` (-1 for padding rows)."""
        rows = []
        for r in range(self.rows_per_rank):
            y = rank * self.band_h + (r // self.band_h) * self.band_h * self.world + r % self.band_h
            rows.append(y if y < self.height else -1)
        return rows

    def gather(self, local, dst: int = 0, volume=None):
        """Gather every rank's ``[rows_per_rank, width, C]`` band buffer on ``dst`` and un-tile it
        into the full ``[height, width, C]`` frame (returned on ``dst``, ``None`` elsewhere)."""
        import torch
        import torch.distributed as dist

        if not self.collective:
            return local
        if tuple(local.shape[:2]) != (self.rows_per_rank, self.width):
            raise ValueError(f"band buffer has shape {tuple(local.shape)}, expected ({self.rows_per_rank}, {self.width}, C)")
        if self.rank == dst:
            if self._gathered is None or self._gathered.shape[1:] != local.shape or self._gathered.dtype != local.dtype:
                self._gathered = torch.empty((self.world, *local.shape), dtype=local.dtype, device=local.device)
                self._frame = torch.empty((self.height, self.width, *local.shape[2:]), dtype=local.dtype, device=local.device)
            dist.gather(local, list(self._gathered.unbind(0)), dst=dst)
            return self.untile(self._gathered, self._frame, volume)
        dist.gather(local, None, dst=dst)
        return None

    # ---- frames in flight: slot-wise asynchronous gathers ------------------------------------------
    def gather_async(self, local, slot: int = 0, dst: int = 0, volume=None):
        """Start the gather of one frame's bands into buffer set ``slot`` and return at once.  The
        collective is ordered after the work already enqueued on the current stream (the render that
        wrote ``local``).  ``local`` must stay untouched until :meth:`finish` of the same slot."""
        import torch
        import torch.distributed as dist

        if not self.collective:
            self._slots()[slot] = ("local", local, volume)
            return
        if tuple(local.shape[:2]) != (self.rows_per_rank, self.width):
            raise ValueError(f"band buffer has shape {tuple(local.shape)}, expected ({self.rows_per_rank}, {self.width}, C)")
        slots = self._slots()
        if slots.get(slot) is not None:
            raise RuntimeError(f"slot {slot} still holds an unfinished gather")
        if self.rank == dst:
            bufs = self.__dict__.setdefault("_slot_bufs", {})
            b = bufs.get(slot)
            if b is None or b[0].shape[1:] != local.shape or b[0].dtype != local.dtype or b[0].device != local.device:
                b = (torch.empty((self.world, *local.shape), dtype=local.dtype, device=local.device),
                     torch.empty((self.height, self.width, *local.shape[2:]), dtype=local.dtype, device=local.device))
                bufs[slot] = b
            work = dist.gather(local, list(b[0].unbind(0)), dst=dst, async_op=True)
        else:
            work = dist.gather(local, None, dst=dst, async_op=True)
        slots[slot] = (work, local, volume)

    def finish(self, slot: int = 0, dst: int = 0):
        """Complete the gather started in ``slot``: the current stream waits for that collective, then
        the bands are un-tiled.  Returns the frame on ``dst`` (``None`` elsewhere or if the slot is idle)."""
        slots = self._slots()
        pending, slots[slot] = slots.get(slot), None
        if pending is None:
            return None
        work, local, volume = pending
        if isinstance(work, str):                     # no collective: the local buffer is the frame
            return local
        work.wait()                                   # the current stream waits for that collective only
        if self.rank != dst:
            return None
        gathered, out = self._slot_bufs[slot]
        return self.untile(gathered, out, volume)

    def _slots(self):
        return self.__dict__.setdefault("_slot_pending", {})

    def gather_pipelined(self, local, dst: int = 0, volume=None):
        """Like :meth:`gather`, but frame k's collective runs beside frame k+1's render: the gather is
        started asynchronously into one of two buffer sets and finished (wait + un-tile) one call later.
        Returns the PREVIOUS frame on ``dst`` (``None`` on the first call and on other ranks).  The
        caller must render successive frames into alternating band buffers.  Call :meth:`flush` at the end."""
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
        """``gathered[rank, r]`` -> ``out[frame_row]``.  On the GPU this is the ``svr_untile_stripes``
        kernel; CPU tensors (gloo tests) are permuted with torch indexing."""
        import torch

        if gathered.is_cuda:
            if volume is None:
                raise ValueError("un-tiling on the GPU needs the SubVolume that owns the device context")
            elem = gathered.element_size() * (gathered.shape[3] if gathered.dim() > 3 else 1)
            N.check(
                N.lib().svr_untile_stripes(
                    volume._rings.handle, C.c_void_p(gathered.data_ptr()), C.c_void_p(out.data_ptr()),
                    self.width, self.height, self.band_h, self.world, self.rows_per_rank, int(elem),
                    C.c_void_p(torch.cuda.current_stream(gathered.device).cuda_stream)),
                "svr_untile_stripes")
            return out
        for rank in range(self.world):
            rows = self.frame_rows_of(rank)
            src = [r for r, y in enumerate(rows) if y >= 0]
            dst = [y for y in rows if y >= 0]
            out[torch.tensor(dst, dtype=torch.long)] = gathered[rank][torch.tensor(src, dtype=torch.long)]
        return out
