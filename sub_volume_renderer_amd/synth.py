"""Deterministic synthetic multi-LOD volumes.

Closed-form and integer-only, so any block of any LOD can be generated on its
own (needed for the 4096^3 streaming config) and the numpy (host) and torch
(device) generators agree bit for bit.  No data files are shipped: the volumes
are regenerated wherever the tests / bench run.

LOD 0 density (uint8) = product of three incommensurate triangle waves
(sparse bright blobs, so LMIP rays hit at varied depths) + 4 bits of hashed
noise.  Labels (uint32) = hashed id of the enclosing 32^3 block, 0 where the
density is < 32.  LOD k = 2x mean-pool of the density (floor) and 2x max-pool of
the labels — the pooling rules of the reference's pyramid builders
(scripts/create_mouse_multiscale.py:23-54, scripts/create_platynereis_multiscale.py:86-134).
"""

from __future__ import annotations

import numpy as np

_M32 = 0xFFFFFFFF


def _periods(n: int) -> tuple[int, int, int]:
    s = n / 1024.0
    return tuple(max(8, int(round(p * s))) for p in (389, 521, 647))


def _tri(u, p, xp):
    m = u % p
    return xp.minimum(m, p - m) * 160 // p          # 0 .. 80


def _hash3(a0, a1, a2):
    h = (a0 * 73856093) ^ (a1 * 19349663) ^ (a2 * 83492791)
    h = h & _M32
    h = h ^ (h >> 13)
    h = (h * 0x5BD1E995) & _M32
    h = h ^ (h >> 15)
    return h


def _coords(off, shape, xp, device=None):
    if xp is np:
        ax = [np.arange(o, o + s, dtype=np.int64) for o, s in zip(off, shape)]
        return ax[0][:, None, None], ax[1][None, :, None], ax[2][None, None, :]
    ax = [xp.arange(o, o + s, dtype=xp.int64, device=device) for o, s in zip(off, shape)]
    return ax[0][:, None, None], ax[1][None, :, None], ax[2][None, None, :]


def lod0_block(n: int, off, shape, n_labels: int = 4096, xp=np, device=None):
    """(density uint8, labels uint32-as-int64) of LOD 0 for the box [off, off+shape) of an n^3 volume."""
    a0, a1, a2 = _coords(off, shape, xp, device)
    p0, p1, p2 = _periods(n)
    t0 = _tri(3 * a0 + a1 + 2 * a2, p0, xp)
    t1 = _tri(a0 + 4 * a1 + 2 * a2, p1, xp)
    t2 = _tri(2 * a0 + a1 + 5 * a2, p2, xp)
    dens = (t0 * t1 * t2 * 3) // 6400 + (_hash3(a0, a1, a2) & 15)          # 0 .. 255
    nb = (n + 31) // 32
    bid = ((a0 >> 5) * nb + (a1 >> 5)) * nb + (a2 >> 5)
    lab = ((bid * 2654435761) & _M32) % n_labels
    lab = xp.where(dens < 32, xp.zeros_like(lab), lab)
    return dens, lab


def _pool(dens, lab, xp):
    s0, s1, s2 = dens.shape
    d = dens.reshape(s0 // 2, 2, s1 // 2, 2, s2 // 2, 2)
    l = lab.reshape(s0 // 2, 2, s1 // 2, 2, s2 // 2, 2)
    if xp is np:
        return d.sum(axis=(1, 3, 5)) // 8, l.max(axis=(1, 3, 5))
    d = d.sum(dim=(1, 3, 5)) // 8
    l = l.amax(dim=(1, 3, 5))
    return d, l


_host_lib = None


def host_lib():
    """``libsvr_synth.so`` (csrc/synth_host.c): the same closed form, fused and multi-threaded; ``None`` when
    it has not been built (``__graft_entry__.build_synth``) or ``SVR_SYNTH_NUMPY`` is set."""
    global _host_lib
    if _host_lib is None:
        import ctypes as C
        import os

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsvr_synth.so")
        if os.environ.get("SVR_SYNTH_NUMPY") or not os.path.exists(path):
            _host_lib = False
        else:
            lib = C.CDLL(path)
            lib.svr_synth_block.restype = C.c_int
            lib.svr_synth_block.argtypes = [C.c_int64, C.c_int, C.c_int64 * 3, C.c_int64 * 3, C.c_int64, C.c_int64 * 3,
                                            C.c_void_p, C.c_void_p, C.c_int]
            _host_lib = lib
    return _host_lib or None


def block_host(n: int, lod: int, off, shape, n_labels: int = 4096, nthreads: int = 0, want=(True, True)):
    """``block`` through the fused host generator: (uint8 | None, uint32 | None) numpy arrays."""
    import ctypes as C

    lib = host_lib()
    if lib is None:
        raise RuntimeError("libsvr_synth.so has not been built")
    I3 = C.c_int64 * 3
    dens = np.empty(tuple(shape), np.uint8) if want[0] else None
    lab = np.empty(tuple(shape), np.uint32) if want[1] else None
    rc = lib.svr_synth_block(n, lod, I3(*[int(v) for v in off]), I3(*[int(v) for v in shape]), int(n_labels),
                             I3(*_periods(n)), dens.ctypes.data if want[0] else None,
                             lab.ctypes.data if want[1] else None, int(nthreads))
    if rc != 0:
        raise ValueError("svr_synth_block: argument out of range")
    return dens, lab


def block(n: int, lod: int, off, shape, n_labels: int = 4096, xp=np, device=None):
    """Block [off, off+shape) of LOD ``lod`` (extent n >> lod) as (uint8, uint32) arrays.

    With ``xp=torch`` the result is (torch.uint8, torch.int32 holding the u32 bit
    pattern) on ``device``.
    """
    f = 1 << lod
    dens, lab = lod0_block(n, [o * f for o in off], [s * f for s in shape], n_labels, xp, device)
    for _ in range(lod):
        dens, lab = _pool(dens, lab, xp)
    if xp is np:
        return dens.astype(np.uint8), lab.astype(np.uint32)
    return dens.to(xp.uint8), lab.to(xp.int32)  # labels < 2^31 here (n_labels is small)


def volume(n: int, lod: int, n_labels: int = 4096, xp=np, device=None, slab: int = 64):
    """Whole LOD as two arrays, generated slab by slab to bound temporaries."""
    m = n >> lod
    slab = min(slab, m)
    if xp is np:
        dens = np.empty((m, m, m), np.uint8)
        lab = np.empty((m, m, m), np.uint32)
    else:
        dens = xp.empty((m, m, m), dtype=xp.uint8, device=device)
        lab = xp.empty((m, m, m), dtype=xp.int32, device=device)
    for z in range(0, m, slab):
        h = min(slab, m - z)
        d, l = block(n, lod, (z, 0, 0), (h, m, m), n_labels, xp, device)
        dens[z:z + h] = d
        lab[z:z + h] = l
    return dens, lab


class LazyLod:
    """A numpy-like view of one LOD that generates blocks on demand (config C4:
    a 4096^3 volume is never resident).  ``labels=True`` selects the label array."""

    # threads of the host generator per read: half of what the process may use, so that a streaming worker calling
    # this beside a render loop does not exhaust the process's CPU quota (a throttled render thread stalls frames)
    threads = 0

    def __init__(self, n: int, lod: int, labels: bool, n_labels: int = 4096):
        self.n, self.lod, self.labels, self.n_labels = n, lod, labels, n_labels
        if not LazyLod.threads:
            import os

            cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:
                with open("/sys/fs/cgroup/cpu.max") as f:
                    q, per = f.read().split()[:2]
                if q != "max":
                    cpus = min(cpus, max(1, int(q) // int(per)))
            except (OSError, ValueError):
                pass
            LazyLod.threads = max(1, min(8, cpus // 2))
        m = n >> lod
        self.shape = (m, m, m)
        self.ndim = 3
        self.dtype = np.dtype(np.uint32 if labels else np.uint8)

    def __getitem__(self, slices):
        import time

        off = [s.start or 0 for s in slices]
        shape = [(s.stop if s.stop is not None else dim) - o for s, o, dim in zip(slices, off, self.shape)]
        t = time.perf_counter()
        if host_lib() is not None:
            d, l = block_host(self.n, self.lod, off, shape, self.n_labels, nthreads=LazyLod.threads,
                              want=(not self.labels, self.labels))
        else:
            d, l = block(self.n, self.lod, off, shape, self.n_labels)
        out = l if self.labels else d
        LazyLod.read_seconds += time.perf_counter() - t          # the "store read" share of a streaming run
        LazyLod.read_bytes += out.nbytes
        return out

    read_seconds = 0.0
    read_bytes = 0
