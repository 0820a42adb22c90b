"""A small reader (and test writer) for zarr v3 stores on a local file system.

The reference feeds ``SubVolume`` with ``zarr.Array`` objects (README.md:18, scripts/mouse.py:32-47) read from
the stores its pyramid builders write: zarr v3 groups ``raw.zarr`` / ``labels.zarr`` with arrays ``scale0`` ..
``scale4``, 16^3 chunks inside 64^3 shards (scripts/create_mouse_multiscale.py:102-131; the ``ShardingCodec`` with
blosc built at :115-118 is discarded, so the arrays get zarr-python 3's default inner codecs, ``bytes`` + ``zstd``).
Neither zarr nor tensorstore is installed on the target and nothing can be installed, so this module reads that
on-disk format directly: what ``load_into_buffer`` needs of a backing array is ``shape``, ``dtype``, ``ndim``,
``chunks`` and ``__getitem__`` over slices (``_wrapping_buffer.py:283-335``).

Supported (zarr v3 core spec + the ``sharding_indexed`` / ``gzip`` / ``zstd`` / ``crc32c`` / ``transpose`` codec
specs): regular chunk grids; ``default`` and ``v2`` chunk key encodings; fill values for missing chunks; codec
chains ``[transpose] -> bytes | sharding_indexed -> [gzip | zstd | crc32c]*``; shards with the index at the end or
the start, read chunk-wise (only the index and the inner chunks a request touches are read from the file).
``blosc`` is refused with a clear error: no blosc library exists here (zstd comes from the system's libzstd
through ctypes; gzip from zlib).  A ``crc32c`` checksum is verified for every payload, whatever its size
(``csrc/host_codecs.c`` when built, a Python loop otherwise); a zstd frame that claims more bytes than its chunk
can hold is refused before anything is allocated.  The format knowledge is restated from the published zarr v3 specification; no
fixture written by zarr-python is available offline, so interoperability is *unpinned* — ``tests/test_zarr3.py``
pins the reader against hand-assembled byte strings of the spec's layout and against this module's own writer.
"""

from __future__ import annotations

import ctypes
import ctypes.util
import json
import os
import struct
import time
import zlib
from collections import OrderedDict
from itertools import product

import numpy as np

_DTYPES = {"bool": "?", "int8": "i1", "int16": "i2", "int32": "i4", "int64": "i8", "uint8": "u1", "uint16": "u2",
           "uint32": "u4", "uint64": "u8", "float16": "f2", "float32": "f4", "float64": "f8"}
_EMPTY = 2 ** 64 - 1


# ---------------------------------------------------------------------------------------------------
# bytes -> bytes codecs
# ---------------------------------------------------------------------------------------------------
_zstd = None


def _libzstd():
    global _zstd
    if _zstd is None:
        name = ctypes.util.find_library("zstd")
        if not name:
            raise RuntimeError("the zstd codec needs libzstd, which was not found on this machine")
        lib = ctypes.CDLL(name)
        lib.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
        lib.ZSTD_getFrameContentSize.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        lib.ZSTD_decompress.restype = ctypes.c_size_t
        lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
        lib.ZSTD_compress.restype = ctypes.c_size_t
        lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        lib.ZSTD_compressBound.restype = ctypes.c_size_t
        lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
        lib.ZSTD_isError.restype = ctypes.c_uint
        lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
        _zstd = lib
    return _zstd


def _zstd_decompress(data: bytes, expected: int | None) -> bytes:
    lib = _libzstd()
    size = lib.ZSTD_getFrameContentSize(data, len(data))
    if size >= 2 ** 64 - 2:                         # unknown / error: fall back to what the chunk must hold
        if expected is None:
            raise ValueError("zstd frame without a content size")
        size = expected
    elif expected is not None and size > expected:   # never allocate what a (corrupt) frame header asks for
        raise ValueError(f"zstd frame claims {size} bytes, the chunk holds {expected}")
    out = ctypes.create_string_buffer(int(size) or 1)
    n = lib.ZSTD_decompress(out, int(size), data, len(data))
    if lib.ZSTD_isError(n):
        raise ValueError("corrupt zstd frame")
    return out.raw[:n]


def _zstd_compress(data: bytes, level: int = 3) -> bytes:
    lib = _libzstd()
    bound = lib.ZSTD_compressBound(len(data))
    out = ctypes.create_string_buffer(bound)
    n = lib.ZSTD_compress(out, bound, data, len(data), int(level))
    if lib.ZSTD_isError(n):
        raise ValueError("zstd compression failed")
    return out.raw[:n]


_CRC32C_TABLE = None
_codec_lib = False          # False: not looked for yet; None: not built


def _host_codecs():
    """``csrc/libsvr_hostcodec.so`` (host_codecs.c, built by ``__graft_entry__.build()``), or ``None``."""
    global _codec_lib
    if _codec_lib is False:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsvr_hostcodec.so")
        _codec_lib = None
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.svr_crc32c.restype = ctypes.c_uint32
            lib.svr_crc32c.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
            vp, i32p, u64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)
            lib.svr_zarr_decode_chunks.restype = ctypes.c_int
            lib.svr_zarr_decode_chunks.argtypes = [ctypes.c_int, vp, u64p, u64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, vp,
                                                   ctypes.POINTER(ctypes.c_int64), i32p, i32p, vp, ctypes.c_int]
            lib.svr_zarr_read_groups.restype = ctypes.c_int
            lib.svr_zarr_read_groups.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), i32p, i32p, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32p, vp,
                                                 ctypes.POINTER(ctypes.c_int64), i32p, i32p, vp, ctypes.c_int, u64p,
                                                 ctypes.POINTER(ctypes.c_int)]
            lib.svr_zarr_encode_bound.restype = ctypes.c_size_t
            lib.svr_zarr_encode_bound.argtypes = [ctypes.c_size_t, ctypes.c_int]
            lib.svr_zarr_encode_chunks.restype = ctypes.c_int
            lib.svr_zarr_encode_chunks.argtypes = [ctypes.c_int, vp, i32p, i32p, ctypes.c_int, i32p, vp, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, u64p, ctypes.c_int]
            _codec_lib = lib
    return _codec_lib


def _crc32c_python(data: bytes) -> int:
    global _CRC32C_TABLE
    if _CRC32C_TABLE is None:
        table = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            table.append(c)
        _CRC32C_TABLE = table
    crc = 0xFFFFFFFF
    for b in data:
        crc = _CRC32C_TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli), as the ``crc32c`` codec appends it (little endian).  Every payload is summed, whatever
    its size: by ``host_codecs.c`` when it has been built, else by a (slow) Python loop."""
    lib = _host_codecs()
    if lib is not None:
        return int(lib.svr_crc32c(bytes(data), len(data), 0))
    return _crc32c_python(data)


def _decode_bytes_codecs(codecs, data: bytes, expected: int | None) -> bytes:
    for c in reversed(codecs):
        name = c["name"]
        if name == "gzip":
            data = zlib.decompress(data, 16 + zlib.MAX_WBITS)
        elif name == "zstd":
            data = _zstd_decompress(data, expected)
        elif name == "crc32c":
            body, tail = data[:-4], data[-4:]
            if len(data) < 4 or struct.unpack("<I", tail)[0] != crc32c(body):
                raise ValueError("crc32c mismatch")
            data = body
        elif name == "blosc":
            raise NotImplementedError("the blosc codec cannot be decoded here: no blosc library is installed "
                                      "(gzip, zstd and uncompressed chunks are supported)")
        else:
            raise NotImplementedError(f"bytes-to-bytes codec {name!r} is not supported")
    return data


def _encode_bytes_codecs(codecs, data: bytes) -> bytes:
    for c in codecs:
        name = c["name"]
        if name == "gzip":
            co = zlib.compressobj(int(c.get("configuration", {}).get("level", 5)), zlib.DEFLATED, 16 + zlib.MAX_WBITS)
            data = co.compress(data) + co.flush()
        elif name == "zstd":
            data = _zstd_compress(data, int(c.get("configuration", {}).get("level", 3)))
        elif name == "crc32c":
            data = data + struct.pack("<I", crc32c(data))
        else:
            raise NotImplementedError(f"cannot write codec {name!r}")
    return data


# ---------------------------------------------------------------------------------------------------
# codec chains
# ---------------------------------------------------------------------------------------------------
class _Chain:
    """``[array->array]* , array->bytes , [bytes->bytes]*`` of one ``codecs`` list."""

    def __init__(self, codecs, dtype: np.dtype, chunk_shape):
        self.dtype, self.chunk_shape = dtype, tuple(chunk_shape)
        self.transposes, self.tail = [], []
        self.kind, self.endian, self.shard = None, "<", None
        for c in codecs:
            name, conf = c["name"], c.get("configuration", {})
            if self.kind is None and name == "transpose":
                self.transposes.append(tuple(conf["order"]))
            elif self.kind is None and name == "bytes":
                self.kind = "bytes"
                self.endian = ">" if conf.get("endian", "little") == "big" else "<"
            elif self.kind is None and name == "sharding_indexed":
                self.kind = "shard"
                inner = tuple(conf["chunk_shape"])
                self.shard = dict(inner=inner, chain=_Chain(conf["codecs"], dtype, inner),
                                  index_codecs=conf.get("index_codecs", [{"name": "bytes", "configuration": {"endian": "little"}},
                                                                         {"name": "crc32c"}]),
                                  at_end=conf.get("index_location", "end") == "end")
            elif self.kind is not None:
                self.tail.append(c)
            else:
                raise NotImplementedError(f"array-to-array codec {name!r} is not supported")
        if self.kind is None:
            raise ValueError("a codec chain needs exactly one array-to-bytes codec (bytes or sharding_indexed)")
        shape = self.chunk_shape
        for order in self.transposes:
            shape = tuple(shape[a] for a in order)
        self.stored_shape = shape                     # shape of the array the array->bytes codec sees

    def decode_block(self, data: bytes) -> np.ndarray:
        """A whole (non-sharded) chunk from its stored bytes."""
        n = int(np.prod(self.stored_shape)) * self.dtype.itemsize
        data = _decode_bytes_codecs(self.tail, data, n)
        a = np.frombuffer(data, self.dtype.newbyteorder(self.endian), count=n // self.dtype.itemsize).reshape(self.stored_shape)
        for order in reversed(self.transposes):
            a = a.transpose(np.argsort(order))
        return a.astype(self.dtype, copy=False)

    def encode_block(self, a: np.ndarray) -> bytes:
        for order in self.transposes:
            a = a.transpose(order)
        raw = np.ascontiguousarray(a, self.dtype.newbyteorder(self.endian)).tobytes()
        return _encode_bytes_codecs(self.tail, raw)


def _native_tail(chain: "_Chain"):
    """(zstd, crc) when the chunk codec chain is one ``host_codecs.c`` decodes — ``bytes`` (little endian, or one-byte
    elements), then optionally ``zstd``, then optionally ``crc32c`` — else ``None`` (the Python path handles the rest)."""
    if chain.kind != "bytes" or chain.transposes or (chain.endian == ">" and chain.dtype.itemsize > 1):
        return None
    names = [c["name"] for c in chain.tail]
    if names not in ([], ["zstd"], ["crc32c"], ["zstd", "crc32c"]):
        return None
    return ("zstd" in names, "crc32c" in names)


def _threads(reading: bool = True) -> int:
    """Worker threads of the native codec: the CPUs this process may use (affinity mask capped by the cgroup quota);
    ``SVR_ZARR_THREADS`` overrides.  The pool's workers sleep between requests, so a full-width team beside a render
    thread does not get the process throttled the way a spinning OpenMP team did (config 4 on the one-GPU box, 8 / 12 /
    16 threads: 4.5 / 4.9 / 5.4 GB/s delivered, frame-time tail unchanged: ``profiles/r03/host_thread_pinning.txt``)."""
    if os.environ.get("SVR_ZARR_THREADS"):
        return max(1, int(os.environ["SVR_ZARR_THREADS"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(32, n))


class ZarrV3Array:
    """Read-only zarr v3 array: ``shape``, ``dtype``, ``ndim``, ``chunks`` (inner chunks when sharded, as
    zarr-python reports them), ``shards`` and ``__getitem__`` over slices / integers."""

    def __init__(self, path: str, cache_chunks: int = 256):
        with open(os.path.join(path, "zarr.json")) as f:
            meta = json.load(f)
        if meta.get("zarr_format") != 3 or meta.get("node_type") != "array":
            raise ValueError(f"{path} is not a zarr v3 array")
        self.path, self.meta = path, meta
        self.shape = tuple(int(v) for v in meta["shape"])
        self.ndim = len(self.shape)
        try:
            self.dtype = np.dtype(_DTYPES[meta["data_type"]])
        except (KeyError, TypeError):
            raise NotImplementedError(f"data_type {meta['data_type']!r} is not supported") from None
        grid = meta["chunk_grid"]
        if grid["name"] != "regular":
            raise NotImplementedError("only regular chunk grids are supported")
        self._outer = tuple(int(v) for v in grid["configuration"]["chunk_shape"])
        enc = meta.get("chunk_key_encoding", {"name": "default"})
        self._v2 = enc["name"] == "v2"
        self._sep = enc.get("configuration", {}).get("separator", "." if self._v2 else "/")
        self.fill_value = self._fill(meta.get("fill_value", 0))
        self._chain = _Chain(meta["codecs"], self.dtype, self._outer)
        sh = self._chain.shard
        self.shards = self._outer if sh else None
        self.chunks = sh["inner"] if sh else self._outer
        self.attrs = meta.get("attributes", {})
        self._cache: OrderedDict = OrderedDict()
        self._cache_chunks = cache_chunks
        self._index_cache: OrderedDict = OrderedDict()
        # the native reader (csrc/host_codecs.c): 3-D arrays whose chunk codecs it knows; `native = False` forces the
        # Python path (tests compare the two)
        inner = sh["chain"] if sh else self._chain
        self._native_codecs = _native_tail(inner) if self.ndim == 3 else None
        self.native = True
        # read statistics (bench.py, config C4): decoded bytes handed out, stored bytes read, seconds inside reads
        self.read_bytes = self.stored_bytes = 0
        self.read_seconds = 0.0

    def _fill(self, v):
        if isinstance(v, str):
            v = {"NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(v, v)
        if isinstance(v, str) and v.startswith("0x"):
            return np.frombuffer(int(v, 16).to_bytes(self.dtype.itemsize, "little"), self.dtype)[0]
        return np.array(v).astype(self.dtype)[()]

    # ---- storage keys ----------------------------------------------------------------------------
    def _key(self, idx) -> str:
        parts = [str(i) for i in idx]
        if self._v2:
            return self._sep.join(parts) if parts else "0"
        return "c" + "".join(self._sep + p for p in parts)

    def _file(self, idx):
        return os.path.join(self.path, *self._key(idx).split("/"))

    # ---- chunks ------------------------------------------------------------------------------------
    def _remember(self, key, block):
        self._cache[key] = block
        while len(self._cache) > self._cache_chunks:
            self._cache.popitem(last=False)
        return block

    def _shard_index(self, path, f, size):
        hit = self._index_cache.get(path)
        if hit is not None:
            return hit
        sh = self._chain.shard
        per = tuple(o // i for o, i in zip(self._outer, sh["inner"]))
        n = int(np.prod(per))
        raw_len = 16 * n + (4 if any(c["name"] == "crc32c" for c in sh["index_codecs"]) else 0)
        f.seek(size - raw_len if sh["at_end"] else 0)
        raw = f.read(raw_len)
        endian = "<"
        for c in sh["index_codecs"]:
            if c["name"] == "bytes" and c.get("configuration", {}).get("endian", "little") == "big":
                endian = ">"
        body = _decode_bytes_codecs([c for c in sh["index_codecs"] if c["name"] != "bytes"], raw, 16 * n)
        index = np.frombuffer(body, endian + "u8", count=2 * n).reshape(per + (2,))
        self._index_cache[path] = index
        while len(self._index_cache) > 64:
            self._index_cache.popitem(last=False)
        return index

    def _inner_chunk(self, inner_idx) -> np.ndarray | None:
        """Decoded chunk at inner-chunk grid position ``inner_idx`` (``None``: not stored -> fill value)."""
        hit = self._cache.get(inner_idx)
        if hit is not None:
            self._cache.move_to_end(inner_idx)
            return hit
        sh = self._chain.shard
        if sh is None:
            path = self._file(inner_idx)
            if not os.path.exists(path):
                return None
            with open(path, "rb") as f:
                return self._remember(inner_idx, self._chain.decode_block(f.read()))
        per = tuple(o // i for o, i in zip(self._outer, sh["inner"]))
        outer = tuple(i // p for i, p in zip(inner_idx, per))
        within = tuple(i % p for i, p in zip(inner_idx, per))
        path = self._file(outer)
        if not os.path.exists(path):
            return None
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            index = self._shard_index(path, f, size)
            off, nbytes = (int(v) for v in index[within])
            if off == _EMPTY and nbytes == _EMPTY:
                return None
            f.seek(off)
            data = f.read(nbytes)
        return self._remember(inner_idx, sh["chain"].decode_block(data))

    # ---- numpy-style reads -----------------------------------------------------------------------------
    def __getitem__(self, key) -> np.ndarray:
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        if len(key) != self.ndim:
            raise IndexError("too many indices")
        lo, hi, squeeze = [], [], []
        for k, n in zip(key, self.shape):
            if isinstance(k, slice):
                a, b, step = k.indices(n)
                if step != 1:
                    raise IndexError("only unit-stride slices are supported")
                lo.append(a); hi.append(max(a, b)); squeeze.append(False)
            else:
                k = int(k) + (n if int(k) < 0 else 0)
                if not 0 <= k < n:
                    raise IndexError("index out of range")
                lo.append(k); hi.append(k + 1); squeeze.append(True)
        t_read = time.perf_counter()
        if self.native and self._native_codecs is not None and _host_codecs() is not None and all(h > l for l, h in zip(lo, hi)):
            out = self._read_native(lo, hi)
            self.read_seconds += time.perf_counter() - t_read
            self.read_bytes += out.nbytes
            return out[tuple(0 if s else slice(None) for s in squeeze)]
        out = np.full([h - l for l, h in zip(lo, hi)], self.fill_value, self.dtype)
        if out.size:
            c = self.chunks
            ranges = [range(l // s, (h - 1) // s + 1) for l, h, s in zip(lo, hi, c)]
            for idx in product(*ranges):
                block = self._inner_chunk(idx)
                if block is None:
                    continue
                src, dst = [], []
                for i, s, l, h in zip(idx, c, lo, hi):
                    b0, b1 = max(l, i * s), min(h, (i + 1) * s)
                    src.append(slice(b0 - i * s, b1 - i * s))
                    dst.append(slice(b0 - l, b1 - l))
                out[tuple(dst)] = block[tuple(src)]
        self.read_seconds += time.perf_counter() - t_read
        self.read_bytes += out.nbytes
        return out[tuple(0 if s else slice(None) for s in squeeze)]

    def _read_native(self, lo, hi) -> np.ndarray:
        """The box [lo, hi) through ``host_codecs.c``.  Stores with the usual shard index (little-endian u64 pairs
        [+ crc32c]) and plain chunk files take ``svr_zarr_read_groups``: ONE call for the whole request — the pool's
        threads open the files, read and verify the shard indexes, read the byte ranges and check (crc32c), decompress
        (zstd) and place every inner chunk.  Other index codecs: the files are read here and only the chunks go to
        ``svr_zarr_decode_chunks``."""
        lib = _host_codecs()
        zstd, crc = self._native_codecs
        c = self.chunks
        sh = self._chain.shard
        ranges = [range(l // s, (h - 1) // s + 1) for l, h, s in zip(lo, hi, c)]
        n = len(ranges[0]) * len(ranges[1]) * len(ranges[2])
        idx = np.stack(np.meshgrid(*[np.asarray(r, np.int64) for r in ranges], indexing="ij"), -1).reshape(n, 3)
        origin = np.ascontiguousarray(idx * np.asarray(c, np.int64) - np.asarray(lo, np.int64), np.int32)
        off = np.full(n, _EMPTY, np.uint64)
        nbytes = np.zeros(n, np.uint64)
        per = tuple(o // i for o, i in zip(self._outer, c)) if sh else (1, 1, 1)
        outer = idx // np.asarray(per, np.int64)
        within = idx % np.asarray(per, np.int64)
        # group the chunks by the file that stores them (a shard, or the chunk's own file)
        span = outer.max(axis=0) + 1
        key = (outer[:, 0] * span[1] + outer[:, 1]) * span[2] + outer[:, 2]
        order = np.argsort(key, kind="stable")
        _, starts = np.unique(key[order], return_index=True)
        bounds = list(starts) + [n]
        # the whole request in ONE native call when the shard index is what the reference's stores have (little-endian
        # u64 pairs [+ crc32c]): files are opened, indexed and read by the pool's threads too
        index_names = [c["name"] for c in sh["index_codecs"]] if sh else ["bytes"]
        index_little = all(c.get("configuration", {}).get("endian", "little") == "little" for c in (sh["index_codecs"] if sh else []) if c["name"] == "bytes")
        if index_names in (["bytes"], ["bytes", "crc32c"]) and index_little and n < 2 ** 31:
            return self._read_native_groups(lib, lo, hi, idx, origin, outer, within, per, order, bounds, zstd, crc,
                                            "crc32c" in index_names)
        pieces, total = [], 0
        for g in range(len(starts)):
            members = order[bounds[g]:bounds[g + 1]]
            o = tuple(int(v) for v in outer[members[0]])
            path = self._file(o)
            try:
                f = open(path, "rb")
            except FileNotFoundError:
                continue                                        # not stored: fill value
            with f:
                if sh is None:
                    data = np.frombuffer(f.read(), np.uint8)
                    if data.size:
                        off[members[0]], nbytes[members[0]] = data.ctypes.data, data.size
                    pieces.append(data)
                    total += data.size
                    continue
                index = self._shard_index(path, f, os.fstat(f.fileno()).st_size)
                ent = index[tuple(within[members].T)]           # [m, 2] (offset, nbytes)
                stored = ent[:, 0] != _EMPTY
                if not stored.any():
                    continue
                e_off, e_len = ent[stored, 0].astype(np.int64), ent[stored, 1].astype(np.int64)
                # one read of the span that covers the wanted chunks (they lie close together in a shard)
                a, b = int(e_off.min()), int((e_off + e_len).max())
                f.seek(a)
                data = np.frombuffer(f.read(b - a), np.uint8)
                if data.size != b - a:
                    raise ValueError(f"shard {path} is shorter than its index says")
                off[members[stored]] = (e_off - a + data.ctypes.data).astype(np.uint64)      # addresses (base = NULL below)
                nbytes[members[stored]] = e_len.astype(np.uint64)
                pieces.append(data)                             # keeps the bytes alive until the decode call returns
                total += data.size
        self.stored_bytes += total
        out = np.empty([h - l for l, h in zip(lo, hi)], self.dtype)
        fill = np.asarray(self.fill_value, self.dtype).reshape(1)
        cs = (ctypes.c_int32 * 3)(*c)
        strides = (ctypes.c_int64 * 3)(*out.strides)
        shape = (ctypes.c_int32 * 3)(*out.shape)
        rc = lib.svr_zarr_decode_chunks(
            n, None, off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nbytes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
            int(zstd), int(crc), self.dtype.itemsize, cs, out.ctypes.data, strides, shape,
            origin.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), fill.ctypes.data, _threads())
        if rc < 0:
            raise RuntimeError("the zstd codec needs libzstd, which was not found on this machine")
        if rc > 0:
            bad = tuple(int(v) for v in idx[rc - 1])
            raise ValueError(f"chunk {bad} of {self.path} is corrupt (crc32c mismatch, corrupt zstd frame or wrong size)")
        return out

    def _read_native_groups(self, lib, lo, hi, idx, origin, outer, within, per, order, bounds, zstd, crc, index_crc):
        sh = self._chain.shard
        c = self.chunks
        n = len(order)
        paths, first = [], [0]
        for g in range(len(bounds) - 1):
            a, b = int(bounds[g]), int(bounds[g + 1])
            path = self._file(tuple(int(v) for v in outer[order[a]])).encode()
            for q in range(a, b, 24):                           # <= 24 chunks per group: the groups spread over the pool
                paths.append(path)
                first.append(min(b, q + 24))
        ng = len(paths)
        c_paths = (ctypes.c_char_p * ng)(*paths)
        grp_first = np.asarray(first, np.int32)
        w = within[order]
        within_lin = np.ascontiguousarray((w[:, 0] * per[1] + w[:, 1]) * per[2] + w[:, 2], np.int32)
        origin_sorted = np.ascontiguousarray(origin[order])
        out = np.empty([h - l for l, h in zip(lo, hi)], self.dtype)
        fill = np.asarray(self.fill_value, self.dtype).reshape(1)
        stored, bad_group = ctypes.c_uint64(0), ctypes.c_int(-1)
        i32p = ctypes.POINTER(ctypes.c_int32)
        rc = lib.svr_zarr_read_groups(
            ng, c_paths, grp_first.ctypes.data_as(i32p), within_lin.ctypes.data_as(i32p),
            int(np.prod(per)) if sh else 0, int(sh["at_end"]) if sh else 0, int(index_crc) if sh else 0,
            int(zstd), int(crc), self.dtype.itemsize, (ctypes.c_int32 * 3)(*c), out.ctypes.data,
            (ctypes.c_int64 * 3)(*out.strides), (ctypes.c_int32 * 3)(*out.shape), origin_sorted.ctypes.data_as(i32p),
            fill.ctypes.data, _threads(), ctypes.byref(stored), ctypes.byref(bad_group))
        self.stored_bytes += int(stored.value)
        if rc == -1:
            raise RuntimeError("the zstd codec needs libzstd, which was not found on this machine")
        if rc == -2:
            raise ValueError(f"{paths[bad_group.value].decode()}: unreadable, shorter than its shard index says, or its index fails "
                             "its crc32c")
        if rc > 0:
            bad = tuple(int(v) for v in idx[order[rc - 1]])
            raise ValueError(f"chunk {bad} of {self.path} is corrupt (crc32c mismatch, corrupt zstd frame or wrong size)")
        return out

    def __array__(self, dtype=None, copy=None):
        a = self[(slice(None),) * self.ndim]
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f"<ZarrV3Array {self.path} shape={self.shape} dtype={self.dtype} chunks={self.chunks} shards={self.shards}>"


class ZarrV3Group:
    """``zarr.open_group(path)[name]`` for the stores the reference's scripts open (scripts/mouse.py:32-47)."""

    def __init__(self, path: str):
        with open(os.path.join(path, "zarr.json")) as f:
            meta = json.load(f)
        if meta.get("zarr_format") != 3 or meta.get("node_type") != "group":
            raise ValueError(f"{path} is not a zarr v3 group")
        self.path, self.attrs = path, meta.get("attributes", {})

    def __getitem__(self, name: str):
        return open_zarr(os.path.join(self.path, *name.split("/")))

    def keys(self):
        return sorted(n for n in os.listdir(self.path) if os.path.exists(os.path.join(self.path, n, "zarr.json")))

    def __contains__(self, name):
        return os.path.exists(os.path.join(self.path, *name.split("/"), "zarr.json"))


def open_zarr(path: str):
    """An array or a group, whatever ``path/zarr.json`` says."""
    with open(os.path.join(path, "zarr.json")) as f:
        node = json.load(f).get("node_type")
    return ZarrV3Group(path) if node == "group" else ZarrV3Array(path)


def open_group(path: str) -> ZarrV3Group:
    return ZarrV3Group(path)


# ---------------------------------------------------------------------------------------------------
# writer (tests and synthetic stores; the layout of the zarr v3 specification)
# ---------------------------------------------------------------------------------------------------
def create_group(path: str, attributes=None) -> ZarrV3Group:
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "zarr.json"), "w") as f:
        json.dump({"zarr_format": 3, "node_type": "group", "attributes": attributes or {}}, f)
    return ZarrV3Group(path)


def create_array(path: str, shape, dtype, chunks, shards=None, compressor: str | None = "zstd", fill_value=0,
                 index_location: str = "end", separator: str = "/") -> "ZarrV3Array":
    """Write the metadata of an EMPTY zarr v3 array (every chunk missing = fill value) and open it: the layout of
    ``group.create_array(name, shape, chunks=(16,16,16), shards=(64,64,64))`` in the reference's builders
    (scripts/create_mouse_multiscale.py:102-131).  Fill it with :func:`write_block`."""
    dtype = np.dtype(dtype)
    names = {v: k for k, v in _DTYPES.items()}
    data_type = names[dtype.str.lstrip("<>|=")] if dtype.kind != "b" else "bool"
    chunks = tuple(int(c) for c in chunks)
    inner_codecs = [{"name": "bytes", "configuration": {"endian": "little"}}]
    if compressor == "gzip":
        inner_codecs.append({"name": "gzip", "configuration": {"level": 5}})
    elif compressor == "zstd":
        inner_codecs.append({"name": "zstd", "configuration": {"level": 3, "checksum": False}})
    elif compressor is not None:
        raise ValueError("compressor must be None, 'gzip' or 'zstd'")
    if shards is not None:
        shards = tuple(int(s) for s in shards)
        if any(s % c for s, c in zip(shards, chunks)):
            raise ValueError("shards must be a multiple of chunks")
        index_codecs = [{"name": "bytes", "configuration": {"endian": "little"}}, {"name": "crc32c"}]
        codecs = [{"name": "sharding_indexed", "configuration": {"chunk_shape": list(chunks), "codecs": inner_codecs,
                                                                  "index_codecs": index_codecs, "index_location": index_location}}]
        outer = shards
    else:
        codecs, outer = inner_codecs, chunks
    meta = {"zarr_format": 3, "node_type": "array", "shape": [int(v) for v in shape], "data_type": data_type,
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": list(outer)}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": separator}},
            "fill_value": fill_value if not isinstance(fill_value, float) or np.isfinite(fill_value) else "NaN",
            "codecs": codecs, "attributes": {}}
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "zarr.json"), "w") as f:
        json.dump(meta, f)
    return ZarrV3Array(path)


def write_block(arr: "ZarrV3Array", outer_index, block, skip_fill_chunks: bool = True) -> int:
    """Store ``block`` — the data of the outer chunk (shard, or plain chunk) at grid position ``outer_index``, clipped to
    the array (``block.shape <= outer chunk shape``; what lies beyond it is padded with the fill value) — as that
    chunk's file.  Inner chunks are encoded by ``host_codecs.c`` (all of a shard in one call) where it knows the codec
    chain, else in Python.  Returns the bytes written (0: nothing but fill values, no file)."""
    outer = arr._outer
    sh = arr._chain.shard
    chunks = arr.chunks
    inner_chain = sh["chain"] if sh else arr._chain
    dtype = arr.dtype
    block = np.ascontiguousarray(block, dtype)
    if block.ndim != arr.ndim or any(b > o for b, o in zip(block.shape, outer)):
        raise ValueError("the block does not fit the outer chunk")
    fill = np.asarray(arr.fill_value, dtype).reshape(1)
    file = arr._file(tuple(int(i) for i in outer_index))
    per = tuple(o // c for o, c in zip(outer, chunks))
    grid = list(product(*[range(p) for p in per]))
    native = _native_tail(inner_chain) if arr.ndim == 3 else None
    lib = _host_codecs()
    encoded = {}
    if native is not None and lib is not None and arr.native:
        zstd, crc = native
        level = 3
        for c in inner_chain.tail:
            if c["name"] == "zstd":
                level = int(c.get("configuration", {}).get("level", 3))
        keep = [i for i in grid if all(ii * c < b for ii, c, b in zip(i, chunks, block.shape))]
        n = len(keep)
        if n:
            raw = int(np.prod(chunks)) * dtype.itemsize
            slot = int(lib.svr_zarr_encode_bound(raw, int(zstd)))
            if slot == 0:
                raise RuntimeError("the zstd codec needs libzstd, which was not found on this machine")
            out = np.empty(n * slot, np.uint8)
            sizes = np.zeros(n, np.uint64)
            corner = np.ascontiguousarray(np.asarray(keep, np.int32) * np.asarray(chunks, np.int32))
            rc = lib.svr_zarr_encode_chunks(
                n, block.ctypes.data, (ctypes.c_int32 * 3)(*block.shape), corner.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                dtype.itemsize, (ctypes.c_int32 * 3)(*chunks), fill.ctypes.data, int(zstd), level, int(crc), int(skip_fill_chunks),
                out.ctypes.data, slot, sizes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _threads(reading=False))
            if rc != 0:
                raise RuntimeError(f"svr_zarr_encode_chunks failed ({rc})")
            for k, i in enumerate(keep):
                if sizes[k]:
                    encoded[i] = out[k * slot:k * slot + int(sizes[k])].tobytes()
    else:
        for i in grid:
            lo = [ii * c for ii, c in zip(i, chunks)]
            if any(l >= b for l, b in zip(lo, block.shape)):
                continue
            piece = np.full(chunks, fill[0], dtype)
            hi = [min(l + c, b) for l, c, b in zip(lo, chunks, block.shape)]
            piece[tuple(slice(0, h - l) for l, h in zip(lo, hi))] = block[tuple(slice(l, h) for l, h in zip(lo, hi))]
            if skip_fill_chunks and np.all(piece == fill[0]):
                continue
            encoded[i] = inner_chain.encode_block(piece)
    if not encoded:
        return 0
    os.makedirs(os.path.dirname(file), exist_ok=True)
    if sh is None:
        data = encoded[grid[0]]
    else:
        index = np.full(per + (2,), _EMPTY, np.uint64)
        index_len = 16 * int(np.prod(per)) + 4
        base = 0 if sh["at_end"] else index_len
        body = bytearray()
        for i in grid:
            if i in encoded:
                index[i] = (base + len(body), len(encoded[i]))
                body += encoded[i]
        raw_index = index.astype("<u8").tobytes()
        raw_index += struct.pack("<I", crc32c(raw_index))
        data = bytes(body) + raw_index if sh["at_end"] else raw_index + bytes(body)
    with open(file, "wb") as f:
        f.write(data)
    arr._index_cache.pop(file, None)
    arr._cache.clear()
    return len(data)


def write_array(path: str, array, chunks, shards=None, compressor: str | None = "zstd", fill_value=0,
                index_location: str = "end", separator: str = "/", skip_fill_chunks: bool = True) -> ZarrV3Array:
    """Store ``array`` as a zarr v3 array: ``chunks`` (inner chunk shape), optional ``shards`` (outer chunk shape,
    a multiple of ``chunks``), ``compressor`` in {None, "gzip", "zstd"} — with shards this is the layout of
    ``group.create_array(name, shape, chunks=(16,16,16), shards=(64,64,64))`` in the reference's builders."""
    a = np.asarray(array)
    arr = create_array(path, a.shape, a.dtype, chunks, shards, compressor, fill_value, index_location, separator)
    outer = arr._outer
    for oidx in product(*[range(-(-n // o)) for n, o in zip(a.shape, outer)]):
        lo = [i * o for i, o in zip(oidx, outer)]
        write_block(arr, oidx, a[tuple(slice(l, l + o) for l, o in zip(lo, outer))], skip_fill_chunks)
    return arr
