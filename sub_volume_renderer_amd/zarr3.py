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
        return out[tuple(0 if s else slice(None) for s in squeeze)]

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


def write_array(path: str, array, chunks, shards=None, compressor: str | None = "zstd", fill_value=0,
                index_location: str = "end", separator: str = "/", skip_fill_chunks: bool = True) -> ZarrV3Array:
    """Store ``array`` as a zarr v3 array: ``chunks`` (inner chunk shape), optional ``shards`` (outer chunk shape,
    a multiple of ``chunks``), ``compressor`` in {None, "gzip", "zstd"} — with shards this is the layout of
    ``group.create_array(name, shape, chunks=(16,16,16), shards=(64,64,64))`` in the reference's builders."""
    a = np.asarray(array)
    dtype = a.dtype
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
    meta = {"zarr_format": 3, "node_type": "array", "shape": list(a.shape), "data_type": data_type,
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": list(outer)}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": separator}},
            "fill_value": fill_value if not isinstance(fill_value, float) or np.isfinite(fill_value) else "NaN",
            "codecs": codecs, "attributes": {}}
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "zarr.json"), "w") as f:
        json.dump(meta, f)
    inner_chain = _Chain(inner_codecs, dtype, chunks)
    fill = np.array(fill_value).astype(dtype)[()]

    def padded(lo, shape):
        block = np.full(shape, fill, dtype)
        hi = [min(l + s, n) for l, s, n in zip(lo, shape, a.shape)]
        if all(h > l for l, h in zip(lo, hi)):
            block[tuple(slice(0, h - l) for l, h in zip(lo, hi))] = a[tuple(slice(l, h) for l, h in zip(lo, hi))]
        return block

    grid = [-(-n // o) for n, o in zip(a.shape, outer)]
    for oidx in product(*[range(g) for g in grid]):
        olo = [i * o for i, o in zip(oidx, outer)]
        key = "c" + "".join(separator + str(i) for i in oidx)
        file = os.path.join(path, *key.split("/"))
        if shards is None:
            block = padded(olo, chunks)
            if skip_fill_chunks and np.all(block == fill):
                continue
            os.makedirs(os.path.dirname(file), exist_ok=True)
            with open(file, "wb") as f:
                f.write(inner_chain.encode_block(block))
            continue
        per = tuple(o // c for o, c in zip(outer, chunks))
        index = np.full(per + (2,), _EMPTY, np.uint64)
        body = bytearray()
        index_len = 16 * int(np.prod(per)) + 4
        base = index_len if index_location == "start" else 0
        for iidx in product(*[range(p) for p in per]):
            lo = [o + i * c for o, i, c in zip(olo, iidx, chunks)]
            if any(l >= n for l, n in zip(lo, a.shape)):
                continue
            block = padded(lo, chunks)
            if skip_fill_chunks and np.all(block == fill):
                continue
            enc = inner_chain.encode_block(block)
            index[iidx] = (base + len(body), len(enc))
            body += enc
        if not len(body):
            continue
        raw_index = index.astype("<u8").tobytes()
        raw_index += struct.pack("<I", crc32c(raw_index))
        os.makedirs(os.path.dirname(file), exist_ok=True)
        with open(file, "wb") as f:
            f.write(raw_index + bytes(body) if index_location == "start" else bytes(body) + raw_index)
    return ZarrV3Array(path)
