"""ctypes binding of the C ABI in ``include/svr.h`` (libsvr_hip.so).

There is NO fallback: if the HIP library has not been built, or a call fails,
this module raises.  The library is built in-tree by ``__graft_entry__.build()``
(``hipcc --offload-arch=gfx950``).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVR_LIB") or os.path.join(_HERE, "csrc", "libsvr_hip.so")      # SVR_LIB: A/B builds (tools/ab_build.py)

SVR_MAX_LODS = 8
SVR_PIX_DISCARD, SVR_PIX_MISS, SVR_PIX_HIT = 0, 1, 2

_DTYPES = {
    np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.uint32): 2, np.dtype(np.uint64): 3,
    np.dtype(np.int8): 4, np.dtype(np.int16): 5, np.dtype(np.int32): 6, np.dtype(np.int64): 7,
    np.dtype(np.float32): 8, np.dtype(np.float64): 9,
}


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.dtype(bool):
        return 0
    try:
        return _DTYPES[dt]
    except KeyError:
        raise TypeError(f"unsupported source dtype {dt} for a ring-buffer upload") from None


class LodDesc(C.Structure):
    _fields_ = [("ring_dims", C.c_int32 * 3), ("density_storage", C.c_int32), ("no_labels", C.c_int32),
                ("blocked_twin", C.c_int32)]


SVR_U8, SVR_U16, SVR_F32 = 0, 1, 8


class LodState(C.Structure):
    _fields_ = [("offset", C.c_int32 * 3), ("shape", C.c_int32 * 3), ("scale", C.c_float * 3)]


class Material(C.Structure):
    _fields_ = [
        ("clim", C.c_float * 2),
        ("gamma", C.c_float),
        ("opacity", C.c_float),
        ("lmip_threshold", C.c_float),
        ("lmip_fall_off", C.c_float),
        ("lmip_max_samples", C.c_int32),
        ("fog_density", C.c_float),
        ("fog_color", C.c_float * 3),
        ("color_count", C.c_uint32),
        ("colors", C.POINTER(C.c_float)),
        ("colorspace_srgb", C.c_int32),
        ("clipping_plane_count", C.c_uint32),
        ("clipping_mode_all", C.c_int32),
        ("clipping_planes", C.POINTER(C.c_float)),
        ("render_mode", C.c_int32),
        ("weight_falloff", C.c_float),
    ]


SVR_MODE_LMIP, SVR_MODE_WEIGHTED_AVERAGE = 0, 1


class Camera(C.Structure):
    _fields_ = [
        ("world", C.c_float * 16),
        ("world_inv", C.c_float * 16),
        ("cam", C.c_float * 16),
        ("cam_inv", C.c_float * 16),
        ("proj", C.c_float * 16),
        ("proj_inv", C.c_float * 16),
        ("volume_dimensions", C.c_float * 3),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("frame_w", C.c_int32), ("frame_h", C.c_int32),
        ("x0", C.c_int32), ("y0", C.c_int32),
        ("out_w", C.c_int32), ("out_h", C.c_int32),
        ("band_h", C.c_int32), ("band_pitch", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("rgba", C.c_void_p),
        ("depth", C.c_void_p),
        ("label", C.c_void_p),
        ("flags", C.c_void_p),
        ("steps", C.c_void_p),
        ("pick", C.c_void_p),
        ("pick_id", C.c_uint32),
    ]


class ComposeParams(C.Structure):
    _fields_ = [("bg_bottom", C.c_float * 4), ("bg_top", C.c_float * 4), ("srgb_encode", C.c_int32)]


_I3 = C.c_int32 * 3
_L3 = C.c_int64 * 3

# name -> (restype, argtypes): every symbol include/svr.h declares
SIGNATURES = {
    "svr_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(LodDesc), C.POINTER(C.c_void_p)]),
    "svr_destroy": (C.c_int, [C.c_void_p]),
    "svr_last_error": (C.c_char_p, []),
    "svr_abi_version": (C.c_int, []),
    "svr_set_lod_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LodState)]),
    "svr_get_lod_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LodState)]),
    "svr_set_material": (C.c_int, [C.c_void_p, C.POINTER(Material)]),
    "svr_upload_region": (C.c_int, [C.c_void_p, C.c_int, _I3, _I3,
                                    C.c_void_p, C.c_int, _L3, C.c_void_p, C.c_int, _L3]),
    "svr_upload_region_device": (C.c_int, [C.c_void_p, C.c_int, _I3, _I3,
                                           C.c_void_p, C.c_int, _L3, C.c_void_p, C.c_int, _L3]),
    "svr_publish_uploads": (C.c_int, [C.c_void_p]),
    "svr_mark_uploads": (C.c_int, [C.c_void_p]),
    "svr_uploads_pending": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "svr_upload_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_int]),
    "svr_upload_ticket": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "svr_ticket_pending": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_int)]),
    "svr_read_region": (C.c_int, [C.c_void_p, C.c_int, _I3, _I3, C.c_void_p, C.c_void_p]),
    "svr_clear_lod": (C.c_int, [C.c_void_p, C.c_int]),
    "svr_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame),
                             C.POINTER(Outputs), C.c_void_p]),
    "svr_set_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "svr_untile_stripes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "svr_untile_grid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "svr_comm_unique_id": (C.c_int, [C.c_char * 128]),
    "svr_comm_init": (C.c_int, [C.c_void_p, C.c_char * 128, C.c_int, C.c_int]),
    "svr_comm_destroy": (C.c_int, [C.c_void_p]),
    "svr_gather_tiles": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "svr_pool2x": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, _I3, C.c_int, C.c_int, C.c_void_p]),
    "svr_compose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                              C.POINTER(ComposeParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "svr_sync": (C.c_int, [C.c_void_p]),
    "svr_sync_uploads": (C.c_int, [C.c_void_p]),
    "svr_debug_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "svr_debug_timers": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    "svr_lod_device_ptrs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "svr_lod_twin_ptr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "svr_time_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame),
                                  C.POINTER(Outputs), C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


class SvrError(RuntimeError):
    pass


def lib():
    """Load libsvr_hip.so (once).  Raises loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SvrError(
            f"HIP extension not built: {LIB_PATH} is missing. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` in the repo root. "
            "There is no CPU fallback."
        )
    # torch ships the HIP runtime (libamdhip64.so.7) this process must share:
    # import it first so the dynamic loader resolves our NEEDED entry to that copy.
    import torch  # noqa: F401

    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is missing: loud
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(status: int, what: str = "") -> None:
    if status == 0:
        return
    msg = lib().svr_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    if status == -1:
        raise ValueError(f"{what}: {msg}")
    if status == -3:
        raise MemoryError(f"{what}: {msg}")
    if status == -4:
        raise IndexError(f"{what}: {msg}")
    raise SvrError(f"{what}: status {status}: {msg}")


def i3(v) -> "C.Array":
    return _I3(int(v[0]), int(v[1]), int(v[2]))


def l3(v) -> "C.Array":
    return _L3(int(v[0]), int(v[1]), int(v[2]))


def mat_to_c(m) -> "C.Array":
    """numpy row-major 4x4 -> column-major float[16] (WGSL mat4x4 layout)."""
    a = np.asarray(m, dtype=np.float32).reshape(4, 4)
    return (C.c_float * 16)(*a.T.reshape(-1).tolist())
