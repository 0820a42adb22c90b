"""The drop-in surface (no GPU): constructor signatures, defaults, validation errors of the
reference's classes (src/sub_volume/_material.py, _wobject.py, _wrapping_buffer.py), and the C ABI:
libsvr_hip.so loads and exports every symbol include/svr.h declares."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

import sub_volume_renderer_amd as svr
from sub_volume_renderer_amd import SubVolume, SubVolumeMaterial, WrappingBuffer, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI -------------------------------------------------------------------
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "svr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_bound_and_exported():
    syms = declared_symbols()
    assert len(syms) >= 18
    assert set(syms) == set(_native.SIGNATURES)            # the binding covers the header exactly
    lib = ctypes.CDLL(_native.LIB_PATH)                    # loads without a GPU
    for s in syms:
        assert hasattr(lib, s), f"{s} not exported by libsvr_hip.so"


def test_abi_version_and_struct_sizes():
    lib = _native.lib()
    header = open(os.path.join(ROOT, "include", "svr.h")).read()
    assert lib.svr_abi_version() == int(re.search(r"#define SVR_ABI_VERSION (\d+)", header).group(1))
    assert ctypes.sizeof(_native.LodState) == 36            # 3xi4 + 3xi4 + 3xf4 (_wrapping_buffer.py:15-19)
    assert ctypes.sizeof(_native.Camera) == 6 * 64 + 12
    assert ctypes.sizeof(_native.Frame) == 32


def test_header_is_plain_c_and_the_c_example_links_against_the_library_alone(tmp_path):
    """include/svr.h compiles as strict C99 / C11 and as C++17 without a warning, and examples/c_abi_demo.c links
    against libsvr_hip.so + the HIP runtime and nothing of Python or torch (it RUNS in tests/test_gpu_c_consumer.py)."""
    import shutil
    import subprocess

    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    if gcc is None or gxx is None:
        pytest.skip("no C / C++ compiler on this machine")
    unit = tmp_path / "unit.c"
    unit.write_text('#include "svr.h"\nint main(void) { return svr_abi_version() == SVR_ABI_VERSION ? 0 : 1; }\n')
    strict = ["-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only"]
    subprocess.run([gcc, "-std=c99", *strict, str(unit)], check=True)
    subprocess.run([gcc, "-std=c11", *strict, str(unit)], check=True)
    subprocess.run([gxx, "-std=c++17", *strict, "-x", "c++", str(unit)], check=True)
    if not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no ROCm headers: the example needs hip_runtime_api.h")
    _native.lib()                                           # built (and loadable) before linking against it
    csrc = os.path.join(ROOT, "sub_volume_renderer_amd", "csrc")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run([gcc, "-O1", "-std=c11", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", csrc, "-lsvr_hip",
                    "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    linked = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libsvr_hip.so" in linked and "torch" not in linked and "python" not in linked


def test_create_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    data = np.zeros((8, 8, 8), np.uint8)
    buf = WrappingBuffer(data, data, (2, 2, 2), (2, 2, 2))
    with pytest.raises(Exception):                          # no silent CPU fallback
        buf.load_logical_roi(svr.Roi((0, 0, 0), (2, 2, 2)))


def test_exports():
    for name in ("SubVolume", "SubVolumeMaterial", "WrappingBuffer"):      # sub_volume/__init__.py:17-21
        assert name in svr.__all__


# ---- SubVolumeMaterial (_material.py:26-159) ----------------------------------
def test_material_signature_and_defaults():
    sig = inspect.signature(SubVolumeMaterial.__init__)
    names = list(sig.parameters)[1:]
    assert names == ["lmip_threshold", "lmip_fall_off", "lmip_max_samples", "fog_density", "fog_color",
                     "colors", "clim", "gamma", "opacity"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["lmip_fall_off"] == 0.5 and d["lmip_max_samples"] == 10 and d["fog_density"] == 0.5
    assert d["fog_color"] == (0.5, 0.5, 0.5) and d["colors"] is None and d["clim"] == (0, 1)
    assert d["gamma"] == 1.0 and d["opacity"] == 1.0
    m = SubVolumeMaterial(0.5)
    assert float(m.lmip_threshold) == 0.5 and int(m.lmip_max_samples) == 10
    assert m.fog_color == (0.5, 0.5, 0.5)
    assert m.colors == [(0.0, 1.0, 1.0, 1.0), (0.25, 1.0, 1.0, 1.0), (0.5, 1.0, 1.0, 1.0), (0.75, 1.0, 1.0, 1.0)]
    assert int(m._color_count) == 4 and m.clim == (0.0, 1.0) and m.depth_test is True


def test_material_validation():
    m = SubVolumeMaterial(0.5)
    with pytest.raises(ValueError):
        m.fog_color = (0.1, 0.2)
    with pytest.raises(ValueError):
        m.fog_color = (0.1, "a", 0.2)
    with pytest.raises(ValueError):
        m.fog_color = (0.1, 1.5, 0.2)
    with pytest.raises(TypeError):
        m.colors = "red"
    with pytest.raises(TypeError):
        m.colors = [(0.1, 0.2)]
    v = m._version
    m.lmip_max_samples = 7.9
    assert int(m.lmip_max_samples) == 7 and m._version > v   # i32 field (_material.py:10-11)
    m.colors = [(0.1, 0.2, 0.3)]
    assert int(m._color_count) == 1 and m.colors[0][3] == 1.0


def test_render_modes_are_a_material_switch():
    """FUTURE.md:111-120: "different materials that could indicate which rendering mode we are in"."""
    m = SubVolumeMaterial(0.5)
    assert m.render_mode == "lmip" and m.weight_falloff == 0.5
    for mode in ("mip", "weighted_average", "LMIP"):
        m.render_mode = mode
        assert m.render_mode == mode.lower()
    with pytest.raises(ValueError):
        m.render_mode = "fading"
    assert m.lmip_uniforms() == (0.5, 0.5, 10)              # the lmip_* properties survive mode changes
    v = m._version
    m.weight_falloff = 2
    assert m.weight_falloff == 2.0 and m._version > v
    for bad in (-0.1, float("inf"), float("nan")):
        with pytest.raises(ValueError):
            m.weight_falloff = bad
    # the C struct carries both (ABI 6)
    assert {"render_mode", "weight_falloff"} <= {name for name, _ in _native.Material._fields_}


# ---- SubVolume (_wobject.py:20-208) ---------------------------------------------
def pairs(n=3, base=32):
    out = []
    for k in range(n):
        s = base >> k
        out.append((np.zeros((s, s, 2 * s), np.uint8), np.zeros((s, s, 2 * s), np.uint32)))
    return out


def test_subvolume_signature():
    names = list(inspect.signature(SubVolume.__init__).parameters)[1:5]
    assert names == ["material", "data_segmentation_pairs", "buffer_shape_in_chunks", "chunk_shape_in_pixels"]
    params = inspect.signature(SubVolume.center_on_position).parameters
    assert list(params)[1:3] == ["position", "sizes"]                      # _wobject.py:135-139
    assert params["asynchronous"].kind is inspect.Parameter.KEYWORD_ONLY   # extension, off by default
    assert params["asynchronous"].default is False
    names = list(inspect.signature(WrappingBuffer.__init__).parameters)[1:6]
    assert names == ["backing_data", "segmentations", "shape_in_chunks", "chunk_shape_in_pixels", "scale_factor"]


def test_subvolume_construction_and_validation():
    m = SubVolumeMaterial(0.5)
    v = SubVolume(m, pairs(), [(2, 2, 2), (3, 3, 3), (4, 4, 4)], [(8, 8, 8), (4, 4, 4), (2, 2, 2)])
    assert len(v.wrapping_buffers) == 3 and len(v.textures) == 3 and len(v.segmentations_textures) == 3
    assert v.volume_dimensions == (32.0, 32.0, 64.0)                       # numpy order (_wobject.py:103-107)
    assert tuple(v._volume_dimensions) == (64.0, 32.0, 32.0)               # uniform is reversed (:121-123)
    assert v.wrapping_buffers[1].scale_factor == (0.5, 0.5, 0.5)
    assert tuple(v.wrapping_buffers[1].shape_in_pixels) == (12, 12, 12)
    assert v.wrapping_buffers[0]._current_logical_roi_in_pixels is None
    u = v.wrapping_buffers[2].uniform_buffer.data
    assert tuple(u["current_logical_shape_in_pixels"]) == (0, 0, 0)        # ROI None <-> zeros (:90-96)
    # a tuple broadcasts to every scale (:34-36, :54-56)
    v2 = SubVolume(m, pairs(), (2, 2, 2), (4, 4, 4))
    assert [tuple(b.shape_in_pixels) for b in v2.wrapping_buffers] == [(8, 8, 8)] * 3
    with pytest.raises(ValueError):
        SubVolume(m, pairs(), [(2, 2, 2)], [(8, 8, 8)] * 3)                # :40-43
    with pytest.raises(ValueError):
        SubVolume(m, pairs(), [(2, 2, 2)] * 3, [(8, 8, 8)] * 2)            # :60-63
    with pytest.raises(ValueError):
        SubVolume(m, pairs(), [(2, 2, 2)] * 3, None)                       # no .chunks (:46-53)
    with pytest.raises(ValueError):
        SubVolume(m, pairs(), [(2, 2, 2)] * 3, [(8, 8)] * 3)               # rank mismatch (:66-70)

    class Chunked(np.ndarray):
        chunks = (4, 4, 4)

    p = [(d.view(Chunked), s) for d, s in pairs()]
    v3 = SubVolume(m, p, (2, 2, 2))
    assert tuple(v3.wrapping_buffers[2].chunk_shape_in_pixels) == (4, 4, 4)
    with pytest.raises(ValueError):
        v.center_on_position((0, 0, 0), sizes=[(1, 1, 1)])                 # :178-181


def test_world_transform_and_camera_matrices():
    cam = svr.PerspectiveCamera(45, 16 / 9, depth_range=(1.0, 100.0))
    cam.world.position = (3.0, 4.0, 5.0)
    cam.look_at((0.0, 0.0, 0.0))
    v = cam.view_matrix @ np.array([0.0, 0.0, 0.0, 1.0])
    assert abs(v[0]) < 1e-9 and abs(v[1]) < 1e-9 and v[2] < 0             # target on the -z axis
    assert np.allclose(cam.view_matrix @ cam.camera_matrix, np.eye(4))
    p = cam.projection_matrix
    near = p @ np.array([0, 0, -1.0, 1]); far = p @ np.array([0, 0, -100.0, 1])
    assert abs(near[2] / near[3]) < 1e-9 and abs(far[2] / far[3] - 1) < 1e-9   # depth range [0, 1]
    t = svr.AffineTransform()
    t.position = (1, 2, 3); t.scale_z = 6
    assert np.allclose(t.inverse_matrix @ t.matrix, np.eye(4)) and t.matrix[2, 2] == 6


def test_segmentations_are_optional_but_all_or_none():
    """FUTURE.md:178-193: label-less volumes; mixing labelled and unlabelled scales is refused."""
    d0, d1 = np.zeros((16, 16, 16), np.uint8), np.zeros((8, 8, 8), np.uint8)
    vol = SubVolume(SubVolumeMaterial(0.5), [(d0, None), (d1, None)], (2, 2, 2), (4, 4, 4))
    assert vol._rings.labels is False and vol.wrapping_buffers[0].segmentations is None
    assert SubVolume(SubVolumeMaterial(0.5), [(d0, d0), (d1, d1)], (2, 2, 2), (4, 4, 4))._rings.labels is True
    with pytest.raises(ValueError):
        SubVolume(SubVolumeMaterial(0.5), [(d0, d0), (d1, None)], (2, 2, 2), (4, 4, 4))


def test_host_codec_header_symbols_are_exported_and_the_header_is_plain_c(tmp_path):
    """include/svr_host_codecs.h (the native half of the chunk-store reader): every function it declares is exported by
    libsvr_hostcodec.so, which loads without a GPU and without torch; the header compiles as strict C and as C++."""
    import shutil
    import subprocess

    import __graft_entry__ as g

    header = open(os.path.join(ROOT, "include", "svr_host_codecs.h")).read()
    names = re.findall(r"^\s*(?:uint32_t|int|size_t)\s+(svr_\w+)\s*\(", header, flags=re.M)
    assert sorted(names) == ["svr_crc32c", "svr_zarr_decode_chunks", "svr_zarr_encode_bound", "svr_zarr_encode_chunks",
                             "svr_zarr_read_groups"]
    lib = ctypes.CDLL(g.build_host_codecs())
    for s in names:
        assert hasattr(lib, s), f"{s} not exported by libsvr_hostcodec.so"
    lib.svr_crc32c.restype = ctypes.c_uint32
    lib.svr_crc32c.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32]
    assert lib.svr_crc32c(b"123456789", 9, 0) == 0xE3069283
    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    if gcc is None or gxx is None:
        pytest.skip("no C / C++ compiler on this machine")
    unit = tmp_path / "unit.c"
    unit.write_text('#include "svr_host_codecs.h"\nint main(void) { return svr_crc32c("", 0, 0) != 0; }\n')
    strict = ["-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only"]
    subprocess.run([gcc, "-std=c99", *strict, str(unit)], check=True)
    subprocess.run([gxx, "-std=c++17", *strict, "-x", "c++", str(unit)], check=True)


def test_blocked_twin_argument():
    from sub_volume_renderer_amd._wrapping_buffer import blocked_twin_lods

    shapes = [(64, 64, 64), (32, 32, 32), (30, 32, 32), (32, 32, 36)]                # numpy order: the last two do not fit
    # per LOD: 0 no copy, 1 a copy taken instead of staging bricks, 2 a copy for waves that stage none
    assert blocked_twin_lods(shapes, "auto") == [1, 0, 0, 0]
    assert blocked_twin_lods(shapes, "all") == [1, 2, 0, 0]
    assert blocked_twin_lods(shapes[2:], "auto") == [0, 0] and blocked_twin_lods(shapes[2:], "all") == [0, 0]
    assert blocked_twin_lods(shapes, True) == [1, 1, 0, 0]
    assert blocked_twin_lods(shapes, False) == [0] * 4
    assert blocked_twin_lods(shapes, [False, True, False, False]) == [0, 1, 0, 0]
    assert blocked_twin_lods(shapes, [2, "fallback", 0, False]) == [2, 2, 0, 0]
    with pytest.raises(ValueError):
        blocked_twin_lods(shapes, [True, True, True, False])
    with pytest.raises(ValueError):
        blocked_twin_lods(shapes, [True])
    with pytest.raises(ValueError):
        blocked_twin_lods(shapes, "always")


def test_auto_twin_is_given_up_when_the_device_has_no_room(monkeypatch):
    """`blocked_twin="auto"`: SVR_ERR_NOMEM from svr_create with the copy is answered by creating the rings without it;
    an explicit request is not second-guessed (MemoryError)."""
    from sub_volume_renderer_amd._wrapping_buffer import DeviceRings

    calls = []

    class FakeLib:
        def svr_create(self, device, n, descs, out):
            twins = [int(descs[i].blocked_twin) for i in range(n)]
            calls.append(twins)
            if any(twins):
                return -3
            out._obj.value = 0x1234
            return 0

        def svr_last_error(self):
            return b"svr_create: out of device memory for ring textures"

        def svr_destroy(self, h):
            return 0

    monkeypatch.setattr(_native, "lib", lambda: FakeLib())
    rings = DeviceRings([(64, 64, 64), (32, 32, 32)], device=0, density_storage="uint8", blocked_twin="all")
    assert rings.blocked_twin == [1, 2]
    assert rings.handle.value == 0x1234
    assert calls == [[1, 2], [0, 0]] and rings.blocked_twin == [0, 0]
    calls.clear()
    wanted = DeviceRings([(64, 64, 64)], device=0, density_storage="uint8", blocked_twin=True)
    with pytest.raises(MemoryError):
        wanted.handle
    assert calls == [[1]]
