"""GPU: the C ABI from a consumer written in plain C (examples/c_abi_demo.c: gcc, libsvr_hip.so + the HIP runtime, no
Python, no torch) — it must draw the very frame the Python host side draws for the same volume, window and camera."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from sub_volume_renderer_amd import PerspectiveCamera, SubVolume, SubVolumeMaterial

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W, H = 64, 160, 96


def level0():
    z, y, x = np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij")
    d = (10 + ((x * 7 + y * 3 + z * 5) & 15)).astype(np.uint8)
    lab = np.zeros((N, N, N), np.uint32)
    for b, (bx, by, bz, r) in enumerate([(20, 22, 18, 9), (44, 30, 40, 11), (30, 48, 24, 7), (12, 44, 50, 6), (50, 12, 14, 8)]):
        inside = (x - bx) ** 2 + (y - by) ** 2 + (z - bz) ** 2 <= r * r
        d[inside] = 150 + 20 * b
        lab[inside] = b + 1
    return d, lab


def test_plain_c_consumer_draws_the_frame_of_the_python_host_side(tmp_path):
    import torch

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler on this machine")
    exe = str(tmp_path / "c_abi_demo")
    csrc = os.path.join(ROOT, "sub_volume_renderer_amd", "csrc")
    subprocess.run([gcc, "-O2", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", csrc, "-lsvr_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
                    f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    linked = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libsvr_hip.so" in linked and "torch" not in linked and "python" not in linked

    # the same scene through the Python host side
    d0, l0 = level0()
    d1 = (d0.reshape(32, 2, 32, 2, 32, 2).astype(np.uint32).sum(axis=(1, 3, 5)) // 8).astype(np.uint8)
    l1 = l0.reshape(32, 2, 32, 2, 32, 2).max(axis=(1, 3, 5))
    material = SubVolumeMaterial(lmip_threshold=120.0, lmip_fall_off=0.5, lmip_max_samples=10, fog_density=0.3,
                                 fog_color=(0.5, 0.5, 0.5), clim=(0.0, 255.0), gamma=1.0, opacity=1.0,
                                 colors=[(0.0, 0.0, 1.0), (0.0, 1.0, 1.0), (0.17, 1.0, 1.0), (0.33, 1.0, 1.0), (0.55, 1.0, 1.0), (0.8, 1.0, 1.0)])
    volume = SubVolume(material, data_segmentation_pairs=[(d0, l0), (d1, l1)], chunk_shape_in_pixels=[(16, 16, 16), (16, 16, 16)],
                       buffer_shape_in_chunks=[(2, 2, 2), (2, 2, 2)])
    volume.center_on_position((32.0, 32.0, 32.0), [(32, 32, 32), (32, 32, 32)])          # level 0 window [16, 48)^3, level 1 whole
    assert tuple(volume.wrapping_buffers[0]._current_logical_roi_in_pixels.offset) == (16, 16, 16)
    camera = PerspectiveCamera(fov=50, aspect=W / H, depth_range=(0.5, 2000.0))
    camera.world.position = -70.0, 95.0, -55.0
    camera.look_at((31.5, 31.5, 31.5))
    frame = volume.render(camera, W, H)
    torch.cuda.synchronize()

    # hand the C program the six matrices of this camera, column-major float32, in svr_camera's order
    cb = volume.camera_block(camera)
    with open(tmp_path / "camera.bin", "wb") as f:
        for name in ("world", "world_inv", "cam", "cam_inv", "proj", "proj_inv"):
            f.write(np.asarray(list(getattr(cb, name)), np.float32).tobytes())
    run = subprocess.run([exe, str(tmp_path / "c"), str(tmp_path / "camera.bin")], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    rgba = np.fromfile(tmp_path / "c.rgba.f32", np.float32).reshape(H, W, 4)
    label = np.fromfile(tmp_path / "c.label.u32", np.uint32).reshape(H, W)
    flags = np.fromfile(tmp_path / "c.flags.u8", np.uint8).reshape(H, W)
    assert (flags == 2).sum() > 500 and len(np.unique(label[flags == 2])) >= 3
    np.testing.assert_array_equal(flags, frame.flags.cpu().numpy())
    np.testing.assert_array_equal(label, frame.label.cpu().numpy())
    np.testing.assert_array_equal(rgba, frame.rgba.cpu().numpy())             # same kernel, same inputs: the same bits

    # and with its own camera (no matrices given) it still finds the volume
    alone = subprocess.run([exe, str(tmp_path / "own")], capture_output=True, text=True)
    assert alone.returncode == 0, alone.stdout + alone.stderr
    assert os.path.getsize(tmp_path / "own.ppm") > W * H * 3
