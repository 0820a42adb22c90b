#!/usr/bin/env python3
"""Register / LDS / scratch / spill figures of every march kernel in a built object (code-object metadata, as the
loader sees it).  usage: kernel_regs.py [object file, default csrc/_obj/march_nl3.o] [name filter]
(rocprofv3's kernel trace prints VGPR_Count in allocation granules of its own and LDS_Block_Size without the dynamic
part: these are the numbers of the ELF notes.)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "sub_volume_renderer_amd", "csrc", "_obj", "march_nl3.o")
flt = sys.argv[2] if len(sys.argv) > 2 else "march_"
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as td:
    fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
for blk in notes.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt not in dem:
        continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk).group(1)  # noqa: E731
    short = re.sub(r"\(anonymous namespace\)::|void |\(MarchParams\)", "", dem)
    print(f"{short:55s} vgpr={g('vgpr_count'):>3s} sgpr={g('sgpr_count'):>3s} vgpr_spill={g('vgpr_spill_count'):>3s} "
          f"sgpr_spill={g('sgpr_spill_count'):>3s} scratch={g('private_segment_fixed_size'):>4s} lds_static={g('group_segment_fixed_size')}")
