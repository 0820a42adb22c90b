#!/usr/bin/env python3
"""Build variants of libsvr_hip.so with extra -D flags into _ab/libs/<name>.so for same-box A/B runs
(`SVR_LIB=_ab/libs/<name>.so python bench.py ...`).  usage: ab_build.py name=-DFLAG1,-DFLAG2 ...
`exp=-DSVR_EXPERIMENTS` is the build that knows the SVR_* environment switches and the timing bits of svr_set_variant
(tools/README.md); the shipped library has neither."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out_dir = os.path.join(ROOT, "_ab", "libs")
os.makedirs(out_dir, exist_ok=True)
g.build_hip()
for spec in sys.argv[1:]:
    name, _, flags = spec.partition("=")
    flags = [f for f in flags.split(",") if f]
    objdir = os.path.join(ROOT, "_ab", "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in g.HIP_FLAGS if f != "-shared"] + flags

    def cc(unit):
        src, obj, extra = unit
        subprocess.run(["hipcc", *cflags, *extra, "-c", os.path.join(g.CSRC, src), "-o", obj], check=True)
        return obj

    # every translation unit is rebuilt with the flags (-DSVR_EXPERIMENTS reaches svr_api.hip and ring_kernels.hip too)
    units = [(s, os.path.join(objdir, s.replace(".hip", ".o")), []) for s in g.HIP_SOURCES]
    units += [("march_kernel.hip", os.path.join(objdir, f"march_nl{k}.o"), [f"-DSVR_NL={k}"]) for k in range(1, 9)]
    with ThreadPoolExecutor(8) as pool:
        objs = list(pool.map(cc, units))
    lib = os.path.join(out_dir, name + ".so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"], check=True)
    print("built", lib)
