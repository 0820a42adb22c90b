#!/usr/bin/env python3
"""Build variants of libsvr_hip.so with extra -D flags into _ab/libs/<name>.so for same-box A/B runs
(`SVR_LIB=_ab/libs/<name>.so python bench.py ...`).  usage: ab_build.py name=-DFLAG1,-DFLAG2 ..."""
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

    def cc(k):
        obj = os.path.join(objdir, f"march_nl{k}.o")
        subprocess.run(["hipcc", *cflags, f"-DSVR_NL={k}", "-c", os.path.join(g.CSRC, "march_kernel.hip"), "-o", obj], check=True)
        return obj

    with ThreadPoolExecutor(8) as pool:
        march = list(pool.map(cc, range(1, 9)))
    others = [os.path.join(g.CSRC, "_obj", s.replace(".hip", ".o")) for s in g.HIP_SOURCES]
    lib = os.path.join(out_dir, name + ".so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *others, *march, "-ldl"], check=True)
    print("built", lib)
