import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build_synth(); g.build_host_codecs()
from sub_volume_renderer_amd import zarr3, synth
d, l = synth.block_host(1024, 0, (0, 0, 0), (192, 512, 528), 4096, nthreads=0)
root = tempfile.mkdtemp()
zd = zarr3.write_array(root + "/d", d, (16, 16, 16), (64, 64, 64)); zl = zarr3.write_array(root + "/l", l, (16, 16, 16), (64, 64, 64))
for rep in range(2):
    tot_b = tot_t = 0
    for z in (zd, zl):
        for box in [((16, 0, 0), (32, 512, 528)), ((0, 16, 0), (192, 32, 528)), ((0, 0, 48), (192, 512, 96)), ((64, 0, 0), (80, 512, 528)), ((0, 0, 0), (192, 512, 528))]:
            sl = tuple(slice(a, b) for a, b in zip(*box))
            t = time.perf_counter(); x = z[sl]; dt = time.perf_counter() - t
            tot_b += x.nbytes; tot_t += dt
    print("pin", os.environ.get("SVR_ZARR_PIN", "default(1)"), "threads", zarr3._threads(), "rep", rep, "%.2f GB/s decoded over mixed requests" % (tot_b / tot_t / 1e9), flush=True)
