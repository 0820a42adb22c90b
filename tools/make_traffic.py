#!/usr/bin/env python3
"""profiles/<round>/traffic.json from the PMC passes of tools/profile_bench.sh: HBM-side bytes per march launch
(FETCH_SIZE and WRITE_SIZE collected in separate passes, gfx950 correction applied as
/opt/skills/guides/MI355X_MICROARCH.md prescribes), stamped with the command, the workload and the hash of the
kernel sources it was taken on — bench.py only carries the figure into `roofline.traffic` when all of them match
what it is running.  usage: make_traffic.py <profile_dir> <out.json> <command...>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

prof, out = sys.argv[1], sys.argv[2]
command = " ".join(sys.argv[3:])


def mean_counter(name, pat="8, false,"):        # the production build of march_span (not the instrumented <..., true, ...> one)
    vals = []
    for f in glob.glob(os.path.join(prof, "*", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == name and pat in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


fetch_kb, nf = mean_counter("FETCH_SIZE")
write_kb, nw = mean_counter("WRITE_SIZE")
hit, _ = mean_counter("TCC_HIT_sum")
miss, _ = mean_counter("TCC_MISS_sum")
if fetch_kb is None or write_kb is None:
    raise SystemExit("FETCH_SIZE / WRITE_SIZE passes not found under " + prof)
line = json.load(open(os.path.join(prof, "bench_line.json")))
cfgd = line["config"]
doc = {
    "command": command,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-include-regex march), mean per "
              f"march_span dispatch over {nf} / {nw} dispatches; tools/profile_bench.sh",
    "workload": {"config": "C2", "n": 1024, "width": 1920, "height": 1080, "camera": "K1",
                 "variant": cfgd["kernel_variant"], "ring_storage": cfgd["ring_storage"],
                 "kernel_source_sha16": bench.kernel_source_hash()},
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) loads -> x2 "
                  "(MI355X_MICROARCH.md, HBM); most of this kernel's fetched bytes are the 16-B/lane LDS-DMA brick "
                  "loads, the 1-B/lane gathers are uncalibrated (upper bound); WRITE_SIZE is exact",
    "traffic_bytes_per_launch": int(fetch_kb * 1024 * 2 + write_kb * 1024),
    "l2": {"TCC_HIT": hit, "TCC_MISS": miss},
    "note": "FETCH_SIZE counts the L2's memory-side (fabric) read requests; Infinity Cache hits are included, not "
            "excluded: read it as L2-miss bytes, an upper bound on HBM bytes.",
}
with open(out, "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(doc["workload"]), doc["traffic_bytes_per_launch"])
