#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/pmc.sh for the march kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "march_"


def rows(path):
    with open(path, newline="") as f:
        yield from csv.DictReader(f)


for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in rows(f):
        if pat in r["Name"]:
            print(f"kernel-trace: {r['Name'][:90]}  calls={r['Calls']} avg={float(r['AverageNs'])/1e6:.4f} ms "
                  f"min={float(r['MinNs'])/1e6:.4f} max={float(r['MaxNs'])/1e6:.4f}")
seen = set()
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in rows(f):
        if pat in r["Kernel_Name"] and r["Kernel_Name"] not in seen:      # one line per distinct kernel
            seen.add(r["Kernel_Name"])
            print(f"dispatch: {r['Kernel_Name'][:70]} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} vgpr={r['VGPR_Count']} "
                  f"agpr={r.get('Accum_VGPR_Count')} sgpr={r['SGPR_Count']} lds={r['LDS_Block_Size']} scratch={r['Scratch_Size']}")
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in rows(f):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counters (mean per dispatch over %d dispatches):" % (max(len(v) for v in acc.values()) if acc else 0))
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:34s} {sum(v)/len(v):16.1f}")
