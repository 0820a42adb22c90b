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
            # rocprofv3's own units: VGPR_Count is not the code object's .vgpr_count (a 91-VGPR kernel reads 48 here) and
            # LDS_Block_Size leaves out the dynamic LDS of the launch (8 KiB per wave for the brick region): the code
            # object's figures are what tools/kernel_regs.py prints
            print(f"dispatch: {r['Kernel_Name'][:70]} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} "
                  f"rocprofv3_VGPR_Count={r['VGPR_Count']} rocprofv3_Accum_VGPR_Count={r.get('Accum_VGPR_Count')} "
                  f"rocprofv3_SGPR_Count={r['SGPR_Count']} LDS_Block_Size(static only)={r['LDS_Block_Size']} scratch={r['Scratch_Size']}")
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    if os.sep + "lmip_pmc_" in f:               # profile_bench.sh keeps its LMIP-only passes beside the full-mode ones
        continue
    for r in rows(f):
        # (the production instantiation only: the one instrumented launch bench.py makes to count steps — COUNT = true,
        # ", 8, true," in the name — would skew the means)
        if pat in r["Kernel_Name"] and ", 8, true," not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counters (mean per dispatch over %d dispatches):" % (max(len(v) for v in acc.values()) if acc else 0))
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:34s} {sum(v)/len(v):16.1f}")
