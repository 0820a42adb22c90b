#!/usr/bin/env python3
"""One entry of profiles/<round>/traffic.json from the PMC passes of tools/profile_bench.sh: HBM-side bytes per march launch
(FETCH_SIZE and WRITE_SIZE collected in separate passes, gfx950 correction applied as
/opt/skills/guides/MI355X_MICROARCH.md prescribes), stamped with the command, the workload and the hash of the
kernel sources it was taken on — bench.py only carries the figure into `roofline.traffic` when all of them match
what it is running — and `binding`: the resource the SQ / TA counters of the same passes show busiest.
usage: make_traffic.py <profile_dir> <out.json> <command...>"""
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


def mean_counter(name, pat="8, false,", passes="pmc_*"):        # the production build of march_span (not the instrumented <..., true, ...> one)
    vals = []
    # (the full-mode passes only: the LMIP-only passes live in lmip_pmc_* beside them — round 2's glob took those in too,
    # 5 LMIP dispatches among 60)
    for f in glob.glob(os.path.join(prof, passes, "**", "*counter_collection.csv"), recursive=True):
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
                 "blocked_twin": bool(cfgd.get("blocked_twin", [False])[0]),
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
# ---- what binds the kernel: busy share of the candidates, from the same passes
# (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units": x 4 = SIMD cycles;
#  GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the kernel's duration in shader cycles; TA_BUSY_avr is per TA)
insts_valu, _ = mean_counter("SQ_INSTS_VALU")
act_valu, _ = mean_counter("SQ_ACTIVE_INST_VALU")
gui, _ = mean_counter("GRBM_GUI_ACTIVE")
ta_busy, _ = mean_counter("TA_BUSY_avr")
tcp_stall, _ = mean_counter("TCP_PENDING_STALL_CYCLES_sum")
wave_cycles, _ = mean_counter("SQ_WAVE_CYCLES")
wait_any, _ = mean_counter("SQ_WAIT_ANY")
SIMDS, CUS = 1024, 256
if insts_valu and act_valu and gui:
    cycles = gui / 8.0
    cand = {"valu_issue": 4.0 * act_valu / (SIMDS * cycles)}
    if ta_busy:
        cand["ta_address_path"] = ta_busy / cycles
    if tcp_stall:
        cand["l1_miss_stall"] = tcp_stall / (CUS * cycles)
    top = max(cand, key=cand.get)
    doc["binding"] = {
        "resource": top, "frac": cand[top], "insts": insts_valu,
        "kernel_cycles": cycles,
        "busy_share": {k: round(v, 4) for k, v in cand.items()},
        "hbm_busy_share": round(doc["traffic_bytes_per_launch"] / (cycles / 2.4e9) / 8e12, 4) if cycles else None,
        "def": "valu_issue = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / (1024 SIMDs x kernel cycles); ta_address_path = TA_BUSY_avr / "
               "kernel cycles; l1_miss_stall = TCP_PENDING_STALL_CYCLES_sum / (256 CUs x kernel cycles); kernel cycles = "
               "GRBM_GUI_ACTIVE / 8 (under the profiler); hbm_busy_share = traffic at 2.4 GHz nominal over 8 TB/s",
        "wave_residency": {"waiting_share": (wait_any / wave_cycles) if wave_cycles and wait_any else None},
        "SQ_INSTS_VALU": insts_valu, "SQ_ACTIVE_INST_VALU": act_valu, "GRBM_GUI_ACTIVE": gui,
        "TA_BUSY_avr": ta_busy, "TCP_PENDING_STALL_CYCLES_sum": tcp_stall,
    }
# ---- the same for LMIP-only frames (tools/prof_driver.py lmip: the north_star workload), when those passes exist
lm = {k: mean_counter(k, passes="lmip_pmc_*")[0] for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "TA_BUSY_avr",
                                                            "TCP_PENDING_STALL_CYCLES_sum", "FETCH_SIZE", "WRITE_SIZE")}
if lm["SQ_ACTIVE_INST_VALU"] and lm["GRBM_GUI_ACTIVE"]:
    cyc = lm["GRBM_GUI_ACTIVE"] / 8.0
    cand = {"valu_issue": 4.0 * lm["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc)}
    if lm["TA_BUSY_avr"]:
        cand["ta_address_path"] = lm["TA_BUSY_avr"] / cyc
    if lm["TCP_PENDING_STALL_CYCLES_sum"]:
        cand["l1_miss_stall"] = lm["TCP_PENDING_STALL_CYCLES_sum"] / (CUS * cyc)
    top = max(cand, key=cand.get)
    traffic_lmip = int((lm["FETCH_SIZE"] or 0) * 1024 * 2 + (lm["WRITE_SIZE"] or 0) * 1024) if lm["FETCH_SIZE"] else None
    doc["lmip"] = {
        "command": "tools/prof_driver.py lmip 1024 12 0 K1 --ring-storage " + ("float32" if cfgd["ring_storage"] == "float32" else "native"),
        "traffic_bytes_per_launch": traffic_lmip,
        "binding": {"resource": top, "frac": cand[top], "insts": lm["SQ_INSTS_VALU"], "kernel_cycles": cyc,
                    "busy_share": {k: round(v, 4) for k, v in cand.items()},
                    "hbm_busy_share": round(traffic_lmip / (cyc / 2.4e9) / 8e12, 4) if traffic_lmip else None},
    }
with open(out, "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(doc["workload"]), doc["traffic_bytes_per_launch"], json.dumps(doc.get("binding", {}).get("busy_share")))
