#!/bin/bash
# rocprofv3 evidence for bench.py's roofline blocks: for each ring storage of the default workload (native u8 rings,
# float32 rings) one --kernel-trace --stats pass of the bench command (full mode only, so that every march dispatch in
# the trace is the timed workload), then separate --pmc passes (never combined with other trace domains; FETCH_SIZE and
# WRITE_SIZE in passes of their own), then an LMIP-only trace + SQ + FETCH pass.  Writes gpurun_out/profile_<tag>/<storage>/
# and the PMC-derived gpurun_out/profiles_<round>/traffic.json (copy what is to be judged into profiles/<round>/ by hand).
# usage: tools/profile_bench.sh <tag> [round] [storages, default "native float32"]
TAG=${1:-r03}; ROUND=${2:-r03}; STORAGES=${3:-native float32}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TOP=$ROOT/gpurun_out/profile_$TAG; rm -rf $TOP; mkdir -p $TOP $ROOT/gpurun_out/profiles_$ROUND
cd /tmp && export TMPDIR=/tmp
ENTRIES=""
for ST in $STORAGES; do
  OUT=$TOP/$ST; mkdir -p $OUT
  # one frame at a time: per-dispatch durations stay comparable with roofline.kernel_ms
  ARGS="--steps 10 --warmup 2 --modes full --no-cpu-baseline --in-flight 1 --repeats 1 --prime-s 0 --no-float32-block --ring-storage $ST"
  CMD="python3 $ROOT/bench.py $ARGS"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo "trace pass failed ($ST)"; tail -5 $OUT/trace.log; exit 1; }
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
              "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $pass --kernel-include-regex march --output-format csv -d $OUT/pmc_$name -- $CMD > $OUT/pmc_$name.log 2>&1 || { echo "pass $name failed ($ST)"; tail -3 $OUT/pmc_$name.log; exit 1; }
  done
  # the LMIP mode (early-out + empty-space skipping) on its own: every march dispatch of this trace is an LMIP frame
  DRV="python3 $ROOT/tools/prof_driver.py lmip 1024 12 0 K1 --ring-storage $ST"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lmip -- $DRV > $OUT/trace_lmip.log 2>&1 || echo "lmip trace pass failed"
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-include-regex march --output-format csv -d $OUT/lmip_pmc_SQ -- $DRV > $OUT/lmip_pmc_SQ.log 2>&1 || echo "lmip pmc pass failed"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex march --output-format csv -d $OUT/lmip_pmc_FETCH -- $DRV > $OUT/lmip_pmc_FETCH.log 2>&1 || echo "lmip fetch pass failed"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex march --output-format csv -d $OUT/lmip_pmc_WRITE -- $DRV > $OUT/lmip_pmc_WRITE.log 2>&1 || echo "lmip write pass failed"
  timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr --kernel-include-regex march --output-format csv -d $OUT/lmip_pmc_TA -- $DRV > $OUT/lmip_pmc_TA.log 2>&1 || echo "lmip TA pass failed"
  timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex march --output-format csv -d $OUT/lmip_pmc_GRBM -- $DRV > $OUT/lmip_pmc_GRBM.log 2>&1 || echo "lmip GRBM pass failed"
  grep -h '^{' $OUT/trace.log | tail -1 > $OUT/bench_line.json
  echo "=== ring storage $ST: $CMD" > $OUT/summary.txt
  python3 $ROOT/tools/pmc_summary.py $OUT march_span >> $OUT/summary.txt 2>&1
  f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -1 $f > $OUT/kernel_stats_march.csv; grep march_ $f >> $OUT/kernel_stats_march.csv
  cp $f $OUT/kernel_stats_all.csv
  python3 $ROOT/tools/make_traffic.py $OUT $OUT/traffic_entry.json "bench.py $ARGS" >> $OUT/summary.txt 2>&1
  ENTRIES="$ENTRIES $OUT/traffic_entry.json"
  f2=$(find $OUT/trace_lmip -name "*kernel_stats.csv" | head -1); [ -n "$f2" ] && { echo "--- LMIP mode ($DRV)" >> $OUT/summary.txt; head -1 $f2 > $OUT/kernel_stats_march_lmip.csv; grep march_ $f2 >> $OUT/kernel_stats_march_lmip.csv; cat $OUT/kernel_stats_march_lmip.csv >> $OUT/summary.txt; }
  python3 - >> $OUT/summary.txt 2>&1 <<PY
import csv, glob
for d in ("lmip_pmc_SQ", "lmip_pmc_FETCH", "lmip_pmc_WRITE", "lmip_pmc_TA", "lmip_pmc_GRBM"):
    acc = {}
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "march_span" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("  LMIP %-28s %16.1f  (mean of %d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
  cat $OUT/summary.txt
done
python3 - <<PY
import json
entries = [json.load(open(p)) for p in "$ENTRIES".split()]
json.dump({"entries": entries}, open("$ROOT/gpurun_out/profiles_$ROUND/traffic.json", "w"), indent=1)
print("wrote gpurun_out/profiles_$ROUND/traffic.json with", [e["workload"]["ring_storage"] for e in entries])
PY
