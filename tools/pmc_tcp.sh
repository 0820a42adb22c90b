#!/bin/bash
# TCP accesses per wave-load for several variants. usage: tools/pmc_tcp.sh "<variants>" [camera] [mode]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CAM=${2:-K1}; MODE=${3:-full}
cd /tmp && export TMPDIR=/tmp
for VAR in $1; do
  OUT=$ROOT/gpurun_out/tcpv/$CAM-$VAR; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUFFER_WAVEFRONTS_sum --kernel-include-regex march --output-format csv -d $OUT -- python3 $ROOT/tools/prof_driver.py $MODE 1024 2 $VAR $CAM > $OUT/log.txt 2>&1
  echo "variant $VAR camera $CAM: $(python3 $ROOT/tools/pmc_summary.py $OUT/.. march_ 2>/dev/null | tail -0)"
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
print("   accesses/wave-load = %.1f   L1->L2 req/wave-load = %.2f   wave-loads = %.1fM" % (m["TCP_TOTAL_CACHE_ACCESSES_sum"]/m["TA_BUFFER_WAVEFRONTS_sum"], m["TCP_TCC_READ_REQ_sum"]/m["TA_BUFFER_WAVEFRONTS_sum"], m["TA_BUFFER_WAVEFRONTS_sum"]/1e6))
PY
done
