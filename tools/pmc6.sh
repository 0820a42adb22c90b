#!/bin/bash
# VALU/SALU instruction counts only. usage: tools/pmc6.sh "<variants>" [camera] [mode]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CAM=${2:-K1}; MODE=${3:-full}
cd /tmp && export TMPDIR=/tmp
for VAR in $1; do
  OUT=$ROOT/gpurun_out/pmc6/$CAM-$MODE-$VAR; rm -rf $OUT; mkdir -p $OUT
  run() { local name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-include-regex march --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_driver.py $MODE 1024 2 $VAR $CAM > $OUT/$name.log 2>&1; }
  run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
  echo "== variant $VAR camera $CAM mode $MODE"
  python3 $ROOT/tools/pmc_summary.py $OUT | grep -v "^counters"
done
