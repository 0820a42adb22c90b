#!/bin/bash
# instruction-cache / issue counters of the march kernel. usage: tools/pmc5.sh "<variants>" [camera] [mode]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CAM=${2:-K1}; MODE=${3:-full}
cd /tmp && export TMPDIR=/tmp
for VAR in $1; do
  OUT=$ROOT/gpurun_out/pmc5/$CAM-$MODE-$VAR; rm -rf $OUT; mkdir -p $OUT
  run() { local name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-include-regex march --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_driver.py $MODE 1024 2 $VAR $CAM > $OUT/$name.log 2>&1; }
  run a SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
  run b SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
  run c SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run d SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
  echo "== variant $VAR camera $CAM mode $MODE"
  python3 $ROOT/tools/pmc_summary.py $OUT | grep -v "^counters"
done
