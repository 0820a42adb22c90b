#!/bin/bash
# compact counter set for several variants. usage: tools/pmc4.sh "<variants>" [camera] [mode]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CAM=${2:-K1}; MODE=${3:-full}
cd /tmp && export TMPDIR=/tmp
for VAR in $1; do
  OUT=$ROOT/gpurun_out/pmc4/$CAM-$VAR; rm -rf $OUT; mkdir -p $OUT
  run() { local name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-include-regex march --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_driver.py $MODE 1024 2 $VAR $CAM > $OUT/$name.log 2>&1; }
  run a SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
  run b SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM
  run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUFFER_WAVEFRONTS_sum
  run d GRBM_GUI_ACTIVE
  echo "== variant $VAR camera $CAM mode $MODE"
  python3 $ROOT/tools/pmc_summary.py $OUT | grep -v "^counters"
done
