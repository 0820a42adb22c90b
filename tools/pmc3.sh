#!/bin/bash
# quick counter passes.  usage: tools/pmc3.sh <tag> <mode> <n> <variant> <camera>
TAG=${1:-q}; MODE=${2:-full}; NVOL=${3:-1024}; VAR=${4:-0}; CAM=${5:-K1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-include-regex march --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_driver.py $MODE $NVOL 3 $VAR $CAM > $OUT/$name.log 2>&1 || { echo "pass $name failed"; grep -iE "error|invalid|exceed" $OUT/$name.log | head -3; }
}
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run ta1 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
run grbm GRBM_GUI_ACTIVE
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
