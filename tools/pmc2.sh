#!/bin/bash
# extra TCP/TA/TCC counter passes for the march kernel.  usage: tools/pmc2.sh <tag> [mode] [n] [variant] [camera]
set -o pipefail
TAG=${1:-r01}; MODE=${2:-full}; NVOL=${3:-1024}; VAR=${4:-0}; CAM=${5:-K1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-include-regex march --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_driver.py $MODE $NVOL 3 $VAR $CAM > $OUT/$name.log 2>&1 || { echo "pass $name failed"; grep -iE "error|invalid|exceed" $OUT/$name.log | head -3; }
}
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcp3 TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum
run tcp4 TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run ta1 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run ta3 TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary2.txt 2>&1
cat $OUT/summary2.txt
