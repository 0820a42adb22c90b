#!/bin/bash
# Collect rocprofv3 evidence for the march kernel: one --kernel-trace --stats pass and
# separate --pmc passes (never combined with other trace domains; FETCH_SIZE and WRITE_SIZE
# in passes of their own, per /opt/skills/guides/MI355X_MICROARCH.md).
# usage: tools/pmc.sh <tag> [mode] [n] [variant] [camera]
set -o pipefail
TAG=${1:-r01}; MODE=${2:-full}; NVOL=${3:-1024}; VAR=${4:-0}; CAM=${5:-K1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, args...  (counter passes only look at the march kernel)
  local name=$1; shift
  timeout -k 10 240 rocprofv3 "$@" --kernel-include-regex march --output-format csv -d $OUT/$name -- python3 $ROOT/tools/prof_driver.py $MODE $NVOL 5 $VAR $CAM > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
}
run trace --kernel-trace --stats &&
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY &&
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU &&
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr &&
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
