#!/bin/bash
# rocprofv3 counter passes over a minimal render driver (tools/prof_driver.py): one counter group per pass, never
# combined with trace domains, FETCH_SIZE / WRITE_SIZE in passes of their own (MI355X_MICROARCH.md).
# usage: [PROF_EXTRA='--config C5 --ring-storage float32'] tools/pmc.sh <tag> [groups] [mode] [n] [variant] [camera]
#   groups: comma list of  trace,sq,sq2,lds,tcp,tcp2,ta,tcc,fetch,write,grbm   (default: trace,sq,lds,tcc,fetch,write)
set -o pipefail
TAG=${1:-r02}; GROUPS_=${2:-trace,sq,lds,tcc,fetch,write}; MODE=${3:-full}; NVOL=${4:-1024}; VAR=${5:-0}; CAM=${6:-K1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
declare -A SETS=(
  [sq]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
  [sq2]="SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
  [lds]="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES"
  [tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
  [tcp2]="TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
  [ta]="TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  [tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [grbm]="GRBM_GUI_ACTIVE"
)
for G in ${GROUPS_//,/ }; do
  if [ "$G" = trace ]; then FLAGS="--kernel-trace --stats"; else FLAGS="--pmc ${SETS[$G]}"; fi
  timeout -k 10 240 rocprofv3 $FLAGS --kernel-include-regex march --output-format csv -d $OUT/$G -- python3 $ROOT/tools/prof_driver.py $MODE $NVOL 5 $VAR $CAM $PROF_EXTRA > $OUT/$G.log 2>&1 \
    || { echo "pass $G failed"; tail -5 $OUT/$G.log; exit 1; }
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
