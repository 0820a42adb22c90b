#!/bin/bash
# Ablation bounds of the brick path (C2, K1, full mode, kernel alone + one frame at a time), same box, back to back:
# what would be gained AT MOST by removing a cost altogether.  The ablated builds render wrong pixels (-DSVR_EXPERIMENTS).
#   build them first (in the container):  python tools/ab_build.py exp=-DSVR_EXPERIMENTS \
#       noload=-DSVR_EXPERIMENTS,-DSVR_EXP_NO_BRICK_LOADS nobox=-DSVR_EXPERIMENTS,-DSVR_EXP_NO_BRICK_LOADS,-DSVR_EXP_NO_BOX_REDUCE
# usage: tools/exp_ablate.sh [extra bench args, e.g. --ring-storage float32]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for round in 1 2; do
for name in shipped exp noload nobox; do
  lib=""; [ "$name" != "shipped" ] && lib="SVR_LIB=$ROOT/_ab/libs/$name.so"
  [ "$name" != "shipped" ] && [ ! -f "$ROOT/_ab/libs/$name.so" ] && continue
  env $lib python $ROOT/bench.py --modes full --no-cpu-baseline --no-float32-block --repeats 2 "$@" 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s round $round: kernel %.4f ms  one frame at a time %.4f ms  4 in flight %.4f ms' % ('$name', d['roofline']['kernel_ms'], d['sequential']['median_ms'], d['ms_per_step']))"
done; done
