#!/bin/bash
# config 5 (2048^3, wide pixel footprint), full mode: default against brick slabs of half the length and 8 / 16 / 32 KiB of LDS per wave
# prints: ms per step (4 frames in flight), one frame at a time, kernel alone
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# the knobs exist only in a -DSVR_EXPERIMENTS build: python tools/ab_build.py exp=-DSVR_EXPERIMENTS
[ -f "$ROOT/_ab/libs/exp.so" ] || { echo "build _ab/libs/exp.so first (tools/ab_build.py exp=-DSVR_EXPERIMENTS)"; exit 1; }
export SVR_LIB=$ROOT/_ab/libs/exp.so
for cfg in "" "SVR_SLAB_SHIFT=1" "SVR_SLAB_SHIFT=1 SVR_BRICK_BYTES=8192" "SVR_SLAB_SHIFT=1 SVR_BRICK_BYTES=32768" "SVR_BRICK_BYTES=8192"; do
  echo "== $cfg"
  env $cfg python bench.py --config C5 --modes full --no-cpu-baseline --repeats 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sequential']['median_ms'], d['roofline']['kernel_ms'])"
done
