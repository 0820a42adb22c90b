#!/bin/bash
# C2 on float32 rings (the reference's layout), full mode: slab length x LDS bytes per wave.
# prints: ms per step (4 frames in flight), one frame at a time, kernel alone
# (needs a library built with -DSVR_EXPERIMENTS: the knobs are compiled out of the shipped one)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# the knobs exist only in a -DSVR_EXPERIMENTS build: python tools/ab_build.py exp=-DSVR_EXPERIMENTS
[ -f "$ROOT/_ab/libs/exp.so" ] || { echo "build _ab/libs/exp.so first (tools/ab_build.py exp=-DSVR_EXPERIMENTS)"; exit 1; }
export SVR_LIB=$ROOT/_ab/libs/exp.so
for cfg in "SVR_NOP=1" "SVR_SLAB_SHIFT=1" "SVR_SLAB_SHIFT=2" "SVR_BRICK_BYTES=12288" "SVR_SLAB_SHIFT=1 SVR_BRICK_BYTES=12288" "SVR_BRICK_BYTES=16384" "SVR_SLAB_SHIFT=1 SVR_BRICK_BYTES=16384" "SVR_BRICK_BYTES=10240"; do
  echo "== $cfg"
  env $cfg python bench.py --ring-storage float32 --modes full --no-cpu-baseline --repeats 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sequential']['median_ms'], d['roofline']['kernel_ms'])"
done
