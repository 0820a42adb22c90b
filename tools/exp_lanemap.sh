#!/bin/bash
# Lane -> pixel map of the 8x8 wave tile: rows / columns by the projected x axis (default) against Z order (2x2 quads).
# needs _ab/libs/exp.so (python tools/ab_build.py exp=-DSVR_EXPERIMENTS).  prints kernel ms: full / lmip per view
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for st in native float32; do
for lm in 0 2; do
  echo "== ring storage $st  SVR_LANE_MAP=$lm"
  for cam in K1 K2 -x -y -z diag; do
    SVR_LIB=$ROOT/_ab/libs/exp.so SVR_LANE_MAP=$lm python $ROOT/tools/exp_view_ms.py $cam $st
  done
done; done
