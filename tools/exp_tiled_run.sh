#!/bin/bash
# The micro-block ring layout experiment (tools/exp_tiled_rings.py) over views / storages / configs, on one box.
# needs: python tools/ab_build.py exp=-DSVR_EXPERIMENTS tiled=-DSVR_EXPERIMENTS,-DSVR_EXP_TILED=7 tiled0=-DSVR_EXPERIMENTS,-DSVR_EXP_TILED=1
set -o pipefail
run() { # camera storage config
  SVR_LIB=_ab/libs/exp.so timeout -k 10 400 python tools/exp_tiled_rings.py ref $1 $2 $3 2>&1 | grep -v amdgpu.ids | tail -3 &&
  SVR_LIB=_ab/libs/tiled.so EXP_TILED_LODS=7 timeout -k 10 400 python tools/exp_tiled_rings.py tiled $1 $2 $3 2>&1 | grep -v amdgpu.ids | tail -6 &&
  SVR_LIB=_ab/libs/tiled0.so EXP_TILED_LODS=1 timeout -k 10 400 python tools/exp_tiled_rings.py tiled $1 $2 $3 2>&1 | grep -v amdgpu.ids | tail -6
}
run K1 native C2 && run K1 float32 C2 && run diag native C2 && run -x native C2 && SVR_FORCE_ZSPLIT=1020 run K1 native C5 && SVR_FORCE_ZSPLIT=1020 run K2 native C5
