#!/bin/bash
# Kernel time per view with and without the micro-block copy of LOD 0 (svr_lod_desc::blocked_twin), shipped library.
set -o pipefail
for cfg in "C2 native" "C2 float32" "C5 native"; do
  set -- $cfg
  for cam in K1 K2 -x -y -z diag; do
    for t in notwin twin twinall; do
      timeout -k 10 300 python tools/exp_view_ms.py $cam $2 $1 $t 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
    done
  done
done
