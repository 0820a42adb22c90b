#!/bin/bash
# same-box A/B of libraries built by tools/ab_build.py: kernel time (HIP events), one frame at a time, 4 in flight, LMIP kernel
# usage: exp_ab_libs.sh name1 name2 ...   (each: _ab/libs/<name>.so; "default" = the in-tree library)
for round in 1 2; do
for name in "$@"; do
  lib=""; [ "$name" != "default" ] && lib="SVR_LIB=$PWD/_ab/libs/$name.so"
  env $lib python bench.py --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'kernel %.4f' % d['roofline']['kernel_ms'], 'seq %.4f' % d['sequential']['median_ms'], 'pipelined %.4f' % d['spread']['median_ms'], 'lmip kernel %.4f' % d['lmip']['roofline']['kernel_ms'])"
done; done
