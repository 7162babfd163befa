#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export P3HIP_LIB=$R/build/libp3hip_diag.so
for spec in "b12c128btl3 1024 0" "b12c128btl3 1024 1" "b12c128btl3 1024 2" "b12c128btl3 256 1" "b12c256btl3 1024 1" "b12c256btl3 1024 0" "b8c128nbt 1024 1"; do
  set -- $spec
  P3DIAG_LAUNCH=$3 timeout -k 10 120 python3 $R/tools/gpu_spans.py $1 $2
done > $OUT/r02_spans.log 2>&1
cat $OUT/r02_spans.log
