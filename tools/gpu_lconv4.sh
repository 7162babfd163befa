#!/bin/bash
# C = 384 / C = 192 layer-wise trunks: 4-wave workgroups, two per CU (shipped) against 8-wave ones (P3HIP_LCONV_WG8=1),
# with and without the priority turns
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
timeout -k 10 600 python3 -m pytest $R/tests/test_engine_gpu.py -x -q -m gpu -k "384 or classic or c192 or baseline_configs or eval_match" > $OUT/lconv4_tests.log 2>&1 || { tail -40 $OUT/lconv4_tests.log; exit 1; }
tail -2 $OUT/lconv4_tests.log
for net in b14c384btl3 b10c384nbt b15c192_classic; do
  timeout -k 10 600 python3 $R/tools/gpu_ab_env.py $net - P3HIP_NO_PAIR_TURNS=1 P3HIP_LCONV_WG8=1
done > $OUT/r02_lconv4_ab.log 2>&1
cat $OUT/r02_lconv4_ab.log
