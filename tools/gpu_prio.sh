#!/bin/bash
# C = 128 trunks: the two workgroups of a CU taking turns at the higher wave priority (shipped) against equal priorities
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
timeout -k 10 900 python3 -m pytest $R/tests/test_engine_gpu.py -x -q -m gpu > $OUT/pair_turns_tests.log 2>&1 || { tail -30 $OUT/pair_turns_tests.log; exit 1; }
tail -3 $OUT/pair_turns_tests.log
timeout -k 10 600 python3 $R/tools/gpu_ab_env.py b12c128btl3 - P3HIP_NO_PAIR_TURNS=1 > $OUT/pair_turns_ab.log 2>&1
timeout -k 10 300 python3 $R/tools/gpu_ab_env.py b8c128nbt - P3HIP_NO_PAIR_TURNS=1 >> $OUT/pair_turns_ab.log 2>&1
cat $OUT/pair_turns_ab.log
export P3HIP_LIB=$R/build/libp3hip_diag.so
{ echo "shipped (turns)"; P3DIAG_LAUNCH=1 timeout -k 10 120 python3 $R/tools/gpu_spans.py b12c128btl3 1024
  echo "P3HIP_NO_PAIR_TURNS=1"; P3HIP_NO_PAIR_TURNS=1 P3DIAG_LAUNCH=1 timeout -k 10 120 python3 $R/tools/gpu_spans.py b12c128btl3 1024
  echo "one workgroup per CU (batch 256)"; P3DIAG_LAUNCH=1 timeout -k 10 120 python3 $R/tools/gpu_spans.py b12c128btl3 256; } > $OUT/pair_turns_spans.log 2>&1
cat $OUT/pair_turns_spans.log
