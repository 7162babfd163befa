#!/bin/bash
# C = 128 trunks: do the two 4-wave workgroups of a CU run faster when one of them starts late?
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
timeout -k 10 60 $R/tools/probe/build/cu_map_probe > $OUT/cu_map_probe.log 2>&1
tail -3 $OUT/cu_map_probe.log
timeout -k 10 500 python3 $R/tools/gpu_ab_env.py b12c128btl3 - P3HIP_PAIR_OFFSET=4000 P3HIP_PAIR_OFFSET=8000 P3HIP_PAIR_OFFSET=14000 P3HIP_PAIR_OFFSET=25000 P3HIP_PAIR_OFFSET=8000,P3HIP_PAIR_SHIFT=3 P3HIP_PAIR_OFFSET=8000,P3HIP_PAIR_SHIFT=0 > $OUT/pair_offset_ab.log 2>&1
cat $OUT/pair_offset_ab.log
