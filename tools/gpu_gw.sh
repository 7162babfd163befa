#!/bin/bash
# GW form of k_block (P3HIP_GW=1): parity suite, then A/B timing against the shipped ring form on the same box.
set -o pipefail
mkdir -p gpurun_out
P3HIP_GW=1 timeout -k 10 240 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "golden or fused_block" > gpurun_out/gw_tests_small.log 2>&1 || { tail -30 gpurun_out/gw_tests_small.log; exit 1; }
tail -2 gpurun_out/gw_tests_small.log
P3HIP_GW=1 timeout -k 10 400 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/gw_tests.log 2>&1 || { tail -30 gpurun_out/gw_tests.log; exit 1; }
tail -2 gpurun_out/gw_tests.log
timeout -k 10 900 python tools/gpu_ab_env.py "$@" 2>&1 | tee gpurun_out/gw_ab.log
