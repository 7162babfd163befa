#!/bin/bash
# parity suite of the engine, then A/B timing of environment switches on the same box:  tools/gpu_parity_ab.sh [net] spec...
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_engine_gpu.py -x -q -m gpu > gpurun_out/parity_ab_tests.log 2>&1 || { tail -40 gpurun_out/parity_ab_tests.log; exit 1; }
tail -2 gpurun_out/parity_ab_tests.log
timeout -k 10 900 python tools/gpu_ab_env.py "$@" 2>&1 | tee gpurun_out/parity_ab.log
