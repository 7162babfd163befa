#!/bin/bash
# Builds a timing-only variant of the kernels (results are garbage; see P3_EXP in
# p3achygo_amd/csrc/conv16.h):  tools/build_exp_variant.sh N  ->  p3achygo_amd/csrc/exp/libp3hip_eN.so
# Bits: 1 no lgkm waits in the K loop, 2 no fragment fetch, 4 no barrier in ring acquires,
# 8 no ring acquire, 16 no residual loads, 32 no output stores, 64 staggered workgroup start,
# 128 no 1x1 segments, 256 no 3x3 segments, 512 / 1024 static priority for waves 4-7 / 0-3,
# 2048 k_bdense without its K loop, 4096 k_bdense without its output stores.
# Time it against the production build with  P3HIP_LIB=<variant> python tools/gpu_block_timing.py
# or  python tools/gpu_ab.py libA.so libB.so.
set -e
cd "$(dirname "$0")/../p3achygo_amd/csrc"
n=$1
mkdir -p exp
[ -f engine.o ] || make engine.o
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -DP3_EXP=$n -c kernels.hip -o exp/kernels_e$n.o
hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libp3hip_e$n.so exp/kernels_e$n.o engine.o
echo "built exp/libp3hip_e$n.so"
