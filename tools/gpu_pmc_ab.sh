#!/bin/bash
# SQ counters of k_block for two builds:  tools/gpu_pmc_ab.sh libA.so libB.so
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  export P3HIP_LIB=$R/$lib
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmcab_${tag}_1 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/pmcab_${tag}_1.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT -d $OUT/pmcab_${tag}_2 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/pmcab_${tag}_2.log 2>&1 || true
  echo "=== $tag"
  python3 $R/tools/pmc_summary.py $(find $OUT/pmcab_${tag}_1 $OUT/pmcab_${tag}_2 -name '*counter_collection.csv') | grep -A16 "k_block"
done
