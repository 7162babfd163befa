#!/bin/bash
# Same-box A/B of the block launches for VERDICT r3 item 2: the HIP k_block with one launch per run of residual blocks
# (P3HIP_NO_BFUSE=1) against the generated one-wave-per-SIMD k_blockw (P3HIP_BLOCKW=1), each under
# rocprofv3 --kernel-trace --stats and under two SQ counter passes (separate runs, no trace domains beside them).
# Raw output under gpurun_out/<tag>_*; the summaries are copied into profiles/ by hand.
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for V in hip blockw; do
  if [ $V = hip ]; then export P3HIP_NO_BFUSE=1; unset P3HIP_BLOCKW; else export P3HIP_BLOCKW=1; unset P3HIP_NO_BFUSE; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_ab_${V}_stats -o s -- python3 $R/tools/gpu_run_forward.py 20 > $OUT/${TAG}_ab_${V}_stats.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $OUT/${TAG}_ab_${V}_sq1 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_ab_${V}_sq1.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/${TAG}_ab_${V}_sq2 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_ab_${V}_sq2.log 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $OUT/${TAG}_ab_${V}_sq3 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_ab_${V}_sq3.log 2>&1 || true
  cp $(find $OUT/${TAG}_ab_${V}_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_ab_${V}_kernel_stats.csv
  python3 $R/tools/pmc_summary.py $(find $OUT/${TAG}_ab_${V}_sq1 $OUT/${TAG}_ab_${V}_sq2 $OUT/${TAG}_ab_${V}_sq3 -name '*counter_collection.csv') > $OUT/${TAG}_ab_${V}_sq_pmc.txt
done
cd $R
head -4 $OUT/${TAG}_ab_hip_kernel_stats.csv $OUT/${TAG}_ab_blockw_kernel_stats.csv
head -30 $OUT/${TAG}_ab_hip_sq_pmc.txt $OUT/${TAG}_ab_blockw_sq_pmc.txt
