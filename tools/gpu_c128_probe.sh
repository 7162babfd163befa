#!/bin/bash
# what bounds the C = 128 and C = 384 trunks: clock / power while they loop, and SQ counters of one forward pass
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
for net in b12c128btl3 b14c384btl3; do
  timeout -k 10 200 python3 $R/tools/gpu_clock_sample.py 3 $net > $OUT/r02_clock_sample_$net.log 2>&1
done
cd /tmp
net=b12c128btl3
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $OUT/r02_pmc_${net}_sq1 -o p -- python3 $R/tools/gpu_run_forward.py 3 $net > $OUT/r02_pmc_${net}_sq1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/r02_pmc_${net}_sq2 -o p -- python3 $R/tools/gpu_run_forward.py 3 $net > $OUT/r02_pmc_${net}_sq2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/r02_pmc_${net}_sq3 -o p -- python3 $R/tools/gpu_run_forward.py 3 $net > $OUT/r02_pmc_${net}_sq3.log 2>&1 || true
cd $R
python3 tools/pmc_summary.py $(find $OUT/r02_pmc_${net}_sq1 $OUT/r02_pmc_${net}_sq2 $OUT/r02_pmc_${net}_sq3 -name '*counter_collection.csv') > $OUT/r02_sq_pmc_$net.txt
cat $OUT/r02_clock_sample_*.log
head -40 $OUT/r02_sq_pmc_$net.txt
