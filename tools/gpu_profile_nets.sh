#!/bin/bash
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
for net in b12c128btl3 b14c384btl3 b10c384nbt; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/r02b_prof_$net -o e -- python3 $R/tools/gpu_run_forward.py 10 $net > $OUT/r02b_prof_$net.log 2>&1
  cp $(find $OUT/r02b_prof_$net -name '*kernel_stats.csv' | head -1) $OUT/r02_${net}_kernel_stats.csv
done
head -4 $OUT/r02_b12c128btl3_kernel_stats.csv
