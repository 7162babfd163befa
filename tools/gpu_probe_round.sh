#!/bin/bash
# issue-port probe, clock/power sampler and per-kernel stats of the C=128 / C=384 trunks
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
timeout -k 10 120 $R/build/issue_probe > $OUT/r02_issue_probe.log 2>&1
timeout -k 10 200 python3 $R/tools/gpu_clock_sample.py 4 > $OUT/r02_clock_sample.log 2>&1
cd /tmp
for net in b12c128btl3 b14c384btl3 b10c384nbt; do
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/r02_prof_$net -o e -- python3 $R/tools/gpu_run_forward.py 10 $net > $OUT/r02_prof_$net.log 2>&1
  cp $(find $OUT/r02_prof_$net -name '*kernel_stats.csv' | head -1) $OUT/r02_${net}_kernel_stats.csv
done
cat $OUT/r02_issue_probe.log $OUT/r02_clock_sample.log
