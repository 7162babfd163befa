#!/bin/bash
# how many logical CPUs does the job have, and does the self-play host gain from more threads than 16?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
{ nproc; lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|Model name|NUMA node\(s\)"; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:40])"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > $OUT/host_cpus.log 2>&1
cat $OUT/host_cpus.log
for thr in 16 24 32; do
  SP_GROUPS=3 timeout -k 10 120 python3 $R/tools/gpu_selfplay.py b8c128nbt 3072 $thr 8 2>&1 | tail -1
done | tee $OUT/host_threads.log
