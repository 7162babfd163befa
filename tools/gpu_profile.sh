#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats around the default bench.py command (128 rounds: eight
#      game groups' streams share the GPU, kernel wall durations include time-slicing), and around
#      20 single-stream forward passes (what bench.py's roofline leg times with HIP events)
#   2. FETCH_SIZE and WRITE_SIZE of three resident forward passes, separate passes (HBM traffic)
#   3. SQ counters of the same passes (MFMA busy, LDS, waits)
# Raw output under gpurun_out/<tag>_*; summaries are copied into profiles/ by hand.
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_prof_bench -o b -- python3 $R/bench.py --steps 128 --no-cpu-baseline --no-pmc > $OUT/${TAG}_bench_under_rocprof.log 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_prof_engine -o e -- python3 $R/tools/gpu_run_forward.py 20 > $OUT/${TAG}_engine_under_rocprof.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_pmc_f -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_pmc_f.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_pmc_w -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_pmc_w.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS -d $OUT/${TAG}_pmc_sq1 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_pmc_sq1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/${TAG}_pmc_sq2 -o p -- python3 $R/tools/gpu_run_forward.py 3 > $OUT/${TAG}_pmc_sq2.log 2>&1
cd $R
python3 tools/pmc_summary.py $(find $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w -name '*counter_collection.csv') > $OUT/${TAG}_hbm_fetch_write_pmc.txt
python3 tools/pmc_summary.py $(find $OUT/${TAG}_pmc_sq1 $OUT/${TAG}_pmc_sq2 -name '*counter_collection.csv') > $OUT/${TAG}_sq_pmc_all_kernels.txt
cp $(find $OUT/${TAG}_prof_bench -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_bench_kernel_stats.csv
cp $(find $OUT/${TAG}_prof_engine -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_engine_only_kernel_stats.csv
tail -c 1500 $OUT/${TAG}_bench_under_rocprof.log
cat $OUT/${TAG}_hbm_fetch_write_pmc.txt | head -30
head -5 $OUT/${TAG}_bench_kernel_stats.csv
