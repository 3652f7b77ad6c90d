#!/bin/bash
# bench + rocprofv3 kernel-trace summary.  Usage: bash scripts/gpu_bench.sh [tag]
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python bench.py --workload unidisc-s-l384 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_small_$TAG.log 2>&1
timeout 1500 python bench.py --steps 10 --warmup 3 $BENCH_ARGS > gpurun_out/bench_$TAG.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head -3
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -40 "$F" > gpurun_out/kernel_stats_$TAG.csv
# keep the merged output small: drop the raw trace
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/bench_small_$TAG.log; tail -3 gpurun_out/bench_$TAG.log; tail -3 gpurun_out/prof_$TAG.log; cat gpurun_out/kernel_stats_$TAG.csv | cut -c1-160 | head -30
