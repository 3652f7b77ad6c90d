#!/bin/bash
# rocprofv3 kernel-trace summary of the bench command (short run, no CPU baseline).  Usage: bash scripts/gpu_prof.sh <tag>
TAG=${1:-x}; mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing $EXTRA > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
F=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && head -45 "$F" > gpurun_out/kernel_stats_$TAG.csv
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/kernel_stats_$TAG.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:58]
    print(f'{n:58s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} pct={float(r["Percentage"]):5.2f}')
print("total ms per step ~", tot/4/1e6)
PY
tail -1 gpurun_out/prof_$TAG.log | cut -c1-300
