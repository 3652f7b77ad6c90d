#!/bin/bash
# Which kernels run right before / after a given kernel name pattern in the bench step (rocprofv3 kernel trace, timestamps).  Usage: gpu_trace_neighbours.sh <pattern> [pattern2]
PAT=${1:-FillFunctor}; PAT2=${2:-copyBuffer}; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/ktr; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktr -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/ktr.log 2>&1
cd $R; python3 - "$PAT" "$PAT2" <<'PY'
import csv, glob, sys, collections
f = glob.glob("gpurun_out/ktr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
for pat in sys.argv[1:]:
    ctx = collections.Counter()
    for i, r in enumerate(rows):
        if pat in r["Kernel_Name"]:
            prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
            nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            ctx[(prev, nxt, round(dur))] += 1
    print("==", pat)
    for (p, n, d), c in ctx.most_common(12):
        print(f"{c:4d} x  [{p}]  ->  {pat} ({d} us)  ->  [{n}]")
PY
rm -rf gpurun_out/ktr
