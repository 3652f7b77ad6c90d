#!/bin/bash
# How much of a step is spent BETWEEN kernels?  rocprofv3 kernel trace of the bench command; gaps between consecutive kernels of the last step (same queue), summed.
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_gaps -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --table-steps 0 > $R/gpurun_out/trace_gaps.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/trace_gaps/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in rows]
# the last step: from the last cast_transpose_multi launch to the end
starts = [i for i, k in enumerate(ks) if "cast_transpose_multi" in k[2]]
i0 = starts[-1]
seg = ks[i0:]
busy = sum(e - s for s, e, _, _ in seg)
span = seg[-1][1] - seg[0][0]
gaps, overlap = 0, 0
end = seg[0][1]
big = []
for j, (s, e, n, q) in enumerate(seg[1:], 1):
    if s > end:
        gaps += s - end
        if s - end > 20000: big.append((s - end, [(k[2][:50], k[3], (k[1] - k[0]) // 1000) for k in seg[max(0, j - 3):j + 3]]))
    else:
        overlap += min(end, e) - s
    end = max(end, e)
out = dict(kernels=len(seg), span_ms=span / 1e6, busy_sum_ms=busy / 1e6, idle_between_kernels_ms=gaps / 1e6, overlapped_ms=overlap / 1e6, mean_gap_us=gaps / 1e3 / max(1, len(seg) - 1),
           gaps_over_20us=sorted(big, reverse=True)[:8])
print(json.dumps(out))
open("gpurun_out/r06_step_kernel_gaps.json", "w").write(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/trace_gaps
