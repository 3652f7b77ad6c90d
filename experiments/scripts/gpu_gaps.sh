#!/bin/bash
# Where the step's idle time sits: rocprofv3 kernel trace of a short bench run, then per-predecessor gap sums over the last timed step
# and the neighbourhood of every small library launch (copyBuffer / fill).  Usage: bash scripts/gpu_gaps.sh <tag>   (EXTRA = extra bench flags)
TAG=${1:-x}; LIST=${LIST:-0}; mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$TAG -o trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --table-steps 0 $EXTRA > $R/gpurun_out/gaps_$TAG.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/gaps_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")) for r in rows))
# steps: every step has exactly one cast_transpose_multi_kernel launch; take the last complete one
idx = [i for i, e in enumerate(ev) if e[2].startswith("cast_transpose_multi")]
lo, hi = idx[-2], idx[-1]
step = ev[lo:hi]
span = step[-1][1] - step[0][0]
busy = sum(e[1] - e[0] for e in step)
print(f"step: {len(step)} launches, span {span/1e6:.3f} ms, kernel sum {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
gap_by = collections.defaultdict(lambda: [0, 0])
for a, b in zip(step[:-1], step[1:]):
    g = max(0, b[0] - a[1])
    k = (a[2][:44], b[2][:44])
    gap_by[k][0] += g; gap_by[k][1] += 1
for k, (g, n) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{g/1e3:9.1f} us  n={n:4d}  avg {g/n/1e3:6.2f}   {k[0]:44s} -> {k[1]}")
lib = [e for e in step if ("at::native" in e[2] or "rocprim" in e[2] or "rocclr" in e[2] or "elementwise" in e[2])]
print(f"library (torch / rocprim / copy) launches in the step: {len(lib)}, {sum(e[1]-e[0] for e in lib)/1e6:.3f} ms")
first_udm = next(i for i, e in enumerate(step) if e[2].startswith("gemm_") or e[2].startswith("norm_fwd"))
print(f"step start -> first GEMM/norm launch: {(step[first_udm][0]-step[0][0])/1e6:.3f} ms over {first_udm} launches")
if "$LIST" == "1":
    for i, e in enumerate(step):
        if e in lib:
            print(f"{i:4d} {(e[1]-e[0])/1e3:7.1f} us  {e[2][:230]}")
print("---- small library launches and their neighbours")
seen = collections.Counter()
for i, e in enumerate(step):
    if "rocclr" in e[2] or "FillFunctor" in e[2] or "elementwise" in e[2] or "reduce_kernel" in e[2] or "index" in e[2]:
        key = (step[i - 1][2][:50], e[2][:60], step[i + 1][2][:50] if i + 1 < len(step) else "")
        seen[key] += 1
for k, n in seen.most_common(40):
    print(n, " | ", k[0], " | ", k[1], " | ", k[2])
PY
rm -rf gpurun_out/gaps_$TAG
