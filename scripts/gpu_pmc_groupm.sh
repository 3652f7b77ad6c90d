#!/bin/bash
# VERDICT r4 item 2 (second half) / weak 4: does the tile walk order (rows of tiles an XCD works through per pass, UDM_GEMM_GROUP_M; default 8) change the GEMMs'
# fabric traffic, and what does that do to the CLOCK and the time?  Per setting: the un-profiled step time, then three separate profiler passes of the same
# 1-step bench (FETCH_SIZE; GRBM_GUI_ACTIVE; kernel trace) -> per kernel: corrected fetch bytes, GUI-active cycles, duration, effective clock = cycles / duration.
# Usage: bash scripts/gpu_pmc_groupm.sh <tag> "<group_m values>"
TAG=${1:-r05}; GMS=${2:-"8 4 16"}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/groupm; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for gm in $GMS; do
  export UDM_GEMM_GROUP_M=$gm
  python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 > $O/bench_$gm.json 2>/dev/null
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$gm -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/fetch_$gm.log 2>&1
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui_$gm -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/gui_$gm.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$gm -o trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/trace_$gm.log 2>&1
  find $O/trace_$gm -name "*kernel_trace.csv" -delete
done
cd $R; GMS="$GMS" TAG=$TAG python3 - <<'PY'
import csv, glob, collections, json, os
O = "gpurun_out/groupm"
def short(n): return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
res = {}
for gm in os.environ["GMS"].split():
    per = collections.defaultdict(dict)
    for key, d in (("fetch", f"{O}/fetch_{gm}"), ("gui", f"{O}/gui_{gm}")):
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])] += float(r["Counter_Value"]); cnt[short(r["Kernel_Name"])] += 1
        for k in agg:
            v = agg[k] / cnt[k]
            per[k][key] = v * 1024 * 2 if key == "fetch" else v / 8     # guide: FETCH_SIZE in KiB, x2 on gfx950 for wide streams; GUI_ACTIVE summed over the 8 XCDs
    for f in glob.glob(f"{O}/trace_{gm}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[short(r["Name"])]["avg_us"] = float(r["AverageNs"]) / 1e3
            per[short(r["Name"])]["calls"] = int(r["Calls"])
    try:
        s = open(f"{O}/bench_{gm}.json").read(); ms = json.loads(s[s.index("{"):])["ms_per_step"]
    except Exception:
        ms = None
    rows = {}
    for k, v in per.items():
        if "gemm" in k and "avg_us" in v and "gui" in v:
            rows[k] = dict(avg_us=round(v["avg_us"], 1), calls_in_trace=v["calls"], fetch_MB=round(v.get("fetch", 0) / 1e6, 1), gui_active_cycles=round(v["gui"]),
                           effective_clock_GHz=round(v["gui"] / (v["avg_us"] * 1e3), 3))
    res[f"group_m={gm}"] = dict(ms_per_step_unprofiled=ms, gemm_kernels=dict(sorted(rows.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls_in_trace"])[:8]))
json.dump(res, open(f"gpurun_out/gemm_groupm_fetch_clock_{os.environ['TAG']}.json", "w"), indent=1)
for gm, v in res.items():
    print(gm, "step ms", v["ms_per_step_unprofiled"])
    for k, r in list(v["gemm_kernels"].items())[:5]:
        print("   ", k[:70], r)
PY
