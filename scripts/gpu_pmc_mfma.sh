#!/bin/bash
# MFMA-pipe utilisation per kernel of the bench step from PMC counters (own run: no kernel trace / stats beside --pmc).
# util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024), at the actual clock.
TAG=${1:-r01}; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmcm; export TMPDIR=/tmp; cd /tmp
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmcm/a -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing $EXTRA > $R/gpurun_out/pmcm/a.log 2>&1
cd $R; python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/pmcm/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(kn, r["Counter_Name"])] += 1
out = {}
for kn, c in agg.items():
    n = max(cnt[(kn, "GRBM_GUI_ACTIVE")], 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0) / n
    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n
    out[kn] = dict(launches=n, gui_active_cycles=gui, mfma_busy_cycles=mf, mfma_util=(mf / (gui * 128) if gui else None),
                   mfma_mops_bf16=c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) / n, sq_busy_cycles=c.get("SQ_BUSY_CYCLES", 0) / n,
                   lds_bank_conflict_cycles=c.get("SQ_LDS_BANK_CONFLICT", 0) / n, lds_active_cycles=c.get("SQ_LDS_IDX_ACTIVE", 0) / n)
json.dump(out, open("gpurun_out/pmc_mfma_$TAG.json", "w"), indent=1)
for kn, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles"] * kv[1]["launches"])[:14]:
    u = v["mfma_util"]
    print(f'{kn[:62]:62s} n={v["launches"]:4d} gui={v["gui_active_cycles"]:11.0f} mfma_busy={v["mfma_busy_cycles"]:13.0f} util={(u if u is not None else 0):6.3f} lds_conf/act={v["lds_bank_conflict_cycles"] / max(v["lds_active_cycles"], 1):5.3f}')
PY
tail -2 gpurun_out/pmcm/a.log | cut -c1-200
