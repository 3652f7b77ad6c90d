#!/bin/bash
# Round 4: in-box A/B of the UDM_EXP experiment bits on the headline step (same box, interleaved runs).  Usage: bash scripts/gpu_r4_exp.sh "0 1 2 4 8 0"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
WL=${WL:-unidisc-1.4b-l1280}
for e in ${1:-0 1 2 4 8 0}; do
  UDM_EXP=$e timeout 300 python bench.py --workload $WL --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --table-steps ${TABLE:-0} > gpurun_out/r4_exp_$e.json 2> gpurun_out/r4_exp_$e.err
  python3 - <<PY
import json
s=open("gpurun_out/r4_exp_$e.json").read()
try:
    j=json.loads(s[s.index("{"):]); print("UDM_EXP=$e", round(j["ms_per_step"],3), round(j["ms_per_step_median"],3), round(j["ms_per_step_min"],3))
    for r in sorted(j.get("roofline_table",[]), key=lambda r:-r.get("ms_per_step",0))[:12]:
        print("   ", r["entry_point"], round(r["ms_per_step"],3))
except Exception as ex: print("UDM_EXP=$e FAILED", ex, s[-300:])
PY
done
