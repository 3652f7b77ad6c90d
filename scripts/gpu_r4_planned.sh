#!/bin/bash
# Round 4: robustness of the CU-planned GEMM dispatch (ddp.py `overlap_planned` sets it for the rest of a backward): every workload for 20 steps with the plan
# in force for the WHOLE step (UDM_GEMM_CUS=224) - the [MASK]-row counts, and with them the compacted head / last-block shapes, differ every step.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for wl in unidisc-1.4b-l1280 unidisc-s-l384 unidisc-1.4b-interleaved-l4608; do
  UDM_GEMM_CUS=224 timeout 500 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --table-steps 0 > gpurun_out/r4_plan224_$wl.json 2> gpurun_out/r4_plan224_$wl.err
  echo "$wl rc=$?"; cut -c1-330 gpurun_out/r4_plan224_$wl.json; tail -2 gpurun_out/r4_plan224_$wl.err
done
