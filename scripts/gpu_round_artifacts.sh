#!/bin/bash
# Everything the round's measurement artifacts come from, in one GPU call: the full `-m gpu` suite (parity ledger, per-parameter gradient-error tables,
# RCCL world-1 check), the bench workloads (headline with the CPU baseline; config E with bf16 and fp8 attention; UniDisc-S), the rocprofv3 kernel-trace
# summaries of all three workloads and the two PMC passes (HBM traffic, MFMA utilisation) of the headline command, the CU-reservation measurement.
# Usage: bash scripts/gpu_round_artifacts.sh r03     (outputs under gpurun_out/, copied to profiles/ by scripts/collect_profiles.sh)
TAG=${1:-r06}; export TAG; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp
cd $R
[ -n "$SKIP_TESTS" ] || UDM_DUMP_GRAD_ERRS=gpurun_out/graderrs_$TAG UDM_LEDGER=gpurun_out/parity_ledger_$TAG.json timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1500 2>&1 | tail -15 > gpurun_out/gputests_$TAG.log
timeout 900 python bench.py --steps 25 --warmup 5 > gpurun_out/bench_1.4b_b8_$TAG.json 2> gpurun_out/bench_1.4b_b8_$TAG.err
# every workload of SURVEY §8(d) with its cpu_baseline in the same run
timeout 900 python bench.py --workload unidisc-s-l384 --steps 12 --warmup 3 > gpurun_out/bench_unidisc_s_b64_$TAG.json 2>/dev/null
timeout 900 python bench.py --workload unidisc-1.4b-interleaved-l4608 --steps 12 --warmup 3 > gpurun_out/bench_1.4b_interleaved_l4608_b2_$TAG.json 2>/dev/null
timeout 900 python bench.py --workload unidisc-1.4b-interleaved-l4608 --batch 1 --steps 12 --warmup 3 > gpurun_out/bench_1.4b_interleaved_l4608_b1_$TAG.json 2>/dev/null
timeout 900 python bench.py --workload unidisc-1.4b-l1280-adaln --steps 12 --warmup 3 > gpurun_out/bench_1.4b_adaln_b8_$TAG.json 2>/dev/null
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 --hog-cus 16 > gpurun_out/bench_1.4b_b8_16cus_held_$TAG.json 2>/dev/null
bash scripts/gpu_prof.sh $TAG > gpurun_out/prof_summary_$TAG.log 2>&1
cp gpurun_out/kernel_stats_$TAG.csv gpurun_out/kernel_stats_1.4b_b8_$TAG.csv
EXTRA="--workload unidisc-1.4b-interleaved-l4608" bash scripts/gpu_prof.sh ${TAG}e > gpurun_out/prof_summary_${TAG}e.log 2>&1
EXTRA="--workload unidisc-s-l384" bash scripts/gpu_prof.sh ${TAG}s > gpurun_out/prof_summary_${TAG}s.log 2>&1
bash scripts/gpu_pmc_bench.sh $TAG > gpurun_out/pmc_traffic_summary_$TAG.log 2>&1
bash scripts/gpu_pmc_mfma.sh $TAG > gpurun_out/pmc_mfma_summary_$TAG.log 2>&1
rm -rf gpurun_out/pmcb gpurun_out/pmcm gpurun_out/prof_$TAG gpurun_out/prof_${TAG}e gpurun_out/prof_${TAG}s
tail -4 gpurun_out/gputests_$TAG.log; cut -c1-400 gpurun_out/bench_1.4b_b8_$TAG.json; tail -3 gpurun_out/prof_summary_$TAG.log; tail -16 gpurun_out/pmc_traffic_summary_$TAG.log; tail -16 gpurun_out/pmc_mfma_summary_$TAG.log
python3 - <<'PY'
import json,glob
import os
for f in sorted(glob.glob('gpurun_out/bench_*_' + os.environ['TAG'] + '.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4))
    except Exception as e: print(f,'FAILED',e)
PY
