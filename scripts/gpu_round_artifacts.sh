#!/bin/bash
# Everything the round's measurement artifacts come from, in one GPU call: the full `-m gpu` suite (parity ledger, RCCL world-1 check),
# the three bench workloads, the rocprofv3 kernel-trace summary and the two PMC passes (HBM traffic, MFMA utilisation) of the headline command.
# Usage: bash scripts/gpu_round_artifacts.sh r02     (outputs under gpurun_out/, copied to profiles/ by hand)
TAG=${1:-r02}; R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out; export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/gputests_$TAG.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_1.4b_b8_$TAG.json 2> gpurun_out/bench_1.4b_b8_$TAG.err
timeout 600 python bench.py --workload unidisc-s-l384 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_unidisc_s_b64_$TAG.json 2>/dev/null
timeout 600 python bench.py --workload unidisc-1.4b-interleaved-l4608 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1.4b_interleaved_l4608_b2_$TAG.json 2>/dev/null
bash scripts/gpu_prof.sh $TAG > gpurun_out/prof_summary_$TAG.log 2>&1
bash scripts/gpu_pmc_bench.sh $TAG > gpurun_out/pmc_traffic_summary_$TAG.log 2>&1
bash scripts/gpu_pmc_mfma.sh $TAG > gpurun_out/pmc_mfma_summary_$TAG.log 2>&1
rm -rf gpurun_out/pmcb gpurun_out/pmcm
tail -4 gpurun_out/gputests_$TAG.log; cut -c1-400 gpurun_out/bench_1.4b_b8_$TAG.json; tail -3 gpurun_out/prof_summary_$TAG.log; tail -16 gpurun_out/pmc_traffic_summary_$TAG.log; tail -16 gpurun_out/pmc_mfma_summary_$TAG.log
