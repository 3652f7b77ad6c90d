#!/bin/bash
# round 3, first call: attention-backward noise anatomy, 24-block oracle parity, the bench with its new accounting, the self-launching N = 2 rehearsal
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/attn_bwd_noise.py > gpurun_out/attn_bwd_noise.json 2> gpurun_out/attn_bwd_noise.err
UDM_LEDGER=gpurun_out/ledger_24.json timeout 1500 python -m pytest tests/test_gpu_fullwidth_oracle.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "24blocks or 2blocks" 2>&1 | tail -25 > gpurun_out/fullwidth24.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r3_first.json 2> gpurun_out/bench_r3_first.err
UDM_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --table-steps 0 > gpurun_out/bench_gpus2_selflaunch.json 2> gpurun_out/bench_gpus2_selflaunch.err
cat gpurun_out/attn_bwd_noise.json; tail -5 gpurun_out/attn_bwd_noise.err; cat gpurun_out/fullwidth24.log; cut -c1-1500 gpurun_out/bench_r3_first.json; tail -3 gpurun_out/bench_r3_first.err; cut -c1-600 gpurun_out/bench_gpus2_selflaunch.json; tail -5 gpurun_out/bench_gpus2_selflaunch.err
