#!/bin/bash
# round 3, second call: fp8 attention kernels (unit tests, step tests, full-width oracle test), fp8 forward micro-benchmark, per-parameter gradient-error dump at 24 blocks
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -k "fp8" 2>&1 | tail -30 > gpurun_out/fp8_kernel_tests.log
timeout 600 python scripts/bench_attn_fp8.py > gpurun_out/bench_attn_fp8.log 2>&1
UDM_LEDGER=gpurun_out/ledger_fp8.json timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullwidth_oracle.py tests/test_gpu_fullsize.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "fp8" 2>&1 | tail -30 > gpurun_out/fp8_step_tests.log
UDM_DUMP_GRAD_ERRS=gpurun_out/graderrs UDM_LEDGER=gpurun_out/ledger_24b.json timeout 1500 python -m pytest tests/test_gpu_fullwidth_oracle.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "2blocks_b2" 2>&1 | tail -8 > gpurun_out/fullwidth24b.log
cat gpurun_out/fp8_kernel_tests.log; cat gpurun_out/bench_attn_fp8.log; cat gpurun_out/fp8_step_tests.log; cat gpurun_out/fullwidth24b.log
