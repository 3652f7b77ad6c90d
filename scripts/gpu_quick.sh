#!/bin/bash
# quick loop: GEMM tests + kernel micro-bench + 1.4B bench without the CPU baseline
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 180 -p no:cacheprovider -k "${1:-gemm}" 2>&1 | tail -15 > gpurun_out/quick_tests.log
timeout 600 python scripts/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
tail -6 gpurun_out/quick_tests.log; python3 - <<'PY'
import json
s=open('gpurun_out/bench_kernels.log').read()
try:
    j=json.loads(s[s.index('{'):])
    for k,v in j.items(): print(k, v)
except Exception as e: print(s[-2000:])
PY
tail -2 gpurun_out/bench_quick.log | cut -c1-1800
