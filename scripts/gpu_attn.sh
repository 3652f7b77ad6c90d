#!/bin/bash
# attention loop: parity tests + micro-bench of both dK/dV kernels (UDM_DKV_WS=0: single-role, default: wave-specialised) + cycle timeline
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 180 -p no:cacheprovider -k "attention" 2>&1 | tail -15 > gpurun_out/attn_tests.log
tail -5 gpurun_out/attn_tests.log
for p in 0 1; do UDM_DKV_WS=$p timeout 300 python scripts/bench_attn.py 2>&1 | tail -1 | sed "s/^/ws=$p /"; done
UDM_DKV_TIMELINE=3 timeout 300 python scripts/bench_attn.py 2>&1 | grep -E "^TL wave [04] j 1[456]"
