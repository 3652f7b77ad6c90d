#!/bin/bash
# attention loop: parity tests + micro-bench of the wave-specialised backward kernels (UDM_DKV_WS=0: single-role dK/dV kernel)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 180 -p no:cacheprovider -k "attention" 2>&1 | tail -15 > gpurun_out/attn_tests.log
tail -5 gpurun_out/attn_tests.log
UDM_DKV_WS=0 timeout 300 python scripts/bench_attn.py 2>&1 | tail -1 | sed "s/^/single-role /"
timeout 300 python scripts/bench_attn.py 2>&1 | tail -1 | sed "s/^/specialised /"
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o t -- python3 $GRAFT_REPO_ROOT/scripts/bench_attn.py > /dev/null 2>&1; grep attn /tmp/pa/*kernel_stats.csv | cut -d, -f1,4 | sed 's/(anonymous namespace):://g' | cut -c1-120
