#!/bin/bash
# One GPU session: kernel + e2e parity suites, then kernel micro-benchmarks.  Logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/device.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 180 -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/kernels.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --timeout 300 -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/e2e.log
timeout 600 python scripts/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1
tail -5 gpurun_out/kernels.log; tail -5 gpurun_out/e2e.log; tail -40 gpurun_out/bench_kernels.log
