#!/bin/bash
# rest of the GPU suite after the first failure of the health run + the new tests
mkdir -p gpurun_out; export TMPDIR=/tmp
UDM_LEDGER=gpurun_out/ledger_full.json timeout 3000 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/gpu_suite.log
cat gpurun_out/gpu_suite.log
