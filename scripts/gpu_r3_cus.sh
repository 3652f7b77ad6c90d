#!/bin/bash
# CU reservation: step with n CUs held by the spinning kernel, GEMMs planning for 256 (default) or fewer (UDM_GEMM_CUS)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/cus.log; : > $O
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 $2 2>/dev/null | python3 -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$1', round(j['ms_per_step'],2), round(j['ms_per_step_median'],2))" >> $O; }
UDM_GEMM_CUS=224 run hog16_plan224 "--hog-cus 16"
UDM_GEMM_CUS=224 run hog8_plan224 "--hog-cus 8"
UDM_GEMM_CUS=232 run hog16_plan232 "--hog-cus 16"
UDM_GEMM_CUS=192 run hog32_plan192 "--hog-cus 32"
cat $O
